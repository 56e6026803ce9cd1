#!/bin/bash
# Tile-configuration sweep of the weight-gradient kernel (GRAFP_WGRAD_TILE is read once per process):
#   correctness of every forced configuration, then tools/gemm_bench.py --wgrad per configuration and size.
out=gpurun_out/wgrad_sweep.txt
: > $out
for t in ${TILES:-s m l}; do
  echo "== pytest with tile $t" >> $out
  GRAFP_WGRAD_TILE=$t timeout 300 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_kernels.py -q -x -k wgrad 2>&1 | tail -3 >> $out
done
for c in ${CLIPS:-512 2048}; do
  for t in ${BENCH_TILES:-default S L s m l}; do
    echo "== clips $c tile $t" >> $out
    if [ $t = default ]; then timeout 200 python tools/gemm_bench.py --clips $c --wgrad 2>&1 | grep -v amdgpu | cut -c1-75 >> $out
    else GRAFP_WGRAD_TILE=$t timeout 200 python tools/gemm_bench.py --clips $c --wgrad 2>&1 | grep -v amdgpu | cut -c1-75 >> $out; fi
  done
done
