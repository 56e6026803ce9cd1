#!/bin/bash
# Tile-configuration sweep of the weight-gradient kernel: tools/gemm_bench.py --wgrad per configuration and size, on the
# MEASUREMENT build (make -C grafp_amd/csrc measure), where GRAFP_WGRAD_TILE = 0 ... 7 (T, S, L, S32, M32, L32, SG, LG)
# overrides the measured rule.  Correctness of every configuration is tests/test_gpu_gemm.py's job (the `tile` argument).
out=gpurun_out/wgrad_sweep.txt
: > $out
export GRAFP_HIP_LIB=$PWD/grafp_amd/libgrafp_hip_measure.so
for c in ${CLIPS:-512 2048}; do
  for t in ${BENCH_TILES:-default 1 2 3 4 5 6 7}; do
    echo "== clips $c tile $t" >> $out
    if [ $t = default ]; then timeout 200 python tools/gemm_bench.py --clips $c --wgrad 2>&1 | grep -v amdgpu | cut -c1-75 >> $out
    else GRAFP_WGRAD_TILE=$t timeout 200 python tools/gemm_bench.py --clips $c --wgrad 2>&1 | grep -v amdgpu | cut -c1-75 >> $out; fi
  done
done
