#!/bin/bash
# Builds measurement variants of the GEMM (GM_ABLATE = 1 no DMA, 2 no epilogue, 4 no MFMA, sums) next to the product
# library: tools/_ablate/libgrafp_hip_N.so; run tools/gemm_ablate.py on the GPU box afterwards.
set -e
cd "$(dirname "$0")/../grafp_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../tools/_ablate
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I../../include -Wno-unused-function -Wno-inline-asm"
OTHERS=$(ls _obj/*.o | grep -v "gemm.o\|-hip-")
for n in ${VARIANTS:-1 2 3 4 5 6}; do
  ( /opt/rocm/bin/hipcc $FLAGS -DGM_ABLATE=$n -c gemm.hip -o ../../tools/_ablate/gemm_$n.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS ../../tools/_ablate/gemm_$n.o -o ../../tools/_ablate/libgrafp_hip_$n.so ) &
done
wait
ls -la ../../tools/_ablate/*.so
