// mfma_data_bench.hip -- does the rate of v_mfma_f32_32x32x16_bf16 depend on the operand DATA (round 4)?
// 2048 workgroups x 4 waves, each wave 4000 x 8 MFMAs on 4 independent accumulators.  Operands: (0) one constant register for
// A and B (the form of mfma_valu_bench.hip), (1) eight different B registers of hashed bits with exponents near 1.0 (what a
// scan over real fingerprints multiplies), A hashed as well, (2) as 1 with A re-hashed every iteration by two VALU ops.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/mfma_data_bench.hip -o tools/microbench/mfma_data_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bf16x8 rnd8(unsigned seed) {
    bf16x8 v;
    for (int i = 0; i < 8; ++i) {
        const unsigned h = hash32(seed * 8 + i);
        v[i] = (short)((h & 0x807f) | ((0x7b + ((h >> 8) & 3)) << 7));     // sign, 7 mantissa bits, exponent 2^-4 .. 2^-1
    }
    return v;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a[8], b[8];
    for (int s = 0; s < 8; ++s) {
        if (MODE == 0) { for (int i = 0; i < 8; ++i) { a[s][i] = (short)(0x3f80 + threadIdx.x); b[s][i] = a[s][i]; } }
        else { a[s] = rnd8(gid * 16 + s); b[s] = rnd8(gid * 16 + 8 + s); }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            acc[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MODE == 0 ? a[0] : a[s], MODE == 0 ? a[0] : b[s], acc[s & 3], 0, 0, 0);
        }
        if (MODE == 2) {
#pragma unroll
            for (int s = 0; s < 8; ++s) a[s][it & 7] = (short)(a[s][it & 7] ^ (short)(it * 0x1d));
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[gid] = s;
}
template <int MODE> float run(float *out, int iters, int grid) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(s);
        hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e); hipEventSynchronize(e);
        hipEventElapsedTime(&ms, s, e);
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}
int main() {
    float *out; (void)hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 4000, grid = 2048;   // 8 workgroups (32 waves) per CU; 2048*4*4000*8 MFMAs = 256 000 per SIMD
    const double per_simd = 2048.0 * 4 * iters * 8 / 1024.0;
    const char *names[3] = {"constant operands", "hashed operands, 8 B registers", "hashed, A changing every iteration"};
    float ms[3] = {run<0>(out, iters, grid), run<1>(out, iters, grid), run<2>(out, iters, grid)};
    for (int m = 0; m < 3; ++m)
        printf("%-40s %.3f ms  %.2f ns per MFMA and SIMD  = 32 cycles at %.2f GHz  (%.0f TFLOP/s)\n", names[m], ms[m],
               ms[m] * 1e6 / per_simd, 32.0 / (ms[m] * 1e6 / per_simd), per_simd * 1024 * 32768.0 / (ms[m] * 1e-3) / 1e12);
    // long run: does the rate sag once the part has been busy for a while?
    for (int rep = 0; rep < 3; ++rep) {
        float t = run<1>(out, iters * 8, grid);
        printf("hashed operands, 8x longer launch       %.3f ms  %.2f ns per MFMA and SIMD\n", t, t * 1e6 / (per_simd * 8));
    }
    return 0;
}
