#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int BF16>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    typedef short bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 ab; for (int i = 0; i < 8; ++i) ab[i] = (short)(0x3f80 + threadIdx.x);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (BF16) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc[t], 0, 0, 0);
            else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NV; ++n) v[n & 7] = __builtin_fmaf(v[n & 7], b, a);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int BF16> float run(float *out, int iters, int grid) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(s);
        hipLaunchKernelGGL((k<NV, BF16>), dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e); hipEventSynchronize(e);
        hipEventElapsedTime(&ms, s, e);
    }
    return ms;
}
int main() {
    float *out; (void)hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 4000, grid = 2048;   // 8 workgroups (32 waves) per CU
    printf("f32 MFMA 32x32x2 (64 cycles each) + NV independent-chain v_fma_f32 per MFMA:\n");
    printf("  NV=0  %.3f ms\n", run<0, 0>(out, iters, grid));
    printf("  NV=4  %.3f ms\n", run<4, 0>(out, iters, grid));
    printf("  NV=8  %.3f ms\n", run<8, 0>(out, iters, grid));
    printf("  NV=16 %.3f ms\n", run<16, 0>(out, iters, grid));
    printf("  NV=32 %.3f ms\n", run<32, 0>(out, iters, grid));
    printf("bf16 MFMA 32x32x16 (32 cycles each?) + NV v_fma_f32 per MFMA:\n");
    printf("  NV=0  %.3f ms\n", run<0, 1>(out, iters, grid));
    printf("  NV=4  %.3f ms\n", run<4, 1>(out, iters, grid));
    printf("  NV=8  %.3f ms\n", run<8, 1>(out, iters, grid));
    printf("  NV=16 %.3f ms\n", run<16, 1>(out, iters, grid));
    return 0;
}
