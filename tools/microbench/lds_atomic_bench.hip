#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    __shared__ float sf[8192];
    int *si = reinterpret_cast<int *>(sf);
    unsigned long long *sl = reinterpret_cast<unsigned long long *>(sf);
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) sf[i] = 0.f;
    __syncthreads();
    int a = tid;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int addr = (a + u * 259) & 8191;     // lanes consecutive -> conflict-free
            if (MODE == 0) atomicAdd(&sf[addr], 1.0f);
            else if (MODE == 1) atomicAdd(&si[addr], 1);
            else if (MODE == 2) sf[addr] = (float)it;
            else if (MODE == 4) atomicAdd(&sl[addr & 4095], (unsigned long long)it);
            else acc += sf[addr];
        }
        a += 4099;
    }
    __syncthreads();
    out[blockIdx.x * 256 + tid] = sf[tid] + acc;
}
int main() {
    float *out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 2000, grid = 2048;
    for (int mode = 0; mode < 5; ++mode) {
        hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(s);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, out, iters);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, iters);
            hipEventRecord(e); hipEventSynchronize(e);
        }
        float ms; hipEventElapsedTime(&ms, s, e);
        const double ops = (double)grid * 256 * iters * 16;
        const char *names[5] = {"ds_add_f32", "ds_add_u32", "ds_write_b32", "ds_read_b32", "ds_add_u64"};
        printf("%-13s %8.3f ms  %7.1f Gops/s  %.2f lane-ops/clk/CU (2.4 GHz, 256 CUs)\n", names[mode], ms, ops / ms / 1e6, ops / (ms * 1e-3) / 2.4e9 / 256);
    }
    return 0;
}
