// vmem_pipe_bench.hip -- what ONE CU's vector-memory pipe moves per clock, by instruction kind (round 4).
// One 256-thread workgroup per CU (256 workgroups), each hammering its own small region so that everything after the first
// pass is an L2 hit (loads) or an L2-absorbed write (stores to a region that is rewritten), or a large region (HBM).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/vmem_pipe_bench.hip -o vmem_pipe_bench && ./vmem_pipe_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_base) : "memory", "m0");
}

// mode 0: LDS-DMA 16 B/lane; 1: global_load_dwordx4; 2: global_store_dwordx4 nt; 3: plain store; 4: DMA + nt store 2:1;
// 5: load + nt store 2:1
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(char *base, size_t region, int iters, unsigned *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *mine = base + (size_t)blockIdx.x * region;
    const unsigned lds0 = (unsigned)(uintptr_t)(void __attribute__((address_space(3))) *)smem;
    u32x4 acc = {0, 0, 0, 0};
    const u32x4 val = {threadIdx.x, 1u, 2u, 3u};
    const size_t per_iter = 256 * 16 * 8;                       // 8 instructions per wave and iteration: 32 KB per workgroup
    for (int it = 0; it < iters; ++it) {
        char *p = mine + ((size_t)it * per_iter) % region + (size_t)threadIdx.x * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            char *q = p + j * 4096;
            if (MODE == 0 || (MODE == 4 && j % 3 != 2)) dma16(q, lds0 + ((it * 8 + j) % 32) * 4096 + wave * 1024);
            else if (MODE == 1 || (MODE == 5 && j % 3 != 2)) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(q);
                acc ^= v;
            } else if (MODE == 2 || MODE == 4 || MODE == 5) __builtin_nontemporal_store(val, reinterpret_cast<u32x4 *>(q));
            else *reinterpret_cast<u32x4 *>(q) = val;
        }
        if (MODE == 0 || MODE == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc[0] == 0x12345u) sink[0] = acc[1];
}

int main() {
    const int nwg = 256;
    char *buf;
    unsigned *sink;
    const size_t big = (size_t)4 << 30;
    hipMalloc(&buf, big);
    hipMalloc(&sink, 64);
    hipMemset(buf, 1, big);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char *names[6] = {"LDS-DMA dwordx4", "global_load_dwordx4", "global_store_dwordx4 nt", "global_store_dwordx4",
                            "LDS-DMA + nt store (2:1)", "global_load + nt store (2:1)"};
    struct R { size_t region; const char *what; } regs[3] = {{64 << 10, "64 KB per CU (L2-resident)"}, {1 << 20, "1 MB per CU (256 MB: Infinity Cache)"},
                                                               {16 << 20, "16 MB per CU (4 GB: HBM)"}};
    for (auto &r : regs) {
        printf("region %s\n", r.what);
        for (int mode = 0; mode < 6; ++mode) {
            const int iters = 4096;
            void (*fn)(char *, size_t, int, unsigned *) = mode == 0 ? k<0> : mode == 1 ? k<1> : mode == 2 ? k<2> : mode == 3 ? k<3> : mode == 4 ? k<4> : k<5>;
            hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            hipLaunchKernelGGL(fn, dim3(nwg), dim3(256), 131072, 0, buf, r.region, 64, sink);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(fn, dim3(nwg), dim3(256), 131072, 0, buf, r.region, iters, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)nwg * iters * 32768.0;
            printf("  %-30s %7.2f TB/s = %6.1f GB/s per CU = %5.1f B/clk/CU at 2.1 GHz\n", names[mode], bytes / ms / 1e9,
                   bytes / ms / 1e6 / nwg, bytes / ms / 1e6 / nwg / 2.1);
        }
    }
    return 0;
}
