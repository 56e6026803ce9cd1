// gemm_traffic_bench.hip -- the MEMORY side of the stage-2 FFN-1 product (1024 x 256 x 524 288, conv1x1_gemm on the
// 256 x 256 tile) without any arithmetic: do its L2->LDS DMA stream and its output stores overlap, and does it matter
// WHICH WAVES issue them?  (round 5; VERDICT r4 item 4: "build the store-wave-group variant, with a stop rule")
//
// One workgroup per 256 x 256-tile range exactly as gemm_plan cuts it (4 row tiles x 256 column ranges x 8 tiles, XCD remap
// so that the four row tiles of a column range share an L2), a ring of 4 x 32 KB stages filled with the kernel's own piece
// pattern (W: 16 rows x 64 B per 1-KiB piece out of a 512 KB matrix; X: 2 k-rows x 512 B per piece, row stride M * 2 B),
// chunk c + 3 issued when chunk c + 1 has landed, one barrier per chunk, optional delay per chunk standing in for the 32
// MFMAs (1 024 cycles at the matrix peak), 128 KB of nt stores per tile as 16-byte row pieces.
//   mode: 1 W pieces | 2 X pieces | 4 stores issued by the DMA waves, interleaved with the pieces (what gemm_xl.h does)
//         | 8 stores issued by FOUR MORE waves that never load (own vmcnt) | 16 the X pieces come from a padded row pitch
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/gemm_traffic_bench.hip -o gemm_traffic_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_base) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = n >> 3, r = n & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

constexpr int KC = 32, TR = 256, TN = 256, STAGE = 32768, NS = 4;

template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void traffic(const unsigned short *__restrict__ A, const unsigned short *__restrict__ X,
                                                         unsigned short *__restrict__ Y, int64_t pitch, int K, int row_tiles,
                                                         int tiles_range, int nblocks, int delay) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(void __attribute__((address_space(3))) *)smem;
    const int logical = xcd_remap(blockIdx.x, nblocks);
    const int rt = logical % row_tiles, range = logical / row_tiles;
    const int nch = K / KC, T = tiles_range * nch;
    const int r0 = rt * TR;
    const int64_t col0 = (int64_t)range * tiles_range * TN;
    const bool loader = wave < 4;
    // W piece q = wave + 4 j: rows 16 q .. 16 q + 15, 64 B each; X piece p = wave + 4 j: k-rows 2 p, 2 p + 1, 512 B each
    const unsigned short *srcw = A + (size_t)(r0 + 16 * (wave & 3) + (lane >> 2)) * K + (lane & 3) * 8;
    const unsigned short *srcx = X + (size_t)(2 * (wave & 3) + lane / 32) * pitch + col0 + (lane % 32) * 8;
    const int64_t w_step = (int64_t)16 * 4 * K, x_step = (int64_t)2 * 4 * pitch;
    int is_ch = 0, is_stage = 0, issued = 0;
    const unsigned short *srcx_cur = srcx;
    auto issue_chunk = [&]() {
        const unsigned st = lds0 + is_stage * STAGE + (wave & 3) * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (MODE & 1) dma16(srcw + (j * w_step + is_ch * KC), st + j * 4096);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (MODE & 2) dma16(srcx_cur + (j * x_step + (int64_t)is_ch * KC * pitch), st + 16384 + j * 4096);
        if (++is_ch == nch) {
            is_ch = 0;
            srcx_cur += TN;
        }
        if (++is_stage == NS) is_stage = 0;
        if (++issued == T) {
            is_ch = 0;
            srcx_cur = srcx;
        }
    };
    constexpr int PIECES = ((MODE & 1) ? 4 : 0) + ((MODE & 2) ? 4 : 0);
    if (loader) {
        for (int c = 0; c < NS - 1; ++c) issue_chunk();
        wait_vm<2 * PIECES>();
    }
    __builtin_amdgcn_s_barrier();
    // stores: a tile's 256 rows x 512 B = 128 x 1-KiB instructions per workgroup; spread over the tile's chunks
    const u32x4 val = {(unsigned)tid, 1u, 2u, 3u};
    const int store_waves = (MODE & 8) ? WAVES - 4 : 4;
    const int sw = (MODE & 8) ? wave - 4 : wave;                         // index among the storing waves
    const int per_chunk = 128 / nch / store_waves;                       // store instructions per storing wave and chunk
    int ch = 0, tile = 0;
    for (int t = 0; t < T; ++t) {
        for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(8);      // ~64 x 8 cycles each
        if (loader) {
            // everything but chunk t + 2's pieces (and, when this wave also stores, the stores issued since) may be in flight
            // (in order on gfx950: behind chunk t + 1's pieces lie S(t-2), chunk t + 2's pieces, S(t-1))
            if ((MODE & 4)) {
                if (per_chunk == 4) wait_vm<PIECES + 8>();
                else if (per_chunk == 2) wait_vm<PIECES + 4>();
                else if (per_chunk == 1) wait_vm<PIECES + 2>();
                else wait_vm<PIECES + 16>();
            } else wait_vm<PIECES>();
        }
        __builtin_amdgcn_s_barrier();
        if (loader) issue_chunk();
        const bool storer = (MODE & 8) ? !loader : ((MODE & 4) != 0);
        if (storer && tile > 0) {
            // previous tile's outputs: instruction i of the tile = 4 rows x 256 B of one 128-column half (gemm_xl.h's row stores)
            for (int s = 0; s < per_chunk; ++s) {
                const int i = (ch * per_chunk + s) * store_waves + sw;   // 0 .. 127
                const int row = r0 + 4 * (i >> 1) + (lane >> 4);
                unsigned short *dst = Y + (size_t)row * pitch + col0 + (int64_t)(tile - 1) * TN + (i & 1) * 128 + (lane & 15) * 8;
                __builtin_nontemporal_store(val, reinterpret_cast<u32x4 *>(dst));
            }
        }
        if (++ch == nch) {
            ch = 0;
            ++tile;
        }
    }
    {   // the last tile's stores (nothing left to overlap with)
        const bool storer = (MODE & 8) ? !loader : ((MODE & 4) != 0);
        if (storer)
            for (int s = 0; s < per_chunk * nch; ++s) {
                const int i = s * store_waves + sw;
                const int row = r0 + 4 * (i >> 1) + (lane >> 4);
                unsigned short *dst = Y + (size_t)row * pitch + col0 + (int64_t)(tiles_range - 1) * TN + (i & 1) * 128 + (lane & 15) * 8;
                __builtin_nontemporal_store(val, reinterpret_cast<u32x4 *>(dst));
            }
    }
    wait_vm<0>();
}

template <int MODE, int WAVES>
static float run(const unsigned short *A, const unsigned short *X, unsigned short *Y, int64_t pitch, int K, int R, int64_t M, int delay,
                 int reps) {
    const int row_tiles = R / TR, col_tiles = (int)(M / TN), tiles_range = 8;
    const int nblocks = row_tiles * (col_tiles / tiles_range);
    auto fn = traffic<MODE, WAVES>;
    hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, NS * STAGE + 32768);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(nblocks), dim3(WAVES * 64), NS * STAGE + 32768, 0, A, X, Y, pitch, K, row_tiles, tiles_range, nblocks, delay);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(fn, dim3(nblocks), dim3(WAVES * 64), NS * STAGE + 32768, 0, A, X, Y, pitch, K, row_tiles, tiles_range, nblocks,
                           delay);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main(int argc, char **argv) {
    const int R = 1024, K = argc > 1 ? atoi(argv[1]) : 256;
    const int64_t M = 524288;
    const int64_t pad = 512;                                      // elements: 1 KiB more per row
    unsigned short *A, *X, *Y;
    hipMalloc(&A, (size_t)R * K * 2);
    hipMalloc(&X, (size_t)K * (M + pad) * 2);
    hipMalloc(&Y, (size_t)R * (M + pad) * 2);
    hipMemset(A, 1, (size_t)R * K * 2);
    hipMemset(X, 1, (size_t)K * (M + pad) * 2);
    hipMemset(Y, 0, (size_t)R * (M + pad) * 2);
    const double w_b = (double)R / TR * (M / TN) * (K / KC) * 16384.0, x_b = w_b, y_b = (double)R * M * 2.0;
    printf("# product %d x %d x %lld on 256 x 256 tiles: W pieces %.2f GB, X pieces %.2f GB (%.2f GB unique), stores %.2f GB per launch\n", R, K,
           (long long)M, w_b / 1e9, x_b / 1e9, (double)K * M * 2 / 1e9, y_b / 1e9);
    for (int delay = 0; delay <= 2; ++delay) {
        printf("delay per chunk: %d x s_sleep 8 (~%d cycles)\n", delay, delay * 512);
        const int reps = 10;
#define RUN(mode, waves, what) printf("  %-62s %8.1f us\n", what, run<mode, waves>(A, X, Y, M, K, R, M, delay, reps))
#define RUNP(mode, waves, what) printf("  %-62s %8.1f us\n", what, run<mode, waves>(A, X, Y, M + pad, K, R, M, delay, reps))
        RUN(1, 4, "W pieces only (L2 hits)");
        RUN(2, 4, "X pieces only (1 HBM read + 3 L2 hits)");
        RUNP(2, 4, "X pieces only, row pitch M + 512");
        RUN(3, 4, "W + X pieces");
        RUN(4, 4, "stores only (four waves)");
        RUN(8, 8, "stores only (four extra waves)");
        RUN(7, 4, "W + X + stores, all from the same four waves");
        RUN(11, 8, "W + X from four waves, stores from four MORE waves");
        RUNP(11, 8, "the same, row pitch M + 512");
    }
    return 0;
}
