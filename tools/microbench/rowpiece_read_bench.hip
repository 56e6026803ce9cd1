// rowpiece_read_bench.hip -- what does the READ side of a split-K kernel reach on MI355X?
//
// The weight gradient (wgrad.hip) and the read-heavy GEMM shapes stream a (rows x M) bf16 matrix as chunks of
// ROWS x PIECE bytes -- ROWS row pieces that are M * 2 bytes apart -- by LDS-DMA into a ring of NS stages.  They read at
// 3.5-4 TB/s where a linear read reaches 5-6.6.  This kernel is that loop with the arithmetic removed: which of piece
// size, rows per chunk, ring depth and workgroups per CU moves the number?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 rowpiece_read_bench.hip -o rowpiece_read_bench && ./rowpiece_read_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_base) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int PER> __device__ __forceinline__ void wait_ahead(int ahead) {      // `ahead` newer chunks may stay in flight
    switch (ahead) {
    case 0: wait_vm<0>(); break;
    case 1: wait_vm<(PER < 64 ? PER : 63)>(); break;
    case 2: wait_vm<(2 * PER < 64 ? 2 * PER : 63)>(); break;
    case 3: wait_vm<(3 * PER < 64 ? 3 * PER : 63)>(); break;
    case 4: wait_vm<(4 * PER < 64 ? 4 * PER : 63)>(); break;
    case 5: wait_vm<(5 * PER < 64 ? 5 * PER : 63)>(); break;
    case 6: wait_vm<(6 * PER < 64 ? 6 * PER : 63)>(); break;
    default: wait_vm<(7 * PER < 64 ? 7 * PER : 63)>(); break;
    }
}

// one workgroup = one slice of `cols` columns of all `ROWS` rows starting at row0 = blockIdx.y * ROWS
template <int ROWS, int PIECE, int NS, int NW>
__global__ __launch_bounds__(64 * NW) void ring_read(const unsigned short *__restrict__ X, int64_t M, int64_t cols,
                                                     unsigned *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int STAGE = ROWS * PIECE, SLOTS = PIECE / 16, RPD = 1024 / PIECE, PER = STAGE / 1024 / NW, D = NS - 1;
    static_assert(STAGE % (1024 * NW) == 0, "whole DMA instructions per wave");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(void __attribute__((address_space(3))) *)smem;
    const unsigned short *base = X + (size_t)blockIdx.y * ROWS * M + (size_t)blockIdx.x * cols;
    const unsigned short *src[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int row = RPD * (PER * wave + j) + lane / SLOTS;
        src[j] = base + (size_t)row * M + (lane % SLOTS) * 8;
    }
    const int T = (int)(cols * 2 / PIECE);
    auto issue = [&](int t) {
#pragma unroll
        for (int j = 0; j < PER; ++j) dma16(src[j] + (size_t)t * (PIECE / 2), lds0 + (t % NS) * STAGE + (PER * wave + j) * 1024);
    };
#pragma unroll
    for (int c = 0; c < D; ++c)
        if (c < T) issue(c);
    unsigned acc = 0;
    for (int t = 0; t < T; ++t) {
        const int ahead = (T - 1 - t < D - 1) ? T - 1 - t : D - 1;
        wait_ahead<PER>(__builtin_amdgcn_readfirstlane(ahead));
        __builtin_amdgcn_s_barrier();
        if (t + D < T) issue(t + D);
        acc ^= *reinterpret_cast<const unsigned *>(smem + (t % NS) * STAGE + tid * 4);   // one token LDS read per chunk
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// the same traffic through registers: a thread loads 16 bytes of UNROLL consecutive chunks before using any
template <int ROWS, int PIECE, int UNROLL>
__global__ __launch_bounds__(256) void reg_read(const unsigned short *__restrict__ X, int64_t M, int64_t cols,
                                                unsigned *__restrict__ sink) {
    constexpr int SLOTS = PIECE / 16, PER = ROWS * SLOTS / 256;
    const int tid = threadIdx.x;
    const unsigned short *base = X + (size_t)blockIdx.y * ROWS * M + (size_t)blockIdx.x * cols;
    const int T = (int)(cols * 2 / PIECE);
    uint4 a = make_uint4(0, 0, 0, 0);
    for (int t0 = 0; t0 < T; t0 += UNROLL) {
        uint4 v[UNROLL][PER];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int p = tid + 256 * j, row = p / SLOTS, slot = p % SLOTS;
                if (t0 + u >= T) { v[u][j] = make_uint4(0, 0, 0, 0); continue; }
                v[u][j] = *reinterpret_cast<const uint4 *>(base + (size_t)row * M + (size_t)(t0 + u) * (PIECE / 2) + slot * 8);
            }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int j = 0; j < PER; ++j) { a.x ^= v[u][j].x; a.y ^= v[u][j].y; a.z ^= v[u][j].z; a.w ^= v[u][j].w; }
    }
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345678u) sink[0] = a.x;
}

// MFMA-fragment order straight from global memory: lane (r = lane & 31, h = lane >> 5) of wave w owns row 32 w + r and
// loads the 64 bytes [64 h, 64 h + 64) of its row's 128-byte piece as four 16-byte loads (the four k-steps of a chunk);
// UNROLL chunks are requested before any is used.  NW waves per workgroup = 32 NW rows.
template <int NW, int UNROLL>
__global__ __launch_bounds__(64 * NW) void frag_read(const unsigned short *__restrict__ X, int64_t M, int64_t cols,
                                                     unsigned *__restrict__ sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned short *src = X + ((size_t)blockIdx.y * 32 * NW + wave * 32 + (lane & 31)) * M + (size_t)blockIdx.x * cols + (lane >> 5) * 32;
    const int T = (int)(cols / 64);
    uint4 a = make_uint4(0, 0, 0, 0);
    for (int t0 = 0; t0 < T; t0 += UNROLL) {
        uint4 v[UNROLL][4];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (t0 + u >= T) { v[u][j] = make_uint4(0, 0, 0, 0); continue; }
                v[u][j] = *reinterpret_cast<const uint4 *>(src + (size_t)(t0 + u) * 64 + j * 8);
            }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) { a.x ^= v[u][j].x; a.y ^= v[u][j].y; a.z ^= v[u][j].z; a.w ^= v[u][j].w; }
    }
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345678u) sink[0] = a.x;
}
template <int NW, int UNROLL> void run_frag(const unsigned short *X, int rows, int64_t M, int slices, unsigned *sink) {
    const int64_t cols = M / slices;
    float best = 1e9;
    hipEvent_t s, e;
    (void)hipEventCreate(&s);
    (void)hipEventCreate(&e);
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(s);
        hipLaunchKernelGGL((frag_read<NW, UNROLL>), dim3(slices, rows / (32 * NW)), dim3(64 * NW), 0, 0, X, M, cols, sink);
        (void)hipEventRecord(e);
        (void)hipEventSynchronize(e);
        float ms;
        (void)hipEventElapsedTime(&ms, s, e);
        if (rep && ms < best) best = ms;
    }
    printf("  frag waves %d (rows %3d) unroll %d slices %4d : %.2f TB/s\n", NW, 32 * NW, UNROLL, slices, (double)rows * M * 2 / best / 1e9);
}

template <typename F> float timeit(F f) {
    hipEvent_t s, e;
    hipEventCreate(&s);
    hipEventCreate(&e);
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(s);
        f();
        hipEventRecord(e);
        hipEventSynchronize(e);
        float ms;
        hipEventElapsedTime(&ms, s, e);
        if (rep && ms < best) best = ms;
    }
    return best;
}

template <int ROWS, int PIECE, int NS, int NW> void run_ring(const unsigned short *X, int rows, int64_t M, int slices, unsigned *sink) {
    const int64_t cols = M / slices;
    const size_t lds = (size_t)NS * ROWS * PIECE;
    (void)hipFuncSetAttribute((const void *)ring_read<ROWS, PIECE, NS, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const float ms = timeit([&] {
        hipLaunchKernelGGL((ring_read<ROWS, PIECE, NS, NW>), dim3(slices, rows / ROWS), dim3(64 * NW), lds, 0, X, M, cols, sink);
    });
    printf("  ring rows %3d piece %3d B stages %d waves %d (%3zu KB LDS) slices %4d : %.2f TB/s\n", ROWS, PIECE, NS, NW,
           lds >> 10, slices, (double)rows * M * 2 / ms / 1e9);
}
template <int ROWS, int PIECE, int UNROLL> void run_reg(const unsigned short *X, int rows, int64_t M, int slices, unsigned *sink) {
    const int64_t cols = M / slices;
    const float ms = timeit([&] {
        hipLaunchKernelGGL((reg_read<ROWS, PIECE, UNROLL>), dim3(slices, rows / ROWS), dim3(256), 0, 0, X, M, cols, sink);
    });
    printf("  regs rows %3d piece %3d B unroll %d slices %4d : %.2f TB/s\n", ROWS, PIECE, UNROLL, slices,
           (double)rows * M * 2 / ms / 1e9);
}

int main() {
    unsigned *sink;
    setvbuf(stdout, nullptr, _IONBF, 0);
    (void)hipMalloc(&sink, 64);
    for (int64_t M : {(int64_t)1 << 19, (int64_t)1 << 21}) {
        const int rows = M == (1 << 19) ? 1024 : 512;                  // 1 GB / 2 GB
        unsigned short *X;
        (void)hipMalloc(&X, (size_t)rows * M * 2);
        (void)hipMemset(X, 1, (size_t)rows * M * 2);
        printf("matrix %d rows x %lld columns bf16 (row stride %lld KB)\n", rows, (long long)M, (long long)(M * 2 >> 10));
        for (int slices : {512, 1024, 2048}) {
            run_ring<128, 128, 4, 4>(X, rows, M, slices, sink);        // the T tile of wgrad.hip
            run_ring<128, 128, 2, 4>(X, rows, M, slices, sink);
            run_ring<128, 256, 2, 4>(X, rows, M, slices, sink);
            run_ring<128, 256, 3, 4>(X, rows, M, slices, sink);
            run_ring<128, 512, 2, 4>(X, rows, M, slices, sink);
            run_ring<128, 64, 8, 4>(X, rows, M, slices, sink);
            run_ring<64, 256, 4, 4>(X, rows, M, slices, sink);
            run_ring<32, 256, 4, 4>(X, rows, M, slices, sink);         // the X tile of gemm.hip
            run_ring<32, 512, 4, 4>(X, rows, M, slices, sink);
            run_ring<256, 128, 2, 4>(X, rows, M, slices, sink);        // S
            run_ring<256, 64, 5, 4>(X, rows, M, slices, sink);         // S32
            run_reg<128, 128, 2>(X, rows, M, slices, sink);
            run_reg<128, 128, 4>(X, rows, M, slices, sink);
            run_reg<128, 256, 2>(X, rows, M, slices, sink);
            run_reg<32, 256, 8>(X, rows, M, slices, sink);
        }
        for (int slices : {256, 512, 1024}) {
            run_frag<8, 1>(X, rows, M, slices, sink);
            run_frag<8, 2>(X, rows, M, slices, sink);
            run_frag<8, 3>(X, rows, M, slices, sink);
            run_frag<4, 2>(X, rows, M, slices, sink);
            run_frag<4, 4>(X, rows, M, slices, sink);
        }
        for (int slices : {256, 512}) {
            run_ring<512, 128, 2, 8>(X, rows, M, slices, sink);        // L
            run_ring<512, 64, 4, 8>(X, rows, M, slices, sink);         // L32
            run_ring<256, 128, 4, 8>(X, rows, M, slices, sink);
        }
        (void)hipFree(X);
    }
    return 0;
}
