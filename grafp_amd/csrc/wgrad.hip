// wgrad.hip -- weight gradient of a 1x1 convolution on the (C, M = B*N) layout, gfx950.
//
//   dW[o][c] = sum_m G[o][m] * X[c][m]        G = dL/dY (Cout x M), X = layer input (Cin x M), both bf16, rows
//                                             contiguous; M = B*N = 262 144 ... 32 768 at B = 256
// This is the backward of every 1x1 Conv2d of the encoder (/root/reference/encoder/gcn_lib/torch_vertex.py:152-162,
// torch_nn.py:56, encoder/graph_encoder.py:52-55,131,156) w.r.t. its weight.  The output is tiny (64x64 ... 2048x512)
// and the contraction is enormous, with BOTH operands contiguous along the contraction: the library GEMM reaches
// ~10 TFLOP/s here (80-210 us where the operands stream in 13-34 us).  It is a streaming reduction:
//   * split-K: a block owns a slice of M and a 64x64 (128x128) output tile, blockIdx.z a conv group; blocks are
//     numbered so that the tiles of one slice run next to each other on one XCD;
//   * 128-column chunks of the G and X tiles go HBM -> registers (one chunk ahead of the MFMAs) -> LDS with
//     coalesced 16-byte loads;
//   * each of the 4 waves owns a 32x32 quadrant: A and B fragments of v_mfma_f32_32x32x16_bf16 are 16 contiguous
//     bytes of a G / X row (8 consecutive m), read with ds_read_b128 from rows padded to 272 B (conflict-free);
//   * partial tiles go to a (S, Cout, Cin/g) f32 scratch, summed by wgrad_reduce_kernel (deterministic, no atomics).
// HBM-bound: (Cout + Cin) * M * 2 bytes per launch (+ re-reads of the smaller operand across output tiles).
#include "common.h"
#include "dma_ring.h"
#include "tuning.h"

namespace grafp {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int WG_KC = 128;               // contraction columns per LDS chunk
constexpr int WG_LS = WG_KC * 2 + 16;    // LDS row stride in bytes (272: ds_read_b128 rows land on distinct bank quads)

// TW = output tile edge per workgroup (64: one 32x32 MFMA tile per wave; 128: 2x2 tiles per wave, which halves
// the re-reads of the smaller operand for the large outputs).
template <int TW>
__global__ __launch_bounds__(256) void wgrad_partial_kernel(const unsigned short *__restrict__ G,
                                                            const unsigned short *__restrict__ X, int64_t M,
                                                            int cout_g, int cin_g, int tiles_o, int tiles_c,
                                                            int64_t cols_per_split,
                                                            float *__restrict__ part) {
    constexpr int Q = TW / 2;        // quadrant edge per wave
    constexpr int NT = Q / 32;       // MFMA tiles per quadrant edge
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sG = smem, *sX = smem + TW * WG_LS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    // one XCD (= one L2) owns a contiguous range of logical blocks, and the output tiles of one K-slice are
    // consecutive logical blocks: tiles that re-read the same G rows / X rows of a slice hit that L2
    const int ntiles = tiles_o * tiles_c;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / ntiles, tile = logical - split * ntiles, grp = blockIdx.z;
    const int o0 = (tile / tiles_c) * TW, c0 = (tile % tiles_c) * TW;
    const unsigned short *Gg = G + (size_t)grp * cout_g * M;
    const unsigned short *Xg = X + (size_t)grp * cin_g * M;
    const int64_t m_begin = (int64_t)split * cols_per_split;
    const int64_t m_end = (m_begin + cols_per_split < M) ? m_begin + cols_per_split : M;
    const int wo = wave >> 1, wc = wave & 1;

    f32x16 acc[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const bool vec_ok = (M & 7) == 0;
    // chunk m0 of both operands -> registers: thread -> (row = i*16 + tid/16, 16-byte column tid%16)
    uint4 pg[TW / 16], px[TW / 16];
    auto fetch = [&](int64_t m0) {
#pragma unroll
        for (int i = 0; i < TW / 16; ++i) {
            const int row = i * 16 + (tid >> 4), cb = tid & 15;
            const int64_t m = m0 + cb * 8;
            uint4 vg = make_uint4(0, 0, 0, 0), vx = make_uint4(0, 0, 0, 0);
            if (vec_ok && m + 8 <= m_end) {
                if (o0 + row < cout_g) vg = *reinterpret_cast<const uint4 *>(Gg + (size_t)(o0 + row) * M + m);
                if (c0 + row < cin_g) vx = *reinterpret_cast<const uint4 *>(Xg + (size_t)(c0 + row) * M + m);
            } else if (m < m_end) {
                unsigned short tg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int e = 0; e < 8 && m + e < m_end; ++e) {
                    if (o0 + row < cout_g) tg[e] = Gg[(size_t)(o0 + row) * M + m + e];
                    if (c0 + row < cin_g) tx[e] = Xg[(size_t)(c0 + row) * M + m + e];
                }
                vg = make_uint4(tg[0] | (tg[1] << 16), tg[2] | (tg[3] << 16), tg[4] | (tg[5] << 16), tg[6] | (tg[7] << 16));
                vx = make_uint4(tx[0] | (tx[1] << 16), tx[2] | (tx[3] << 16), tx[4] | (tx[5] << 16), tx[6] | (tx[7] << 16));
            }
            pg[i] = vg;
            px[i] = vx;
        }
    };
    if (m_begin < m_end) fetch(m_begin);
    for (int64_t m0 = m_begin; m0 < m_end; m0 += WG_KC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TW / 16; ++i) {
            const int row = i * 16 + (tid >> 4), cb = tid & 15;
            *reinterpret_cast<uint4 *>(sG + row * WG_LS + cb * 16) = pg[i];
            *reinterpret_cast<uint4 *>(sX + row * WG_LS + cb * 16) = px[i];
        }
        __syncthreads();
        if (m0 + WG_KC < m_end) fetch(m0 + WG_KC);      // in flight while this chunk is multiplied
        const unsigned char *ga = sG + (wo * Q + l31) * WG_LS + half * 16;
        const unsigned char *xa = sX + (wc * Q + l31) * WG_LS + half * 16;
#pragma unroll
        for (int kk = 0; kk < WG_KC / 16; ++kk) {
            bf16x8 av[NT], bv[NT];
#pragma unroll
            for (int a = 0; a < NT; ++a) av[a] = *reinterpret_cast<const bf16x8 *>(ga + a * 32 * WG_LS + kk * 32);
#pragma unroll
            for (int b = 0; b < NT; ++b) bv[b] = *reinterpret_cast<const bf16x8 *>(xa + b * 32 * WG_LS + kk * 32);
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
    }
    // partial tile -> part[split][grp][o][c]
    float *pp = part + ((size_t)split * gridDim.z + grp) * cout_g * cin_g;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int c = c0 + wc * Q + b * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + wo * Q + a * 32 + mfma_row(r, half);
                if (o < cout_g && c < cin_g) pp[(size_t)o * cin_g + c] = acc[a][b][r];
            }
        }
}

// f32 operands (the f32 "parity" mode of the training step): same streaming reduction, with every operand value split
// on the way into LDS into hi = bf16(v) and lo = bf16(v - hi), and three bf16 MFMAs per tile step
//   G X^T ~= Gh Xh^T + Gh Xl^T + Gl Xh^T     (products of bf16 pairs are exact in the f32 accumulator; the dropped
//                                             Gl Xl^T term and the 16-bit representation are ~2^-16 relative)
// -- 3 matrix instructions at the bf16 rate (16x the f32 MFMA rate) instead of the library's f32 GEMM, which runs
// this tall-K shape at 42 ms per step.  64 x 64 tiles only (four LDS planes of 64 rows).
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
// two values -> (packed hi pair, packed lo pair): v_cvt_pk_bf16_f32 (round to nearest even), exact residual, again
__device__ __forceinline__ void wg_split2(float a, float b, unsigned &hi, unsigned &lo) {
    const wg_f32x2 v = {a, b};
    const wg_bf16x2 h = __builtin_convertvector(v, wg_bf16x2);
    const wg_bf16x2 l = __builtin_convertvector(v - __builtin_convertvector(h, wg_f32x2), wg_bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void wg_split8(const float4 &a, const float4 &b, uint4 &hi, uint4 &lo) {
    wg_split2(a.x, a.y, hi.x, lo.x);
    wg_split2(a.z, a.w, hi.y, lo.y);
    wg_split2(b.x, b.y, hi.z, lo.z);
    wg_split2(b.z, b.w, hi.w, lo.w);
}

__global__ __launch_bounds__(256) void wgrad3_partial_kernel(const float *__restrict__ G, const float *__restrict__ X,
                                                             int64_t M, int cout_g, int cin_g, int tiles_o, int tiles_c,
                                                             int64_t cols_per_split, float *__restrict__ part) {
    constexpr int TW = 64, PL = TW * WG_LS;       // one LDS plane: 64 rows
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sGh = smem, *sGl = smem + PL, *sXh = smem + 2 * PL, *sXl = smem + 3 * PL;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int ntiles = tiles_o * tiles_c;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / ntiles, tile = logical - split * ntiles, grp = blockIdx.z;
    const int o0 = (tile / tiles_c) * TW, c0 = (tile % tiles_c) * TW;
    const float *Gg = G + (size_t)grp * cout_g * M;
    const float *Xg = X + (size_t)grp * cin_g * M;
    const int64_t m_begin = (int64_t)split * cols_per_split;
    const int64_t m_end = (m_begin + cols_per_split < M) ? m_begin + cols_per_split : M;
    const int wo = wave >> 1, wc = wave & 1;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const bool vec_ok = (M & 3) == 0;
    float4 pg[TW / 16][2], px[TW / 16][2];
    auto ld8 = [&](const float *rowp, int64_t m, float4 (&d)[2]) {
        if (vec_ok && m + 8 <= m_end) {
            d[0] = *reinterpret_cast<const float4 *>(rowp + m);
            d[1] = *reinterpret_cast<const float4 *>(rowp + m + 4);
        } else {
            float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int e = 0; e < 8 && m + e < m_end; ++e) t[e] = rowp[m + e];
            d[0] = make_float4(t[0], t[1], t[2], t[3]);
            d[1] = make_float4(t[4], t[5], t[6], t[7]);
        }
    };
    auto fetch = [&](int64_t m0) {
#pragma unroll
        for (int i = 0; i < TW / 16; ++i) {
            const int row = i * 16 + (tid >> 4), cb = tid & 15;
            const int64_t m = m0 + cb * 8;
            pg[i][0] = pg[i][1] = px[i][0] = px[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_end) {
                if (o0 + row < cout_g) ld8(Gg + (size_t)(o0 + row) * M, m, pg[i]);
                if (c0 + row < cin_g) ld8(Xg + (size_t)(c0 + row) * M, m, px[i]);
            }
        }
    };
    if (m_begin < m_end) fetch(m_begin);
    for (int64_t m0 = m_begin; m0 < m_end; m0 += WG_KC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TW / 16; ++i) {
            const int row = i * 16 + (tid >> 4), cb = tid & 15;
            uint4 h, l;
            wg_split8(pg[i][0], pg[i][1], h, l);
            *reinterpret_cast<uint4 *>(sGh + row * WG_LS + cb * 16) = h;
            *reinterpret_cast<uint4 *>(sGl + row * WG_LS + cb * 16) = l;
            wg_split8(px[i][0], px[i][1], h, l);
            *reinterpret_cast<uint4 *>(sXh + row * WG_LS + cb * 16) = h;
            *reinterpret_cast<uint4 *>(sXl + row * WG_LS + cb * 16) = l;
        }
        __syncthreads();
        if (m0 + WG_KC < m_end) fetch(m0 + WG_KC);
        const int ga = (wo * 32 + l31) * WG_LS + half * 16, xa = (wc * 32 + l31) * WG_LS + half * 16;
#pragma unroll
        for (int kk = 0; kk < WG_KC / 16; ++kk) {
            const bf16x8 gh = *reinterpret_cast<const bf16x8 *>(sGh + ga + kk * 32);
            const bf16x8 gl = *reinterpret_cast<const bf16x8 *>(sGl + ga + kk * 32);
            const bf16x8 xh = *reinterpret_cast<const bf16x8 *>(sXh + xa + kk * 32);
            const bf16x8 xl = *reinterpret_cast<const bf16x8 *>(sXl + xa + kk * 32);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gl, xh, acc, 0, 0, 0);     // small terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, xl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, xh, acc, 0, 0, 0);
        }
    }
    float *pp = part + ((size_t)split * gridDim.z + grp) * cout_g * cin_g;
    const int c = c0 + wc * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = o0 + wo * 32 + mfma_row(r, half);
        if (o < cout_g && c < cin_g) pp[(size_t)o * cin_g + c] = acc[r];
    }
}

// ---- the LDS-DMA form (every shape of the encoder: rows per group % 32 == 0, M per view % 64 == 0) ---------------------
// Same split-K streaming reduction, restructured like gemm.hip: 64-column chunks of the G tile (TO rows x 128 B) and the
// X tile (TC rows x 128 B) arrive by global_load_lds_dwordx4 into a ring of NS stages with NS-1 chunks in flight across
// ONE raw s_barrier per chunk and counted vmcnt (no staging registers, no ds_write pass, no second barrier -- the
// register-staged kernel above spends 119 of 128 us on loads + LDS writes + barriers on the wide shapes).  Rows are
// 128 B; their eight 16-byte slots are XOR-swizzled by (row >> 1) & 7 on the DMA source side and on the ds_read_b128
// side: a 16-lane read group covers 16 distinct (row & 1, (row >> 1) & 7) pairs = all 64 banks once.
//   T: 2 x 2 waves x (1 x 1) MFMA tiles ->  64 x  64 outputs, 4 stages, 2 workgroups per CU (fc1/gconv of stages 0-1)
//   S: 2 x 2 waves x (2 x 2)            -> 128 x 128 outputs, 2 stages, 2 workgroups per CU
//   L: 2 x 4 waves x (4 x 2)            -> 256 x 256 outputs, 2 stages, 1 workgroup per CU (128 flop per DMA byte)
// PRO: X is the RAW output of the previous layer's convolution; its BatchNorm + activation (per X row: scale, shift of
// the slice's view) is applied to the X tile in LDS, so the normalised activation the forward pass never wrote is not
// needed here either (grafp_conv1x1_gemm_bf16's pro_tab, same table).
template <int WR_, int WM_, int RT_, int CT_, int NS_, int KC_ = 64> struct WgCfg {
    static constexpr int WR = WR_, WM = WM_, RT = RT_, CT = CT_, NS = NS_, NW = WR_ * WM_, THREADS = 64 * NW;
    static constexpr int TO = WR_ * RT_ * 32, TC = WM_ * CT_ * 32;
    static constexpr int KC = KC_;                                 // contraction (m) per chunk: 128- or 64-byte rows
    static constexpr int ROWB = KC_ * 2, SLOTS = KC_ / 8;          // bytes and 16-byte slots per tile row
    static constexpr int RPD = 1024 / ROWB;                        // tile rows one DMA instruction (64 x 16 B) covers
    static constexpr int RP256 = 256 / ROWB;                       // tile rows per 256 B = per pass over the 64 banks
    static constexpr int G_BYTES = TO * ROWB, X_BYTES = TC * ROWB, STAGE = G_BYTES + X_BYTES;
    static constexpr int G_DMA = TO / RPD / NW, X_DMA = TC / RPD / NW;   // DMA instructions per wave and chunk
    static_assert(KC_ == 128 || KC_ == 64 || KC_ == 32, "row pieces of 256, 128 or 64 bytes");
    static_assert(TO % (RPD * NW) == 0 && TC % (RPD * NW) == 0, "whole DMA instructions per wave");
    // slot s of row r lives at slot s ^ swz(r): a 16-lane ds_read_b128 group (16 consecutive rows, one slot) then
    // covers every bank exactly once -- 128-byte rows: 2 rows x 8 slots, 64-byte rows: 4 rows x 4 slots per 256 B
    static __device__ __forceinline__ int swz(int row) { return (row / RP256) & (SLOTS - 1); }
};
typedef WgCfg<2, 2, 1, 1, 4> WgT;
// 256-byte row pieces for the layers whose rows lie megabytes apart (stage 0-1 at 1024 pairs per GPU): at a 4 MB row
// stride LDS-DMA pieces of 128 bytes stream at 4.1-4.7 TB/s, 256-byte pieces at 4.9-6.3 (rowpiece_read_bench) -- the
// wave's time goes into ISSUING the DMA instructions against the back-pressure of the memory pipeline (s_memtime: 2 100
// of 3 100 cycles per chunk), so ring depth does not matter there (3, 4, 5 stages, 8 x 64-byte: equal) and the piece does
typedef WgCfg<2, 2, 1, 1, 2, 128> WgT128;                          // 64 x 64, 2 x 32 KB, 2 workgroups per CU
typedef WgCfg<2, 2, 2, 2, 2, 128> WgS128;                          // 128 x 128, 2 x 64 KB, 1 workgroup per CU
typedef WgCfg<2, 2, 2, 2, 2> WgS;
typedef WgCfg<2, 4, 4, 2, 2> WgL;
// 64-byte row pieces: the same LDS holds twice as many chunks, i.e. twice the loads in flight per workgroup -- what the
// wide tiles need (their operand re-reads come from L2 at the rate of bytes-in-flight / ~1.8 us)
typedef WgCfg<2, 2, 2, 2, 5, 32> WgS32;                            // 128 x 128, 5 x 16 KB, 2 workgroups per CU
typedef WgCfg<4, 2, 2, 2, 3, 32> WgM32;                            // 256 x 128, 8 waves, 3 x 24 KB, 2 workgroups per CU
typedef WgCfg<2, 4, 4, 2, 4, 32> WgL32;                            // 256 x 256, 8 waves, 4 x 32 KB, 1 workgroup per CU
typedef WgCfg<2, 2, 2, 2, 3> WgSG;                                 // wgrad_gr_kernel: 128 x 128, 80 KB, 2 workgroups per CU
typedef WgCfg<2, 4, 4, 2, 3> WgLG;                                 // wgrad_gr_kernel: 256 x 256, 160 KB, 1 workgroup per CU

template <typename CFG, bool PRO>
__global__ __launch_bounds__(CFG::THREADS) void wgrad_dma_kernel(
    const unsigned short *__restrict__ G, const unsigned short *__restrict__ X, int64_t M, int cout_g, int cin_g,
    int tiles_o, int tiles_c, int slices_view, int64_t cols_per_slice, int views, const float2 *__restrict__ pro_tab,
    int pro_act, float pro_slope, float *__restrict__ part, int nblocks) {
    constexpr int NS = CFG::NS, D = NS - 1, RT = CFG::RT, CT = CFG::CT, KC = CFG::KC;
    constexpr int DMA_PER_CHUNK = CFG::G_DMA + CFG::X_DMA;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *const s_tab = smem + NS * CFG::STAGE;            // PRO: TC x float2

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave / CFG::WM, wc = wave % CFG::WM;
    const unsigned lds0 = (unsigned)(uintptr_t)(gm_lptr)smem;
    const int ntiles = tiles_o * tiles_c;
    const int logical = xcd_remap(blockIdx.x, nblocks);
    const int slice = logical / ntiles, tile = logical - slice * ntiles, grp = blockIdx.z;
    const int view = slice / slices_view, sl = slice - view * slices_view;
    const int o0 = (tile / tiles_c) * CFG::TO, c0 = (tile % tiles_c) * CFG::TC;
    const int64_t Mv = M / views;
    const int64_t m_begin = (int64_t)view * Mv + (int64_t)sl * cols_per_slice;
    int64_t m_len = Mv - (int64_t)sl * cols_per_slice;
    if (m_len > cols_per_slice) m_len = cols_per_slice;
    const int T = (int)(m_len / KC);
    const unsigned short *Gg = G + (size_t)grp * cout_g * M + m_begin;
    const unsigned short *Xg = X + (size_t)grp * cin_g * M + m_begin;

    if (PRO) {
        const float2 *src = pro_tab + ((size_t)grp * cin_g) * views;
        for (int r = tid; r < CFG::TC; r += CFG::THREADS) {
            const int c = c0 + r < cin_g ? c0 + r : cin_g - 1;
            reinterpret_cast<float2 *>(s_tab)[r] = src[(size_t)c * views + view];
        }
    }
    // DMA: instruction q covers tile rows RPD q .. RPD q + RPD - 1; lane -> row RPD q + lane / SLOTS, and its LDS slot
    // lane % SLOTS receives the source slot (lane % SLOTS) ^ swz(row).  Rows beyond the matrix re-read its last row
    // (their outputs are not stored).
    const unsigned short *g_src[CFG::G_DMA], *x_src[CFG::X_DMA];
#pragma unroll
    for (int j = 0; j < CFG::G_DMA; ++j) {
        const int row = CFG::RPD * (CFG::G_DMA * wave + j) + lane / CFG::SLOTS;
        int o = o0 + row;
        if (o > cout_g - 1) o = cout_g - 1;
        g_src[j] = Gg + (size_t)o * M + (((lane & (CFG::SLOTS - 1)) ^ CFG::swz(row)) << 3);
    }
#pragma unroll
    for (int j = 0; j < CFG::X_DMA; ++j) {
        const int row = CFG::RPD * (CFG::X_DMA * wave + j) + lane / CFG::SLOTS;
        int c = c0 + row;
        if (c > cin_g - 1) c = cin_g - 1;
        x_src[j] = Xg + (size_t)c * M + (((lane & (CFG::SLOTS - 1)) ^ CFG::swz(row)) << 3);
    }
    auto issue = [&](int t) {
        const unsigned st = lds0 + (t % NS) * CFG::STAGE;
#pragma unroll
        for (int j = 0; j < CFG::G_DMA; ++j) gm_dma16(g_src[j] + (size_t)t * KC, st + (CFG::G_DMA * wave + j) * 1024);
#pragma unroll
        for (int j = 0; j < CFG::X_DMA; ++j)
            gm_dma16(x_src[j] + (size_t)t * KC, st + CFG::G_BYTES + (CFG::X_DMA * wave + j) * 1024);
    };
    // fragment read offsets: row * ROWB + ((2 ks + half) ^ swz(row)) * 16; the XOR is applied per k-step below
    int goff[RT], gx[RT], xoff[CT], xx[CT];
#pragma unroll
    for (int a = 0; a < RT; ++a) {
        const int row = wo * 32 * RT + a * 32 + l31;
        goff[a] = row * CFG::ROWB;
        gx[a] = CFG::swz(row);
    }
#pragma unroll
    for (int b = 0; b < CT; ++b) {
        const int row = wc * 32 * CT + b * 32 + l31;
        xoff[b] = CFG::G_BYTES + row * CFG::ROWB;
        xx[b] = CFG::swz(row);
    }
    f32x16 acc[RT][CT];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    float2 ssb[CT];
#pragma unroll
    for (int b = 0; b < CT; ++b) ssb[b] = make_float2(1.0f, 0.0f);
    if (PRO) {
        __syncthreads();
#pragma unroll
        for (int b = 0; b < CT; ++b) ssb[b] = reinterpret_cast<const float2 *>(s_tab)[wc * 32 * CT + b * 32 + l31];
    }
#pragma unroll
    for (int c = 0; c < D; ++c)
        if (c < T) issue(c);
    for (int t = 0; t < T; ++t) {
        // chunks t .. min(t + D, T) - 1 are in flight: allow all but chunk t's
        const int ahead = (T - 1 - t < D - 1) ? T - 1 - t : D - 1;
        gm_wait_allowed(__builtin_amdgcn_readfirstlane(ahead * DMA_PER_CHUNK));
        __builtin_amdgcn_s_barrier();
        if (t + D < T) issue(t + D);
        unsigned char *const st = smem + (t % NS) * CFG::STAGE;
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            gm_bf16x8 av[RT], bv[CT];
#pragma unroll
            for (int a = 0; a < RT; ++a)
                av[a] = *reinterpret_cast<const gm_bf16x8 *>(st + goff[a] + (((2 * ks + half) ^ gx[a]) << 4));
#pragma unroll
            for (int b = 0; b < CT; ++b)
                bv[b] = *reinterpret_cast<const gm_bf16x8 *>(st + xoff[b] + (((2 * ks + half) ^ xx[b]) << 4));
            if (PRO) {
                // the fragment is 8 columns of ONE operand row: its (scale, shift) sits in two registers of the lane
                // (in registers, not as a pass over the staged tile: see conv1x1_gemm_kernel)
#pragma unroll
                for (int b = 0; b < CT; ++b) {
                    gm_u32x4 w = __builtin_bit_cast(gm_u32x4, bv[b]);
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        float lo = __uint_as_float(w[d] << 16), hi = __uint_as_float(w[d] & 0xffff0000u);
                        lo = __builtin_fmaf(lo, ssb[b].x, ssb[b].y);
                        hi = __builtin_fmaf(hi, ssb[b].x, ssb[b].y);
                        if (pro_act == 1) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
                        else if (pro_act == 2) { lo = lo > 0.f ? lo : lo * pro_slope; hi = hi > 0.f ? hi : hi * pro_slope; }
                        w[d] = gm_pack_bf16(lo, hi);
                    }
                    bv[b] = __builtin_bit_cast(gm_bf16x8, w);
                }
            }
#pragma unroll
            for (int a = 0; a < RT; ++a)
#pragma unroll
                for (int b = 0; b < CT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
    }
    // partial tile -> part[slice][grp][o][c]
    float *pp = part + ((size_t)slice * gridDim.z + grp) * cout_g * cin_g;
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b) {
            const int c = c0 + wc * 32 * CT + b * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + wo * 32 * RT + a * 32 + mfma_row(r, half);
                if (o < cout_g && c < cin_g) pp[(size_t)o * cin_g + c] = acc[a][b][r];
            }
        }
}

// ---- G through registers: twice the bytes in flight for the wide tiles -------------------------------------------------
// Measured (tools/microbench/rowpiece_read_bench.hip, tools/wgrad_sweep.sh): the wide tiles are bound by the RATE of their
// LDS-DMA -- 1.6x (256 x 256) to 3.2x (128 x 128) the unique bytes, the re-reads are L2 hits (PMC: FETCH_SIZE = unique
// bytes) -- and that rate is bytes-in-flight / latency: with 64 KB chunks only one fits the LDS beside the one being
// consumed.  The register file is the larger pool (512 KB per CU), so here the G tile takes the other road: every lane
// requests the 16 bytes the DMA would have delivered to its LDS slot (same rows, same source-side swizzle) into VGPRs,
// TWO chunks ahead, and writes them to one of two G stages when it has finished the chunk before; X stays on a 3-stage
// DMA ring.  Per workgroup 2 x (G + X) chunks are in flight instead of 1, with the same single barrier per chunk.
// vmcnt is counted by hand over both kinds (all issued through inline asm, in a fixed order: X DMA, then G loads).
// The staging registers are v224-v239 (set 0) and v240-v255 (set 1), named in the asm text: the kernel is compiled with
// amdgpu_num_vgpr(224), so the compiler never allocates them, and no C++ value stands for data that has been requested
// but not waited for.  (With ordinary "v" operands -- tied or pinned -- the register allocator copied the destination
// registers between the request and the s_waitcnt: v_mov of not-yet-loaded registers, seen in the .s.)
template <int SET> __device__ __forceinline__ void wg_gload4(const void *p0, const void *p1, const void *p2, const void *p3) {
    if constexpr (SET == 0)
        asm volatile("global_load_dwordx4 v[224:227], %0, off\n\tglobal_load_dwordx4 v[228:231], %1, off\n\t"
                     "global_load_dwordx4 v[232:235], %2, off\n\tglobal_load_dwordx4 v[236:239], %3, off"
                     :: "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                     : "memory", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234",
                       "v235", "v236", "v237", "v238", "v239");
    else
        asm volatile("global_load_dwordx4 v[240:243], %0, off\n\tglobal_load_dwordx4 v[244:247], %1, off\n\t"
                     "global_load_dwordx4 v[248:251], %2, off\n\tglobal_load_dwordx4 v[252:255], %3, off"
                     :: "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                     : "memory", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250",
                       "v251", "v252", "v253", "v254", "v255");
}
// wait until at most N newer vector-memory operations are outstanding, then set SET -> LDS [addr, addr + 4 x 1024)
template <int N, int SET> __device__ __forceinline__ void wg_stash4(unsigned addr) {
    if constexpr (SET == 0)
        asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, v[224:227]\n\tds_write_b128 %0, v[228:231] offset:1024\n\t"
                     "ds_write_b128 %0, v[232:235] offset:2048\n\tds_write_b128 %0, v[236:239] offset:3072"
                     :: "v"(addr), "n"(N) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(%1)\n\tds_write_b128 %0, v[240:243]\n\tds_write_b128 %0, v[244:247] offset:1024\n\t"
                     "ds_write_b128 %0, v[248:251] offset:2048\n\tds_write_b128 %0, v[252:255] offset:3072"
                     :: "v"(addr), "n"(N) : "memory");
}

template <typename CFG>
__global__ __launch_bounds__(CFG::THREADS, CFG::NW == 4 ? 2 : 1) __attribute__((amdgpu_num_vgpr(224))) void wgrad_gr_kernel(
    const unsigned short *__restrict__ G, const unsigned short *__restrict__ X, int64_t M, int cout_g, int cin_g,
    int tiles_o, int tiles_c, int slices_view, int64_t cols_per_slice, int views, float *__restrict__ part, int nblocks) {
    constexpr int NSX = 3, RT = CFG::RT, CT = CFG::CT, KC = CFG::KC, GD = CFG::G_DMA, XD = CFG::X_DMA;
    static_assert(KC == 64 && GD == 4, "128-byte row pieces; four G vectors per lane and chunk");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // [3 x X tile][2 x G tile]
    constexpr int G0 = NSX * CFG::X_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave / CFG::WM, wc = wave % CFG::WM;
    const unsigned lds0 = (unsigned)(uintptr_t)(gm_lptr)smem;
    const int ntiles = tiles_o * tiles_c;
    const int logical = xcd_remap(blockIdx.x, nblocks);
    const int slice = logical / ntiles, tile = logical - slice * ntiles, grp = blockIdx.z;
    const int view = slice / slices_view, sl = slice - view * slices_view;
    const int o0 = (tile / tiles_c) * CFG::TO, c0 = (tile % tiles_c) * CFG::TC;
    const int64_t Mv = M / views;
    const int64_t m_begin = (int64_t)view * Mv + (int64_t)sl * cols_per_slice;
    int64_t m_len = Mv - (int64_t)sl * cols_per_slice;
    if (m_len > cols_per_slice) m_len = cols_per_slice;
    const int T = (int)(m_len / KC);
    const unsigned short *Gg = G + (size_t)grp * cout_g * M + m_begin;
    const unsigned short *Xg = X + (size_t)grp * cin_g * M + m_begin;

    const unsigned short *g_src[GD], *x_src[XD];
#pragma unroll
    for (int j = 0; j < GD; ++j) {
        const int row = CFG::RPD * (GD * wave + j) + lane / CFG::SLOTS;
        int o = o0 + row;
        if (o > cout_g - 1) o = cout_g - 1;
        g_src[j] = Gg + (size_t)o * M + (((lane & (CFG::SLOTS - 1)) ^ CFG::swz(row)) << 3);
    }
#pragma unroll
    for (int j = 0; j < XD; ++j) {
        const int row = CFG::RPD * (XD * wave + j) + lane / CFG::SLOTS;
        int c = c0 + row;
        if (c > cin_g - 1) c = cin_g - 1;
        x_src[j] = Xg + (size_t)c * M + (((lane & (CFG::SLOTS - 1)) ^ CFG::swz(row)) << 3);
    }
    auto issue = [&](auto PAR, int t) {                             // chunk t: X by DMA, G into register set PAR
        const unsigned st = lds0 + (t % NSX) * CFG::X_BYTES;
#pragma unroll
        for (int j = 0; j < XD; ++j) gm_dma16(x_src[j] + (size_t)t * KC, st + (XD * wave + j) * 1024);
        wg_gload4<decltype(PAR)::value>(g_src[0] + (size_t)t * KC, g_src[1] + (size_t)t * KC, g_src[2] + (size_t)t * KC,
                                        g_src[3] + (size_t)t * KC);
    };
    auto stash = [&](auto PAR, int t, bool newer) {                 // G of chunk t: registers -> its LDS stage
        // operations issued after chunk t's G loads: chunk t + 1's X DMA and G loads, if there is one
        const unsigned addr = lds0 + G0 + (t & 1) * CFG::G_BYTES + GD * wave * 1024 + lane * 16;
        if (newer) wg_stash4<XD + GD, decltype(PAR)::value>(addr);
        else wg_stash4<0, decltype(PAR)::value>(addr);
    };
    int goff[RT], gx[RT], xoff[CT], xx[CT];
#pragma unroll
    for (int a = 0; a < RT; ++a) {
        const int row = wo * 32 * RT + a * 32 + l31;
        goff[a] = G0 + row * CFG::ROWB;
        gx[a] = CFG::swz(row);
    }
#pragma unroll
    for (int b = 0; b < CT; ++b) {
        const int row = wc * 32 * CT + b * 32 + l31;
        xoff[b] = row * CFG::ROWB;
        xx[b] = CFG::swz(row);
    }
    f32x16 acc[RT][CT];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    if (T > 0) issue(P0{}, 0);
    if (T > 1) issue(P1{}, 1);
    if (T > 0) stash(P0{}, 0, T > 1);
    auto body = [&](auto PAR, int t) {                              // PAR = t & 1
        using NXT = std::integral_constant<int, 1 - decltype(PAR)::value>;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's stash of chunk t
        __builtin_amdgcn_s_barrier();                               // chunk t visible; chunk t - 1's stages are free
        if (t + 2 < T) issue(PAR, t + 2);                           // X -> stage (t + 2) % 3, G -> the set chunk t left
        const unsigned char *const xs = smem + (t % NSX) * CFG::X_BYTES;
        const unsigned char *const gs = smem + (t & 1) * CFG::G_BYTES;
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            gm_bf16x8 av[RT], bv[CT];
#pragma unroll
            for (int a = 0; a < RT; ++a)
                av[a] = *reinterpret_cast<const gm_bf16x8 *>(gs + goff[a] + (((2 * ks + half) ^ gx[a]) << 4));
#pragma unroll
            for (int b = 0; b < CT; ++b)
                bv[b] = *reinterpret_cast<const gm_bf16x8 *>(xs + xoff[b] + (((2 * ks + half) ^ xx[b]) << 4));
#pragma unroll
            for (int a = 0; a < RT; ++a)
#pragma unroll
                for (int b = 0; b < CT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        // chunk t + 1's G (requested one iteration ago) -> the G stage chunk t - 1 used; its X DMA is older, so done too
        if (t + 1 < T) stash(NXT{}, t + 1, t + 2 < T);
    };
    for (int t = 0; t < T; t += 2) {
        body(P0{}, t);
        if (t + 1 < T) body(P1{}, t + 1);
    }
    float *pp = part + ((size_t)slice * gridDim.z + grp) * cout_g * cin_g;
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b) {
            const int c = c0 + wc * 32 * CT + b * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + wo * 32 * RT + a * 32 + mfma_row(r, half);
                if (o < cout_g && c < cin_g) pp[(size_t)o * cin_g + c] = acc[a][b][r];
            }
        }
}

struct WgDmaPlan {
    int cfg;                      // 0 = T (64 x 64), 1 = S (128 x 128), 2 = L (256 x 256), 3 = S32, 4 = M32 (256 x 128), 5 = L32,
                                  // 6 = SG, 7 = LG (G through registers), 8 = T128 (64 x 64, 256-byte row pieces), 10 = S128
    int to, tc, tiles_o, tiles_c, slices_view, nslices;
    int64_t cols;
};
// tile: -1 = the measured heuristic below; 0 ... 8, 10 = that configuration (T, S, L, S32, M32, L32, SG, LG, T128; S128); 9 = the
// register-staged split-K kernel of round 1 (wgrad_partial_kernel).  A per-call argument of the *_tile entry points
// (tests force every configuration on small cases); GRAFP_WGRAD_TILE only in measurement builds (tuning.h).
static int wg_tile(int tile) {
    if (tile < 0) tile = GRAFP_TUNE_INT("GRAFP_WGRAD_TILE", -1);
    return (tile >= 0 && tile <= 10) ? tile : -1;
}
static bool wgrad_dma_ok(int cout_g, int cin_g, int64_t M, int views, int tile) {
    return wg_tile(tile) != 9 && cout_g % 32 == 0 && cin_g % 32 == 0 && views >= 1 && M % views == 0 &&
           (M / views) % 64 == 0;
}
static WgDmaPlan wgrad_dma_plan(int cout_g, int cin_g, int groups, int64_t M, int views, bool pro, int tile) {
    WgDmaPlan p;
    const int64_t outs = (int64_t)cout_g * cin_g;
    // Measured (tools/wgrad_sweep.sh: every configuration on every layer shape at 256 / 512 / 1024 / 2048 clip-views;
    // profiles/r02_wgrad_sweep.txt).  Three regimes by the bytes of the two operands:
    //  * they fit the 256 MB Infinity Cache with room to spare: the 64 x 64 tile (many small workgroups, a few MB of
    //    partial sums, its operand re-reads are cache hits at ~14 TB/s), 128 x 128 from 2^19 outputs;
    //  * from 200 MB the wide layers (FFN and fc2 of stages 2-3; from 400 MB everything with >= 128 x 256 rows) on the
    //    256 x 256 tile with the G operand through registers (LG: two chunks of both operands in flight, 1.6x instead
    //    of 3.2x re-reads); the 64-byte-piece form of that tile (L32) is within 5 % of it up to 750 MB and loses
    //    beyond (64-byte pieces at 1-4 MB row stride read at 2.5-3.8 TB/s -- rowpiece_read_bench);
    //  * beyond 750 MB (1024 pairs on one GPU: every re-read of a small tile goes to HBM) everything else with >= 128
    //    rows on one side on the 128 x 128 register-staged tile (SG).
    // The grouped convolutions (32 ... 128 rows per group) stay on the small tile.  With the in-LDS normalisation (pro)
    // only the three DMA tiles of the first round.
    const double opbytes = (double)(cout_g + cin_g) * groups * (double)M * 2.0;
    const bool big = opbytes > 500e6;
    const int lo = cout_g < cin_g ? cout_g : cin_g, hi = cout_g < cin_g ? cin_g : cout_g;
    p.cfg = 0;
    if (lo >= 128 && (outs >= (1 << 19) || (big && groups == 1))) p.cfg = 1;
    if (big && groups == 1 && cout_g >= 64 && cin_g >= 2 * cout_g) p.cfg = 1;
    if (pro && cin_g >= 256 && cout_g >= 128) p.cfg = 1;
    const bool no_wide = GRAFP_TUNE_INT("GRAFP_WGRAD_NO_WIDE", 0) != 0;        // A/B: the three DMA tiles only
    if (!pro && !no_wide) {
        const bool wide = (lo >= 256 && hi >= 1024 && opbytes >= 200e6) || (lo >= 128 && hi >= 256 && opbytes >= 400e6) ||
                          (lo >= 512 && opbytes >= 250e6);
        if (groups == 1) {
            if (wide) p.cfg = 7;                                   // LG, one workgroup per CU (see targets below)
            else if (opbytes >= 750e6 && hi >= 128) p.cfg = 6;     // SG
        } else if (cout_g >= 256 && opbytes >= 400e6) {
            p.cfg = opbytes >= 750e6 ? 6 : 5;
        }
    }
    // 256-byte row pieces (T128) for the small outputs of stages 0-1 from 2^19 columns on (tools/gemm_bench.py --wgrad with
    // the tile forced, 512 / 1024 / 2048 clip-views: 64-row layers -3 ... -25 %, 128 x 128 -5 ... -13 %, the grouped
    // convolution of stage 0 -8 ... -15 %, that of stage 1 -30 % at 2048 clip-views but +24 % at 1024; equal at 256)
    const bool p256 = GRAFP_TUNE_INT("GRAFP_WGRAD_NO_P256", 0) == 0;          // A/B: without the 256-byte-piece tiles
    if (p256 && M >= (1 << 19)) {
        if (groups == 1 && (lo <= 64 || outs <= 128 * 128)) p.cfg = 8;
        if (groups > 1 && (cout_g <= 32 || (cout_g <= 64 && opbytes >= 750e6))) p.cfg = 8;
    }
    // ... and the 128 x 128 tile on 256-byte pieces (S128, one workgroup per CU) for the next size class, 128 x 256 ...
    // 256 x 256 outputs from 250 MB of operands: -9 ... -28 % at 1024 / 2048 clip-views (stage 1 fc2 / FFN, stage 2 fc1);
    // the grouped convolution of stage 2 only at 2048 clip-views (-20 %; +15 % at 1024).  Larger outputs lose 10-30 %.
    if (p256 && !pro) {
        if (groups == 1 && outs > 128 * 128 && outs <= 256 * 256 && opbytes >= 250e6) p.cfg = 10;
        if (groups > 1 && cout_g == 128 && cin_g == 128 && opbytes >= 750e6) p.cfg = 10;
    }
    // ... and at SMALL batches (operands under 200 MB: 128 pairs per GPU, BASELINE config 3's per-rank shape) for the 2^17 ...
    // 2^19-output layers with >= 256 rows on both sides (stage 2 fc2 / FFN, stage 3 fc1), which the first rule leaves on
    // the 64 x 64 tile: tools/wgrad_sweep.sh at 256 clip-views (profiles/r05_wgrad_sweep_256.txt): 65 -> 55 us (FFN),
    // 39 -> 35 (stage 3 fc1), 37 -> 34.5 (stage 2 fc2); the 2^19+ outputs stay where they are (128 x 128: best there).
    if (p256 && !pro && groups == 1 && lo >= 256 && outs >= (1 << 17) && outs < (1 << 19) && opbytes < 200e6) p.cfg = 10;
    const int forced = wg_tile(tile);
    if (forced >= 0 && forced != 9 && !(pro && (forced == 6 || forced == 7))) p.cfg = forced;
    if (p.cfg == 8 && (M / views) % 128 != 0) p.cfg = 0;          // T128 / S128 need whole 128-column chunks
    if (p.cfg == 10 && (M / views) % 128 != 0) p.cfg = 1;
    static const int tile_o[11] = {64, 128, 256, 128, 256, 256, 128, 256, 64, 0, 128}, tile_c[11] = {64, 128, 256, 128, 128, 256, 128, 256, 64, 0, 128};
    p.to = tile_o[p.cfg];
    p.tc = tile_c[p.cfg];
    p.tiles_o = (cout_g + p.to - 1) / p.to;
    p.tiles_c = (cin_g + p.tc - 1) / p.tc;
    const int64_t tiles = (int64_t)p.tiles_o * p.tiles_c * groups;
    const int64_t Mv = M / views;
    // two rounds of resident workgroups (L: one per CU, T/S: two), at least 8 chunks per slice
    // workgroups to aim for: ONE round of resident workgroups (512 for the tiles with two workgroups per CU: -11 % over
    // all layers at 256 clip-views, -5 % at 512 against two rounds; one workgroup per CU
    // for the 256 x 256 tiles -- a second round doubles their partial sums (128 slices x 1 MB written and read back
    // against 1.3 GB of operands) and was 5-30 % slower on every shape (LG at 2048 clip-views: 16.4 -> 15.0 ms per step)
    static const int64_t targets[11] = {512, 512, 512, 1024, 512, 256, 512, 256, 512, 0, 256};
    int64_t target = targets[p.cfg];
    if ((p.cfg <= 1 || p.cfg == 8) && opbytes >= 750e6) target = 1024;        // the small tiles at 1024 pairs per GPU: two rounds (+2 %)
    if (GRAFP_TUNE_INT("GRAFP_WGRAD_TARGET", 0) > 0) target = GRAFP_TUNE_INT("GRAFP_WGRAD_TARGET", 0);
    int64_t sv = (target + tiles * views - 1) / (tiles * views);
    const int64_t max_sv = (Mv / 64 + 7) / 8;
    if (sv > max_sv) sv = max_sv;
    if (sv < 1) sv = 1;
    int64_t cols = (Mv + sv - 1) / sv;
    const int64_t kc = p.cfg == 8 || p.cfg == 10 ? 128 : 64;
    cols = (cols + kc - 1) / kc * kc;
    p.cols = cols;
    p.slices_view = (int)((Mv + cols - 1) / cols);
    p.nslices = p.slices_view * views;
    return p;
}

// out[i] = sum_k part[k][i]: 16 outputs x 16 split-lanes per workgroup, fixed summation order (deterministic)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, int S, int64_t n,
                                                           float *__restrict__ out) {
    __shared__ float red[16][17];
    const int tid = threadIdx.x, j = tid & 15, q = tid >> 4;
    const int64_t i = (int64_t)blockIdx.x * 16 + j;
    float s = 0.0f;
    if (i < n) {
#pragma unroll 4
        for (int k = q; k < S; k += 16) s += part[(size_t)k * n + i];
    }
    red[q][j] = s;
    __syncthreads();
    if (tid < 16 && i < n) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][tid];
        out[i] = t;
    }
}

struct WgradPlan {
    int tw, tiles_o, tiles_c, S;
    int64_t cols;
};
static WgradPlan wgrad_plan(int cout_g, int cin_g, int groups, int64_t M) {
    WgradPlan p;
    // measured crossover.  (Two chunks in flight per workgroup -- a second register set -- costs the 128-tile its
    // second workgroup per CU (320 registers) and was 30-40 % slower on the stage 2-3 shapes, 3-8 % on the 64-tile.)
    // (A 256 x 128 tile with 64-column chunks -- 85 flop per operand byte instead of 64 -- was
    // tried and is 5-30 % slower on every FFN shape: the halved chunk doubles the barriers per MFMA.)
    p.tw = (cout_g >= 128 && cin_g >= 128 && (int64_t)cout_g * cin_g >= 65536) ? 128 : 64;
    p.tiles_o = (cout_g + p.tw - 1) / p.tw;
    p.tiles_c = (cin_g + p.tw - 1) / p.tw;
    const int64_t tiles = (int64_t)p.tiles_o * p.tiles_c * groups;
    // measured on MI355X: ~1024 workgroups for the 64-tiles (small outputs: parallelism must come from split-K),
    // ~512 for the 128-tiles (large outputs: more splits only add partial-sum traffic)
    const int64_t target = p.tw == 64 ? 1024 : 512;
    int64_t S = (target + tiles - 1) / tiles;
    const int64_t max_s = (M + 4 * WG_KC - 1) / (4 * WG_KC);  // >= 4 chunks per slice
    if (S > max_s) S = max_s;
    if (S > 256) S = 256;
    if (S < 1) S = 1;
    int64_t cols = (M + S - 1) / S;
    cols = (cols + WG_KC - 1) / WG_KC * WG_KC;
    p.cols = cols;
    p.S = (int)((M + cols - 1) / cols);
    return p;
}

}  // namespace grafp

extern "C" int grafp_conv1x1_wgrad_plan(int Cout, int Cin, int groups, int64_t M, int views, int *info) {
    using namespace grafp;
    GRAFP_REQUIRE(info, "conv1x1_wgrad_plan: null pointer");
    GRAFP_REQUIRE(Cout > 0 && Cin > 0 && groups > 0 && M > 0 && views > 0 && Cout % groups == 0 && Cin % groups == 0,
                  "conv1x1_wgrad_plan: bad shape");
    const int cout_g = Cout / groups, cin_g = Cin / groups;
    for (int i = 0; i < 8; ++i) info[i] = 0;
    if (wgrad_dma_ok(cout_g, cin_g, M, views, -1)) {
        const WgDmaPlan p = wgrad_dma_plan(cout_g, cin_g, groups, M, views, false, -1);
        info[0] = p.cfg; info[1] = p.to; info[2] = p.tc; info[3] = p.nslices; info[4] = p.tiles_o * p.tiles_c * groups;
    } else {
        const WgradPlan p = wgrad_plan(cout_g, cin_g, groups, M);
        info[0] = -1; info[1] = info[2] = p.tw; info[3] = p.S; info[4] = p.tiles_o * p.tiles_c * groups;
    }
    return GRAFP_OK;
}

extern "C" size_t grafp_conv1x1_wgrad_tile_workspace(int Cout, int Cin, int groups, int64_t M, int views, int tile) {
    using namespace grafp;
    if (Cout <= 0 || Cin <= 0 || groups <= 0 || M <= 0 || Cout % groups || Cin % groups) return 0;
    const int cout_g = Cout / groups, cin_g = Cin / groups;
    if (wgrad_dma_ok(cout_g, cin_g, M, views, tile)) {
        const WgDmaPlan p = wgrad_dma_plan(cout_g, cin_g, groups, M, views, false, tile);
        const WgDmaPlan q = wgrad_dma_plan(cout_g, cin_g, groups, M, views, true, tile);
        return (size_t)(p.nslices > q.nslices ? p.nslices : q.nslices) * Cout * cin_g * sizeof(float);
    }
    const WgradPlan p = wgrad_plan(cout_g, cin_g, groups, M);
    return (size_t)p.S * Cout * cin_g * sizeof(float);
}

extern "C" size_t grafp_conv1x1_wgrad_pro_workspace(int Cout, int Cin, int groups, int64_t M, int views) {
    return grafp_conv1x1_wgrad_tile_workspace(Cout, Cin, groups, M, views, -1);
}

extern "C" size_t grafp_conv1x1_wgrad_workspace(int Cout, int Cin, int groups, int64_t M) {
    return grafp_conv1x1_wgrad_pro_workspace(Cout, Cin, groups, M, 1);
}

// dweight == nullptr: the partial sums only (n_slices receives how many there are per output element); the caller reduces
// them later, together with other layers' (grafp_wgrad_reduce_multi)
static int wgrad_tile_impl(const void *grad_out, const void *x, int Cout, int Cin, int groups, int64_t M, int views,
                           const float *pro_tab, int pro_act, float pro_slope, int tile, float *dweight, void *ws,
                           size_t ws_bytes, int *n_slices, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(grad_out && x && (dweight || n_slices), "conv1x1_wgrad: null pointer");
    GRAFP_REQUIRE(Cout > 0 && Cin > 0 && groups > 0 && M > 0 && Cout % groups == 0 && Cin % groups == 0,
                  "conv1x1_wgrad: bad shape Cout=%d Cin=%d groups=%d M=%lld", Cout, Cin, groups, (long long)M);
    GRAFP_REQUIRE((((uintptr_t)grad_out | (uintptr_t)x) & 15) == 0, "conv1x1_wgrad: operands must be 16-byte aligned");
    GRAFP_REQUIRE(pro_act >= 0 && pro_act <= 2, "conv1x1_wgrad: bad activation %d", pro_act);
    const size_t need = grafp_conv1x1_wgrad_tile_workspace(Cout, Cin, groups, M, views, tile);
    if (!ws || ws_bytes < need) {
        set_error("conv1x1_wgrad: workspace %zu bytes < required %zu", ws_bytes, need);
        return GRAFP_ERR_WORKSPACE;
    }
    const int cout_g = Cout / groups, cin_g = Cin / groups;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)Cout * cin_g;
    if (wgrad_dma_ok(cout_g, cin_g, M, views, tile)) {
        const WgDmaPlan p = wgrad_dma_plan(cout_g, cin_g, groups, M, views, pro_tab != nullptr, tile);
        const int nblocks = p.nslices * p.tiles_o * p.tiles_c;
        const dim3 grid(nblocks, 1, groups);
#define WG_LAUNCH(CFG, PRO)                                                                                              \
    do {                                                                                                                 \
        const size_t lds = (size_t)CFG::NS * CFG::STAGE + ((PRO) ? (size_t)CFG::TC * 8 : 0);                             \
        (void)hipFuncSetAttribute((const void *)wgrad_dma_kernel<CFG, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                  (int)lds);                                                                             \
        hipLaunchKernelGGL((wgrad_dma_kernel<CFG, PRO>), grid, dim3(CFG::THREADS), lds, s,                               \
                           (const unsigned short *)grad_out, (const unsigned short *)x, M, cout_g, cin_g, p.tiles_o,     \
                           p.tiles_c, p.slices_view, p.cols, views, (const float2 *)pro_tab, pro_act, pro_slope,         \
                           (float *)ws, nblocks);                                                                        \
    } while (0)
#define WG_LAUNCH_GR(CFG)                                                                                                \
    do {                                                                                                                 \
        const size_t lds = (size_t)3 * CFG::X_BYTES + 2 * CFG::G_BYTES;                                                  \
        (void)hipFuncSetAttribute((const void *)wgrad_gr_kernel<CFG>, hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                  (int)lds);                                                                             \
        hipLaunchKernelGGL((wgrad_gr_kernel<CFG>), grid, dim3(CFG::THREADS), lds, s, (const unsigned short *)grad_out,   \
                           (const unsigned short *)x, M, cout_g, cin_g, p.tiles_o, p.tiles_c, p.slices_view, p.cols,     \
                           views, (float *)ws, nblocks);                                                                 \
    } while (0)
        if (pro_tab) {
            switch (p.cfg) {
            case 0: WG_LAUNCH(WgT, true); break;
            case 8: WG_LAUNCH(WgT128, true); break;
            case 10: WG_LAUNCH(WgS128, true); break;
            case 1: WG_LAUNCH(WgS, true); break;
            case 2: WG_LAUNCH(WgL, true); break;
            case 3: WG_LAUNCH(WgS32, true); break;
            case 4: WG_LAUNCH(WgM32, true); break;
            default: WG_LAUNCH(WgL32, true); break;
            }
        } else {
            switch (p.cfg) {
            case 6: WG_LAUNCH_GR(WgSG); break;
            case 7: WG_LAUNCH_GR(WgLG); break;
            case 0: WG_LAUNCH(WgT, false); break;
            case 8: WG_LAUNCH(WgT128, false); break;
            case 10: WG_LAUNCH(WgS128, false); break;
            case 1: WG_LAUNCH(WgS, false); break;
            case 2: WG_LAUNCH(WgL, false); break;
            case 3: WG_LAUNCH(WgS32, false); break;
            case 4: WG_LAUNCH(WgM32, false); break;
            default: WG_LAUNCH(WgL32, false); break;
            }
        }
#undef WG_LAUNCH
#undef WG_LAUNCH_GR
        GRAFP_CHECK_LAUNCH("wgrad_dma_kernel");
        if (n_slices) *n_slices = p.nslices;
        if (!dweight) return GRAFP_OK;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, (const float *)ws,
                           p.nslices, n, dweight);
        GRAFP_CHECK_LAUNCH("wgrad_reduce_kernel");
        return GRAFP_OK;
    }
    GRAFP_REQUIRE(!pro_tab, "conv1x1_wgrad: the normalise-on-load form needs rows per group %% 32 == 0 and columns per "
                            "view %% 64 == 0");
    const WgradPlan p = wgrad_plan(cout_g, cin_g, groups, M);
    GRAFP_REQUIRE((int64_t)p.S * p.tiles_o * p.tiles_c < (1ll << 31), "conv1x1_wgrad: output too large");
    const dim3 grid(p.S * p.tiles_o * p.tiles_c, 1, groups);
    const size_t lds = (size_t)2 * p.tw * WG_LS;
    if (p.tw == 64) {
        hipLaunchKernelGGL(wgrad_partial_kernel<64>, grid, dim3(256), lds, s, (const unsigned short *)grad_out,
                           (const unsigned short *)x, M, cout_g, cin_g, p.tiles_o, p.tiles_c, p.cols, (float *)ws);
    } else {
        (void)hipFuncSetAttribute((const void *)wgrad_partial_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(wgrad_partial_kernel<128>, grid, dim3(256), lds, s, (const unsigned short *)grad_out,
                           (const unsigned short *)x, M, cout_g, cin_g, p.tiles_o, p.tiles_c, p.cols, (float *)ws);
    }
    GRAFP_CHECK_LAUNCH("wgrad_partial_kernel");
    if (n_slices) *n_slices = p.S;
    if (!dweight) return GRAFP_OK;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, (const float *)ws, p.S, n,
                       dweight);
    GRAFP_CHECK_LAUNCH("wgrad_reduce_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_conv1x1_wgrad_tile_bf16(const void *grad_out, const void *x, int Cout, int Cin, int groups,
                                             int64_t M, int views, const float *pro_tab, int pro_act, float pro_slope,
                                             int tile, float *dweight, void *ws, size_t ws_bytes, grafp_stream_t stream) {
    GRAFP_REQUIRE(dweight, "conv1x1_wgrad: null pointer");
    return wgrad_tile_impl(grad_out, x, Cout, Cin, groups, M, views, pro_tab, pro_act, pro_slope, tile, dweight, ws, ws_bytes,
                           nullptr, stream);
}

extern "C" int grafp_conv1x1_wgrad_partials_bf16(const void *grad_out, const void *x, int Cout, int Cin, int groups,
                                                 int64_t M, int views, const float *pro_tab, int pro_act, float pro_slope,
                                                 int tile, void *ws, size_t ws_bytes, int *n_slices,
                                                 grafp_stream_t stream) {
    GRAFP_REQUIRE(n_slices, "conv1x1_wgrad_partials: null pointer");
    return wgrad_tile_impl(grad_out, x, Cout, Cin, groups, M, views, pro_tab, pro_act, pro_slope, tile, nullptr, ws, ws_bytes,
                           n_slices, stream);
}

// The reductions of MANY layers' partial sums in one launch: a training step has 64 weight gradients, each followed by a
// ~4-8 us reduction launch of its own -- 3 % of the GPU time of a 128-pairs-per-GPU step (profiles/r04_a128_kernel_stats).
// The table travels BY VALUE in the kernel arguments (no host-to-device copy: the launch is graph-capturable and needs
// no device buffer); same 16 x 16 summation tree as wgrad_reduce_kernel, so the gradients are bit-identical.
namespace grafp {
constexpr int WG_MULTI = 64;
struct WgReduceTable {
    const float *part[WG_MULTI];
    float *out[WG_MULTI];
    int S[WG_MULTI], n[WG_MULTI], block_base[WG_MULTI + 1];
    int count;
};
// A workgroup takes WG_MULTI_RG groups of 16 consecutive outputs of ONE entry (found by bisection of block_base: the table
// sits in kernel-argument memory, and the 64-step linear walk of the first version cost every one of a million 16-output
// workgroups ~2 us -- 1.1 ms per step, more than the 64 launches it replaced).  Per output the order of the sum is that of
// wgrad_reduce_kernel: 16 split-lanes k = q, q + 16, ..., then the 16 lane sums in lane order.
constexpr int WG_MULTI_RG = 8;
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const WgReduceTable t) {
    __shared__ float red[16][WG_MULTI_RG * 16 + 1];
    int lo = 0, hi = t.count - 1;                                                  // uniform, <= 6 steps
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)blockIdx.x >= t.block_base[mid]) lo = mid;
        else hi = mid - 1;
    }
    const int e = lo;
    const float *__restrict__ part = t.part[e];
    const int S = t.S[e];
    const int64_t n = t.n[e];
    const int tid = threadIdx.x, j = tid & 15, q = tid >> 4;
    const int64_t i0 = (int64_t)(blockIdx.x - t.block_base[e]) * (16 * WG_MULTI_RG) + j;
    float s[WG_MULTI_RG];
#pragma unroll
    for (int g = 0; g < WG_MULTI_RG; ++g) s[g] = 0.0f;
    if (i0 - j + 16 * WG_MULTI_RG <= n) {                  // uniform: every output of the workgroup exists -- plain loads,
        for (int k = q; k < S; k += 16) {                  // eight in flight per lane
            const float *row = part + (size_t)k * n + i0;
            float v[WG_MULTI_RG];
#pragma unroll
            for (int g = 0; g < WG_MULTI_RG; ++g) v[g] = row[16 * g];
#pragma unroll
            for (int g = 0; g < WG_MULTI_RG; ++g) s[g] += v[g];
        }
    } else {                                               // the entry's last workgroup
        for (int k = q; k < S; k += 16) {
            const float *row = part + (size_t)k * n;
#pragma unroll
            for (int g = 0; g < WG_MULTI_RG; ++g) {
                const int64_t i = i0 + 16 * g;
                if (i < n) s[g] += row[i];
            }
        }
    }
#pragma unroll
    for (int g = 0; g < WG_MULTI_RG; ++g) red[q][16 * g + j] = s[g];
    __syncthreads();
    if (tid < 16 * WG_MULTI_RG) {
        const int64_t i = (int64_t)(blockIdx.x - t.block_base[e]) * (16 * WG_MULTI_RG) + tid;
        if (i < n) {
            float r = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k) r += red[k][tid];
            t.out[e][i] = r;
        }
    }
}
}  // namespace grafp

extern "C" int grafp_wgrad_reduce_multi(const void *const *parts, const int *n_slices, const int64_t *n_out,
                                        float *const *outs, int n_entries, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(n_entries >= 0 && (n_entries == 0 || (parts && n_slices && n_out && outs)), "wgrad_reduce_multi: null pointer");
    for (int e0 = 0; e0 < n_entries; e0 += WG_MULTI) {
        WgReduceTable t;
        t.count = n_entries - e0 < WG_MULTI ? n_entries - e0 : WG_MULTI;
        int blocks = 0;
        for (int e = 0; e < t.count; ++e) {
            GRAFP_REQUIRE(parts[e0 + e] && outs[e0 + e] && n_slices[e0 + e] > 0 && n_out[e0 + e] > 0 && n_out[e0 + e] < (1ll << 31),
                          "wgrad_reduce_multi: bad entry %d", e0 + e);
            t.part[e] = (const float *)parts[e0 + e];
            t.out[e] = outs[e0 + e];
            t.S[e] = n_slices[e0 + e];
            t.n[e] = (int)n_out[e0 + e];
            t.block_base[e] = blocks;
            blocks += (int)((n_out[e0 + e] + 16 * WG_MULTI_RG - 1) / (16 * WG_MULTI_RG));
        }
        t.block_base[t.count] = blocks;
        for (int e = t.count; e < WG_MULTI; ++e) {
            t.part[e] = nullptr; t.out[e] = nullptr; t.S[e] = 0; t.n[e] = 0; t.block_base[e + 1] = blocks;
        }
        hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t);
        GRAFP_CHECK_LAUNCH("wgrad_reduce_multi_kernel");
    }
    return GRAFP_OK;
}

extern "C" int grafp_conv1x1_wgrad_pro_bf16(const void *grad_out, const void *x, int Cout, int Cin, int groups,
                                            int64_t M, int views, const float *pro_tab, int pro_act, float pro_slope,
                                            float *dweight, void *ws, size_t ws_bytes, grafp_stream_t stream) {
    return grafp_conv1x1_wgrad_tile_bf16(grad_out, x, Cout, Cin, groups, M, views, pro_tab, pro_act, pro_slope, -1, dweight,
                                         ws, ws_bytes, stream);
}

extern "C" int grafp_conv1x1_wgrad_bf16(const void *grad_out, const void *x, int Cout, int Cin, int groups, int64_t M,
                                        float *dweight, void *ws, size_t ws_bytes, grafp_stream_t stream) {
    return grafp_conv1x1_wgrad_tile_bf16(grad_out, x, Cout, Cin, groups, M, 1, nullptr, 0, 0.0f, -1, dweight, ws, ws_bytes,
                                         stream);
}

static grafp::WgradPlan wgrad3_plan(int cout_g, int cin_g, int groups, int64_t M) {
    using namespace grafp;
    WgradPlan p;
    p.tw = 64;
    p.tiles_o = (cout_g + 63) / 64;
    p.tiles_c = (cin_g + 63) / 64;
    const int64_t tiles = (int64_t)p.tiles_o * p.tiles_c * groups;
    int64_t S = (1024 + tiles - 1) / tiles;
    const int64_t max_s = (M + 4 * WG_KC - 1) / (4 * WG_KC);
    if (S > max_s) S = max_s;
    if (S > 256) S = 256;
    if (S < 1) S = 1;
    int64_t cols = (M + S - 1) / S;
    cols = (cols + WG_KC - 1) / WG_KC * WG_KC;
    p.cols = cols;
    p.S = (int)((M + cols - 1) / cols);
    return p;
}

extern "C" size_t grafp_conv1x1_wgrad_f32_workspace(int Cout, int Cin, int groups, int64_t M) {
    if (Cout <= 0 || Cin <= 0 || groups <= 0 || M <= 0 || Cout % groups || Cin % groups) return 0;
    const grafp::WgradPlan p = wgrad3_plan(Cout / groups, Cin / groups, groups, M);
    return (size_t)p.S * Cout * (Cin / groups) * sizeof(float);
}

extern "C" int grafp_conv1x1_wgrad_f32(const float *grad_out, const float *x, int Cout, int Cin, int groups, int64_t M,
                                       float *dweight, void *ws, size_t ws_bytes, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(grad_out && x && dweight, "conv1x1_wgrad_f32: null pointer");
    GRAFP_REQUIRE(Cout > 0 && Cin > 0 && groups > 0 && M > 0 && Cout % groups == 0 && Cin % groups == 0,
                  "conv1x1_wgrad_f32: bad shape Cout=%d Cin=%d groups=%d M=%lld", Cout, Cin, groups, (long long)M);
    GRAFP_REQUIRE((((uintptr_t)grad_out | (uintptr_t)x) & 15) == 0, "conv1x1_wgrad_f32: operands must be 16-byte aligned");
    const size_t need = grafp_conv1x1_wgrad_f32_workspace(Cout, Cin, groups, M);
    if (!ws || ws_bytes < need) {
        set_error("conv1x1_wgrad_f32: workspace %zu bytes < required %zu", ws_bytes, need);
        return GRAFP_ERR_WORKSPACE;
    }
    const int cout_g = Cout / groups, cin_g = Cin / groups;
    const WgradPlan p = wgrad3_plan(cout_g, cin_g, groups, M);
    GRAFP_REQUIRE((int64_t)p.S * p.tiles_o * p.tiles_c < (1ll << 31), "conv1x1_wgrad_f32: output too large");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(p.S * p.tiles_o * p.tiles_c, 1, groups);
    const size_t lds = (size_t)4 * 64 * WG_LS;
    (void)hipFuncSetAttribute((const void *)wgrad3_partial_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(wgrad3_partial_kernel, grid, dim3(256), lds, s, grad_out, x, M, cout_g, cin_g, p.tiles_o, p.tiles_c,
                       p.cols, (float *)ws);
    GRAFP_CHECK_LAUNCH("wgrad3_partial_kernel");
    const int64_t n = (int64_t)Cout * cin_g;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, (const float *)ws, p.S, n,
                       dweight);
    GRAFP_CHECK_LAUNCH("wgrad_reduce_kernel");
    return GRAFP_OK;
}
