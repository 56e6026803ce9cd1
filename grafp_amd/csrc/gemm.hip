// gemm.hip -- the 1x1 convolutions of the encoder as ONE streaming bf16 GEMM family on the (C, M = B*N) layout, gfx950.
//
//   Y[r][m] = sum_k W[r][k] * f(X[k][m])        W (R x K) bf16 row-major, X (K x M) bf16 rows contiguous along M,
//                                               Y (R x M) bf16; f = identity, or the PREVIOUS layer's BatchNorm +
//                                               activation applied while the operand is staged (PRO)
// Forward of every Conv2d(1x1) (/root/reference/encoder/gcn_lib/torch_vertex.py:152-162, torch_nn.py:56-60,
// encoder/graph_encoder.py:21-24,52-55) and, with W transposed by the caller, its data gradient.  The shapes are
// skinny (R, K = 64 ... 2048; M = 65 536 ... 524 288 at 256 pairs), so at stages 0-1 the kernel is a STREAM of X tiles
// and Y tiles through HBM and only at stages 2-3 matrix-bound; what is fused here is what used to cost whole extra
// passes over the activations:
//   * STATS: per output row the shifted sums  sum(y - s), sum((y - s)^2)  of the bf16-ROUNDED outputs, accumulated in
//     registers over all tiles of a workgroup and written as ONE partial per (row, workgroup-wave): the BatchNorm that
//     follows needs no statistics pass (grafp_bn_finalize + grafp_bn_affine, or the next GEMM's PRO);
//   * PRO: x -> act(x * scale[k] + shift[k]) per operand row while the tile sits in LDS: the normalised hidden
//     activation of the FFN (4C rows, the largest tensor of a block) and of the max-relative conv are never written.
// Structure: 4 waves, 128 x 128 output tile, 64 (r) x 64 (m) per wave as 2 x 2 v_mfma_f32_32x32x16_bf16 tiles with
// the X fragment as the A operand (D[i = m][j = r]: a lane then holds 4 CONSECUTIVE m per register group -> 8-byte
// pieces of a Y row, and a row's statistics reduce over a lane's own registers).  Operands arrive by LDS-DMA
// (global_load_lds_dwordx4) into a ring of NS stages of [W 128 x 32 | X 32 x 128], NS-1 chunks in flight per
// workgroup across raw s_barriers with COUNTED s_waitcnt vmcnt (the epilogue's stores are counted too); the X
// fragment needs 8 consecutive k of one column from a tile whose rows are k: ds_read_b64_tr_b16 (the hardware
// transpose read; lane i of a 16-lane group supplies the address of row i/4, columns 4(i%4)..+3 and receives
// column i, rows 0..3 -- tools/microbench/tr_read_probe.hip) with the 64-byte segments of a row XOR-swizzled by
// (k & 3) on the DMA source side and on the read side (conflict-free: a 32-lane pass reads 4 rows x 64 B).
// The W tile rows are 64 B; their 16-byte slots are swizzled by (r >> 2) & 3 the same way (ds_read_b128).
#include "common.h"

namespace grafp {

typedef short gm_bf16x8 __attribute__((ext_vector_type(8)));
typedef short gm_s16x4 __attribute__((ext_vector_type(4)));
typedef float gm_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 gm_bf16x2 __attribute__((ext_vector_type(2)));
typedef const void __attribute__((address_space(1))) *gm_gptr;
typedef void __attribute__((address_space(3))) *gm_lptr;

constexpr int GM_T = 128;                      // output tile edge (rows r and columns m)
constexpr int GM_KC = 32;                      // contraction per chunk
constexpr int GM_A_BYTES = GM_T * GM_KC * 2;   // W chunk: 128 rows x 64 B
constexpr int GM_B_BYTES = GM_KC * GM_T * 2;   // X chunk: 32 rows x 256 B
constexpr int GM_STAGE = GM_A_BYTES + GM_B_BYTES;
constexpr int GM_OUT_BYTES = 32 * 128;         // per-wave output staging: 32 rows (r) x 64 m bf16
constexpr int GM_DMA_PER_CHUNK = 4;            // LDS-DMA instructions per wave and chunk (2 W + 2 X)
constexpr int GM_STORES_PER_RT = 4;            // 16-byte store instructions per wave and 32-row output tile

__device__ __forceinline__ unsigned gm_pack_bf16(float a, float b) {
    const gm_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gm_bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
}

// One LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global addresses to LDS [lds_base + lane * 16).  Issued
// through inline asm ON PURPOSE: hipcc's wait-count pass treats every ds_read after a builtin LDS-DMA as a possible
// reader of its destination and drains vmcnt(0) in front of it (seen in the .s: one full drain per chunk), which
// would serialise the ring.  Here the DMA is invisible to that pass and the waits are counted by hand (below).
__device__ __forceinline__ void gm_dma16(const void *gsrc, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_base) : "memory", "m0");
}

template <int N> __device__ __forceinline__ void gm_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// at most `allowed` of this wave's newest vector-memory operations may still be in flight (rounded DOWN to a step)
__device__ __forceinline__ void gm_wait_allowed(int allowed) {
    if (allowed >= 32) gm_wait_vm<32>();
    else if (allowed >= 24) gm_wait_vm<24>();
    else if (allowed >= 20) gm_wait_vm<20>();
    else if (allowed >= 16) gm_wait_vm<16>();
    else if (allowed >= 12) gm_wait_vm<12>();
    else if (allowed >= 8) gm_wait_vm<8>();
    else if (allowed >= 4) gm_wait_vm<4>();
    else gm_wait_vm<0>();
}

struct GemmPlan {
    int row_tiles, col_tiles_view, ranges_view, tiles_range, nblocks, P;
};
// R rows per conv group, M columns, `views` column segments with separate statistics (a range never straddles two)
static GemmPlan gemm_plan(int Rg, int K, int groups, int64_t M, int views) {
    GemmPlan p;
    p.row_tiles = (Rg + GM_T - 1) / GM_T;
    p.col_tiles_view = (int)((M / views) / GM_T);
    const int nch = K / GM_KC;
    // ~1024 workgroups (two rounds of 2 per CU), but at least ~8 chunks per workgroup to amortise the pipeline fill
    int64_t want = 1024 / ((int64_t)p.row_tiles * groups * views);
    if (want < 1) want = 1;
    int tiles_range = (int)((p.col_tiles_view + want - 1) / want);
    const int min_tiles = (8 + nch - 1) / nch;
    if (tiles_range < min_tiles) tiles_range = min_tiles;
    if (tiles_range > p.col_tiles_view) tiles_range = p.col_tiles_view;
    p.tiles_range = tiles_range;
    p.ranges_view = (p.col_tiles_view + tiles_range - 1) / tiles_range;
    p.nblocks = p.row_tiles * p.ranges_view * views;
    p.P = p.ranges_view * 2;          // one partial per (range, wave column)
    return p;
}

template <int NS, bool PRO, bool STATS>
__global__ __launch_bounds__(256, 2) void conv1x1_gemm_kernel(
    const unsigned short *__restrict__ A, int lda, const unsigned short *__restrict__ X, unsigned short *__restrict__ Y,
    int64_t M, int Rg, int K, int row_tiles, int ranges_view, int tiles_range, int col_tiles_view, int views,
    const float2 *__restrict__ pro_tab, int pro_act, float pro_slope, float *__restrict__ part, int P, int nblocks) {
    constexpr int D = NS - 1;                               // chunks in flight
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *const s_out = smem + NS * GM_STAGE;      // 4 x GM_OUT_BYTES
    unsigned char *const s_tab = s_out + 4 * GM_OUT_BYTES;  // PRO: K x float2 (scale, shift) of this view

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: everything derived stays in SGPRs
    const int wr = wave >> 1, wm = wave & 1;
    const unsigned lds0 = (unsigned)(uintptr_t)(gm_lptr)smem;
    const int logical = xcd_remap(blockIdx.x, nblocks);
    const int rt = logical % row_tiles, range = logical / row_tiles;
    const int view = range / ranges_view, rloc = range - view * ranges_view;
    const int grp = blockIdx.z;
    const int tile0 = rloc * tiles_range;
    const int ntile = (tiles_range < col_tiles_view - tile0) ? tiles_range : col_tiles_view - tile0;
    const int nch = K / GM_KC;
    const int T = ntile * nch;
    const int r0 = rt * GM_T;
    const int64_t col0 = (int64_t)view * (M / views) + (int64_t)tile0 * GM_T;
    A += (size_t)grp * Rg * lda;
    X += (size_t)grp * K * M;
    Y += (size_t)grp * Rg * M;

    if (PRO) {       // this view's (scale, shift) per operand row -> LDS (ordinary loads: before any DMA is in flight)
        const float2 *src = pro_tab + ((size_t)grp * K) * views;
        for (int k = tid; k < K; k += 256) reinterpret_cast<float2 *>(s_tab)[k] = src[(size_t)k * views + view];
    }

    // ---- DMA source addresses of this lane (LDS side is lane-linear: stage + instruction * 1 KiB + lane * 16) ----
    // W chunk: instruction q = 2*wave + j covers rows 16q .. 16q+15 (64 B each); slot' = lane & 3 holds source slot
    //          slot' ^ ((row >> 2) & 3)
    const unsigned short *a_src[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int row = r0 + 16 * (2 * wave + j) + (lane >> 2);
        if (row > Rg - 1) row = Rg - 1;                      // rows beyond R: duplicates, never stored
        const int slot = (lane & 3) ^ ((lane >> 4) & 3);
        a_src[j] = A + (size_t)row * lda + slot * 8;
    }
    // X chunk: instruction q covers k-rows 4q .. 4q+3 (256 B each); 16-byte slot s' = lane & 15 of row (lane >> 4)
    //          holds source segment (s' >> 2) ^ (row & 3), piece s' & 3
    const unsigned short *b_src[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 4 * (2 * wave + j) + (lane >> 4);
        const int seg = ((lane & 15) >> 2) ^ (lane >> 4);
        b_src[j] = X + (size_t)row * M + col0 + (seg * 4 + (lane & 3)) * 8;
    }
    int is_ch = 0;                                           // chunk-in-tile of the next chunk to issue
    auto issue = [&](int t) {
        const unsigned st = lds0 + (t % NS) * GM_STAGE + 2 * wave * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            gm_dma16(a_src[j] + is_ch * GM_KC, st + j * 1024);
            gm_dma16(b_src[j] + (size_t)is_ch * GM_KC * M, st + GM_A_BYTES + j * 1024);
        }
        if (++is_ch == nch) {
            is_ch = 0;
            b_src[0] += GM_T;
            b_src[1] += GM_T;
        }
    };

    // ---- fragment read offsets of this lane inside a stage ----
    // X fragment (A operand, i = m): lane i = lane & 15 of group (lane >> 4): row 8*(lane>>5) + i/4 (+ 16 ks + 4 t2),
    // columns wm*64 + mi*32 + 16*((lane>>4)&1) + 4*(i&3)
    int xoff[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int i = lane & 15;
        const int bytecol = (wm * 64 + mi * 32 + 16 * ((lane >> 4) & 1) + 4 * (i & 3)) * 2;
        const int seg = (bytecol >> 6) ^ (i >> 2);
        xoff[mi] = GM_A_BYTES + (8 * half + (i >> 2)) * 256 + seg * 64 + (bytecol & 63);
    }
    // W fragment (B operand, j = r): row wr*64 + ri*32 + l31, 16-byte slot (2 ks + half) ^ ((row >> 2) & 3)
    int woff[2][2];
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row = wr * 64 + ri * 32 + l31;
            woff[ri][ks] = row * 64 + (((2 * ks + half) ^ ((row >> 2) & 3)) << 4);
        }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float sS[2] = {0.f, 0.f}, sQ[2] = {0.f, 0.f}, sShift[2] = {0.f, 0.f};
    // output tiles of 32 rows this wave really owns (R is a multiple of 32)
    bool rt_valid[2];
#pragma unroll
    for (int ri = 0; ri < 2; ++ri) rt_valid[ri] = r0 + wr * 64 + ri * 32 < Rg;
    const int stores_per_epi = GM_STORES_PER_RT * ((int)rt_valid[0] + (int)rt_valid[1]);

    // vector-memory operations issued AFTER the DMA of chunk t, by iteration: dma_hist[j] / st_hist[j] = issued in
    // iteration t-1-j (see the header: the wait for chunk t may leave exactly those in flight)
    int dma_hist[D], st_hist[D + 1];
#pragma unroll
    for (int j = 0; j < D; ++j) dma_hist[j] = 0;
#pragma unroll
    for (int j = 0; j <= D; ++j) st_hist[j] = 0;

    if (PRO) __syncthreads();                                // table visible; nothing in flight yet
    // prologue: chunks 0 .. D-1; as "iterations" -D .. -1, so DMA(c) counts as issued in iteration c - D
#pragma unroll
    for (int c = 0; c < D; ++c)
        if (c < T) {
            issue(c);
            if (c >= 1) dma_hist[D - 1 - c] = GM_DMA_PER_CHUNK;     // iteration c - D = -(D - c): slot (−1) − (c − D) = D-1-c
        }

    unsigned char *const my_out = s_out + wave * GM_OUT_BYTES;
    int ch = 0, tile = 0;
    for (int t = 0; t < T; ++t) {
        {   // chunk t landed (this wave's part), then everybody's; the stage of chunk t-1 is free after the barrier
            int allowed = 0;
#pragma unroll
            for (int j = 0; j < D - 1; ++j) allowed += dma_hist[j];         // iterations t-1 .. t-D+1
#pragma unroll
            for (int j = 0; j < D; ++j) allowed += st_hist[j];              // iterations t-1 .. t-D
            gm_wait_allowed(allowed);
            __builtin_amdgcn_s_barrier();
        }
        int issued_now = 0;
        if (t + D < T) {
            issue(t + D);
            issued_now = GM_DMA_PER_CHUNK;
        }
        unsigned char *const st = smem + (t % NS) * GM_STAGE;
        if (PRO) {
            // normalise + activate the X chunk in place: 512 16-byte pieces, row = piece / 16
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = tid + 256 * j;
                const float2 ss = reinterpret_cast<const float2 *>(s_tab)[ch * GM_KC + (p >> 4)];
                uint4 *cell = reinterpret_cast<uint4 *>(st + GM_A_BYTES + p * 16);
                uint4 v = *cell;
                unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
                    lo = __builtin_fmaf(lo, ss.x, ss.y);
                    hi = __builtin_fmaf(hi, ss.x, ss.y);
                    if (pro_act == 1) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
                    else if (pro_act == 2) { lo = lo > 0.f ? lo : lo * pro_slope; hi = hi > 0.f ? hi : hi * pro_slope; }
                    w[e] = gm_pack_bf16(lo, hi);
                }
                *cell = make_uint4(w[0], w[1], w[2], w[3]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        // ---- 2 k-steps x (2 x 2) MFMAs ----
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            gm_bf16x8 xa[2], wb[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const gm_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (gm_s16x4 __attribute__((address_space(3))) *)(st + xoff[mi] + ks * 4096));
                const gm_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (gm_s16x4 __attribute__((address_space(3))) *)(st + xoff[mi] + ks * 4096 + 1024));
                xa[mi] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int ri = 0; ri < 2; ++ri) wb[ri] = *reinterpret_cast<const gm_bf16x8 *>(st + woff[ri][ks]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ri = 0; ri < 2; ++ri)
                    acc[mi][ri] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[mi], wb[ri], acc[mi][ri], 0, 0, 0);
        }
        int stored_now = 0;
        if (++ch == nch) {
            ch = 0;
            // ---- epilogue of one output tile: round, statistics, transpose through LDS, 16-byte row stores ----
            const int64_t mcol = col0 + (int64_t)tile * GM_T + wm * 64;
#pragma unroll
            for (int ri = 0; ri < 2; ++ri) {
                if (rt_valid[ri]) {
                    if (STATS && tile == 0) {
                        // shift = the row's first rounded output of this wave (lane l31 of the lower half holds it)
                        const unsigned pk = gm_pack_bf16(acc[0][ri][0], 0.f);
                        sShift[ri] = __shfl(__uint_as_float(pk << 16), l31);
                    }
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            const unsigned p0 = gm_pack_bf16(acc[mi][ri][4 * rg + 0], acc[mi][ri][4 * rg + 1]);
                            const unsigned p1 = gm_pack_bf16(acc[mi][ri][4 * rg + 2], acc[mi][ri][4 * rg + 3]);
                            if (STATS) {
                                const float v[4] = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u),
                                                    __uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float d = v[e] - sShift[ri];
                                    sS[ri] += d;
                                    sQ[ri] = __builtin_fmaf(d, d, sQ[ri]);
                                }
                            }
                            // m = mi*32 + 8 rg + 4 half + (0..3): 16-byte piece mi*4 + rg, 8-byte half `half`
                            const int p16 = (mi * 4 + rg) ^ (l31 & 7);
                            *reinterpret_cast<uint2 *>(my_out + l31 * 128 + p16 * 16 + half * 8) = make_uint2(p0, p1);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[mi][ri][4 * rg + e] = 0.0f;
                        }
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = it * 8 + (lane >> 3), p16 = lane & 7;
                        const uint4 v = *reinterpret_cast<const uint4 *>(my_out + row * 128 + ((p16 ^ (row & 7)) << 4));
                        const int r = r0 + wr * 64 + ri * 32 + row;
                        *reinterpret_cast<uint4 *>(Y + (size_t)r * M + mcol + p16 * 8) = v;
                    }
                }
            }
            stored_now = stores_per_epi;
            ++tile;
        }
        // shift the issue history by one iteration
#pragma unroll
        for (int j = D - 1; j > 0; --j) dma_hist[j] = dma_hist[j - 1];
        dma_hist[0] = issued_now;
#pragma unroll
        for (int j = D; j > 0; --j) st_hist[j] = st_hist[j - 1];
        st_hist[0] = stored_now;
    }
    if (STATS) {
        // per row: this wave's (sum, sum of squares, shift) over its 64-column share of every tile of the range
#pragma unroll
        for (int ri = 0; ri < 2; ++ri) {
            const float s = sS[ri] + __shfl_xor(sS[ri], 32), q = sQ[ri] + __shfl_xor(sQ[ri], 32);
            const int r = r0 + wr * 64 + ri * 32 + l31;
            if (half == 0 && rt_valid[ri]) {
                float *pp = part + ((((size_t)grp * Rg + r) * views + view) * P + (rloc * 2 + wm)) * 3;
                pp[0] = s;
                pp[1] = q;
                pp[2] = sShift[ri];
            }
        }
    }
}

// Statistics of row c, view v from the P partials of the GEMM (n_p columns each): Chan's combination of
// (count, mean, M2) in double, fixed order -> mean, invstd saved for backward; (scale, shift) for the affine
// kernel / the next GEMM's PRO: z = act(y * scale + shift); running statistics advance once per view, in order.
__global__ __launch_bounds__(64) void bn_finalize_kernel(const float *__restrict__ part, int C, int views, int P,
                                                         int tiles_range, int col_tiles_view, int64_t Mg,
                                                         const float *__restrict__ pre_bias,
                                                         const float *__restrict__ gamma, const float *__restrict__ beta,
                                                         float eps, float momentum, float *__restrict__ running_mean,
                                                         float *__restrict__ running_var, float *__restrict__ save_mean,
                                                         float *__restrict__ save_invstd, float2 *__restrict__ tab) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    const float pb = pre_bias ? pre_bias[c] : 0.0f;
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
    for (int v = 0; v < views; ++v) {
        double n = 0.0, mean = 0.0, m2 = 0.0;
        for (int p = 0; p < P; ++p) {
            const float *pp = part + (((size_t)c * views + v) * P + p) * 3;
            const int range = p >> 1;
            const int tiles = (tiles_range < col_tiles_view - range * tiles_range) ? tiles_range
                                                                                   : col_tiles_view - range * tiles_range;
            const double np = 64.0 * tiles;
            const double S = pp[0], Q = pp[1], sh = pp[2];
            const double mp = sh + S / np, m2p = Q - S * S / np;
            const double nn = n + np, delta = mp - mean;
            mean += delta * (np / nn);
            m2 += m2p + delta * delta * (n * np / nn);
            n = nn;
        }
        double var = m2 / n;
        if (var < 0.0) var = 0.0;
        const float meanf = (float)mean + pb;                       // statistics of y + conv bias
        const float invstd = 1.0f / sqrtf((float)var + eps);
        save_mean[c * views + v] = meanf;
        save_invstd[c * views + v] = invstd;
        const float g = gamma[c] * invstd;
        tab[(size_t)c * views + v] = make_float2(g, beta[c] + (pb - meanf) * g);
        const float unbiased = Mg > 1 ? (float)(m2 / (n - 1.0)) : (float)var;
        rm = (1.0f - momentum) * rm + momentum * meanf;
        rv = (1.0f - momentum) * rv + momentum * unbiased;
    }
    if (running_mean) {
        running_mean[c] = rm;
        running_var[c] = rv;
    }
}

// eval mode: (scale, shift) from the running statistics
__global__ __launch_bounds__(64) void bn_eval_tab_kernel(int C, int views, const float *__restrict__ pre_bias,
                                                         const float *__restrict__ gamma, const float *__restrict__ beta,
                                                         float eps, const float *__restrict__ running_mean,
                                                         const float *__restrict__ running_var,
                                                         float *__restrict__ save_mean, float *__restrict__ save_invstd,
                                                         float2 *__restrict__ tab) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    const float pb = pre_bias ? pre_bias[c] : 0.0f;
    const float mean = running_mean[c], invstd = 1.0f / sqrtf(running_var[c] + eps);
    const float g = gamma[c] * invstd;
    for (int v = 0; v < views; ++v) {
        save_mean[c * views + v] = mean;
        save_invstd[c * views + v] = invstd;
        tab[(size_t)c * views + v] = make_float2(g, beta[c] + (pb - mean) * g);
    }
}

// z = act(y * scale + shift) [+ residual] over rows of bf16 (C, M); 4 x 16-byte vectors in flight per thread
__global__ __launch_bounds__(256) void bn_affine_bf16_kernel(const unsigned short *__restrict__ y, int64_t M, int64_t Mg,
                                                             int views, int chunks_view, int64_t chunk,
                                                             const float2 *__restrict__ tab,
                                                             const unsigned short *__restrict__ residual, int act,
                                                             float slope, unsigned short *__restrict__ out) {
    const int c = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int v = s / chunks_view, sl = s - v * chunks_view;
    const float2 ss = tab[(size_t)c * views + v];
    const unsigned short *row = y + (size_t)c * M, *rrow = residual ? residual + (size_t)c * M : nullptr;
    unsigned short *orow = out + (size_t)c * M;
    const int64_t vend = (int64_t)(v + 1) * Mg;
    const int64_t lo = (int64_t)v * Mg + (int64_t)sl * chunk, hi = (lo + chunk < vend) ? lo + chunk : vend;
    constexpr int U = 4;
    const int64_t step = 256 * 8;
    auto one = [&](const uint4 &rx, const uint4 &rr, unsigned short *dst) {
        const unsigned w[4] = {rx.x, rx.y, rx.z, rx.w}, q[4] = {rr.x, rr.y, rr.z, rr.w};
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = __builtin_fmaf(__uint_as_float(w[e] << 16), ss.x, ss.y);
            float b = __builtin_fmaf(__uint_as_float(w[e] & 0xffff0000u), ss.x, ss.y);
            if (act == 1) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
            else if (act == 2) { a = a > 0.f ? a : a * slope; b = b > 0.f ? b : b * slope; }
            if (rrow) { a += __uint_as_float(q[e] << 16); b += __uint_as_float(q[e] & 0xffff0000u); }
            o[e] = gm_pack_bf16(a, b);
        }
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 t = {o[0], o[1], o[2], o[3]};
        __builtin_nontemporal_store(t, reinterpret_cast<u4 *>(dst));
    };
    int64_t m = lo + (int64_t)tid * 8;
    for (; m + (U - 1) * step + 8 <= hi; m += U * step) {
        uint4 rx[U], rr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rx[u] = *reinterpret_cast<const uint4 *>(row + m + u * step);
            rr[u] = rrow ? *reinterpret_cast<const uint4 *>(rrow + m + u * step) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) one(rx[u], rr[u], orow + m + u * step);
    }
    for (; m + 8 <= hi; m += step) {
        const uint4 rx = *reinterpret_cast<const uint4 *>(row + m);
        const uint4 rr = rrow ? *reinterpret_cast<const uint4 *>(rrow + m) : make_uint4(0, 0, 0, 0);
        one(rx, rr, orow + m);
    }
}

static bool gemm_shape_ok(int R, int K, int groups, int64_t M, int views) {
    if (R <= 0 || K <= 0 || groups <= 0 || M <= 0 || views <= 0) return false;
    if (R % groups || K % groups) return false;
    const int Rg = R / groups, Kg = K / groups;
    return Rg % 32 == 0 && Kg % GM_KC == 0 && M % views == 0 && (M / views) % GM_T == 0;
}

}  // namespace grafp

extern "C" int grafp_conv1x1_gemm_supported(int R, int K, int groups, int64_t M, int views) {
    return grafp::gemm_shape_ok(R, K, groups, M, views) ? 1 : 0;
}

extern "C" int grafp_conv1x1_gemm_partials(int R, int K, int groups, int64_t M, int views) {
    using namespace grafp;
    if (!gemm_shape_ok(R, K, groups, M, views)) return 0;
    return gemm_plan(R / groups, K / groups, groups, M, views).P;
}

extern "C" int grafp_conv1x1_gemm_bf16(const void *w, const void *x, int R, int K, int groups, int64_t M, int views,
                                       const float *pro_tab, int pro_act, float pro_slope, void *y, float *stats_part,
                                       grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(w && x && y, "conv1x1_gemm: null pointer");
    GRAFP_REQUIRE(gemm_shape_ok(R, K, groups, M, views),
                  "conv1x1_gemm: unsupported shape R=%d K=%d groups=%d M=%lld views=%d (rows per group %% 32, K per "
                  "group %% 32, columns per view %% 128)", R, K, groups, (long long)M, views);
    GRAFP_REQUIRE((((uintptr_t)w | (uintptr_t)x | (uintptr_t)y) & 15) == 0, "conv1x1_gemm: operands must be 16-byte aligned");
    GRAFP_REQUIRE(pro_act >= 0 && pro_act <= 2, "conv1x1_gemm: bad activation %d", pro_act);
    const int Rg = R / groups, Kg = K / groups;
    const GemmPlan p = gemm_plan(Rg, Kg, groups, M, views);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(p.nblocks, 1, groups);
    const bool pro = pro_tab != nullptr, stats = stats_part != nullptr;
    const int ns = pro ? 3 : 4;
    const size_t lds = (size_t)ns * GM_STAGE + 4 * GM_OUT_BYTES + (pro ? (size_t)Kg * 8 : 0);
#define GM_LAUNCH(NS, PRO, STATS)                                                                                       \
    do {                                                                                                                \
        (void)hipFuncSetAttribute((const void *)conv1x1_gemm_kernel<NS, PRO, STATS>,                                    \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
        hipLaunchKernelGGL((conv1x1_gemm_kernel<NS, PRO, STATS>), grid, dim3(256), lds, s, (const unsigned short *)w,   \
                           Kg, (const unsigned short *)x, (unsigned short *)y, M, Rg, Kg, p.row_tiles, p.ranges_view,   \
                           p.tiles_range, p.col_tiles_view, views, (const float2 *)pro_tab, pro_act, pro_slope,         \
                           stats_part, p.P, p.nblocks);                                                                 \
    } while (0)
    if (pro && stats) GM_LAUNCH(3, true, true);
    else if (pro) GM_LAUNCH(3, true, false);
    else if (stats) GM_LAUNCH(4, false, true);
    else GM_LAUNCH(4, false, false);
#undef GM_LAUNCH
    GRAFP_CHECK_LAUNCH("conv1x1_gemm_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_bn_finalize(const float *stats_part, int C, int K, int groups, int64_t M, int views,
                                 const float *pre_bias, const float *gamma, const float *beta, float eps, float momentum,
                                 int training, float *running_mean, float *running_var, float *save_mean,
                                 float *save_invstd, float *tab, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(gamma && beta && save_mean && save_invstd && tab, "bn_finalize: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (!training) {
        GRAFP_REQUIRE(running_mean && running_var, "bn_finalize: eval mode needs running statistics");
        hipLaunchKernelGGL(bn_eval_tab_kernel, dim3((C + 63) / 64), dim3(64), 0, s, C, views, pre_bias, gamma, beta, eps,
                           running_mean, running_var, save_mean, save_invstd, (float2 *)tab);
        GRAFP_CHECK_LAUNCH("bn_eval_tab_kernel");
        return GRAFP_OK;
    }
    GRAFP_REQUIRE(stats_part, "bn_finalize: null partials");
    GRAFP_REQUIRE(gemm_shape_ok(C, K, groups, M, views), "bn_finalize: shape does not match a conv1x1_gemm launch");
    const GemmPlan p = gemm_plan(C / groups, K / groups, groups, M, views);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, s, stats_part, C, views, p.P, p.tiles_range,
                       p.col_tiles_view, M / views, pre_bias, gamma, beta, eps, momentum, running_mean, running_var,
                       save_mean, save_invstd, (float2 *)tab);
    GRAFP_CHECK_LAUNCH("bn_finalize_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_bn_affine_bf16(const void *y, int C, int64_t M, int views, const float *tab, const void *residual,
                                    int act, float slope, void *out, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(y && tab && out, "bn_affine: null pointer");
    GRAFP_REQUIRE(C > 0 && M > 0 && views > 0 && M % views == 0 && (M / views) % 8 == 0,
                  "bn_affine: bad shape C=%d M=%lld views=%d", C, (long long)M, views);
    GRAFP_REQUIRE((((uintptr_t)y | (uintptr_t)out | (uintptr_t)residual) & 15) == 0, "bn_affine: 16-byte alignment");
    const int64_t Mg = M / views;
    // ~2048 workgroups in total, >= 8192 elements per workgroup
    int chunks_view = (int)((2048 + (int64_t)C * views - 1) / ((int64_t)C * views));
    const int64_t max_chunks = (Mg + 8191) / 8192;
    if (chunks_view > max_chunks) chunks_view = (int)max_chunks;
    if (chunks_view < 1) chunks_view = 1;
    int64_t chunk = (Mg + chunks_view - 1) / chunks_view;
    chunk = (chunk + 7) / 8 * 8;
    chunks_view = (int)((Mg + chunk - 1) / chunk);
    hipLaunchKernelGGL(bn_affine_bf16_kernel, dim3(chunks_view * views, C), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short *)y, M, Mg, views, chunks_view, chunk, (const float2 *)tab,
                       (const unsigned short *)residual, act, slope, (unsigned short *)out);
    GRAFP_CHECK_LAUNCH("bn_affine_bf16_kernel");
    return GRAFP_OK;
}
