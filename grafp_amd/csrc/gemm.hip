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
//     registers over all tiles of a workgroup, merged over its wave columns in LDS and written as ONE (count, mean, M2)
//     partial per (row, workgroup): the BatchNorm that
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
#define GRAFP_STORE_FAMILY 2        // (common.h: GRAFP_ST_NT experiment builds)
#include "common.h"
#include "dma_ring.h"
#include "tuning.h"

namespace grafp {

constexpr int GM_KC = 32;                      // contraction per chunk
constexpr int GM_STORES_PER_RT = 4;            // 16-byte store instructions per wave and 32-row output tile

// Tile configurations: WR x WM waves, each wave RT x 2 MFMA tiles (32 RT rows x 64 columns); OR = rows of the per-wave
// output staging slab (32: a whole 32-row MFMA tile is transposed at once; 16: in two halves, which frees 2 KB of LDS
// per wave for a fourth ring stage).
//   S:    2 x 2 waves, RT 2 -> 128 x 128, 4 waves, 2 workgroups per CU: small problems and shapes no wider tile divides;
//   L:    2 x 4 waves, RT 4 -> 256 x 256, 8 waves, 1 workgroup per CU: >= 256 output rows per group.  The S tile moves
//         16 KB of operands through the LDS-DMA path per 1 MFLOP (64 flop/B) and that path saturates at about 9 TB/s
//         chip-wide (measured: 580-820 TFLOP/s on every stage 2-3 shape); the L tile moves 32 KB per 4 MFLOP;
//   N32 / N64 / N128: 1 x 8 waves side by side, RT 1 / 2 / 4 -> 32 / 64 / 128 rows x 512 columns (round 3): the layers
//         with <= 128 output rows per group (stages 0-1, every grouped convolution of stages 0-1, the data gradients
//         into C rows).  On the 256-row tile such a launch spent 2-8x its flops on rows that do not exist (s0 ffn2,
//         R = 64, K = 256: 275 GFLOP issued for 69 useful -- the kernel was MFMA-bound on garbage at 4.4 TB/s) and half
//         of every ring stage on a duplicated W tile; here a stage is [W R x 32 | X 32 x 512]: 32 KB of X per stage,
//         1 KB row segments, 96 KB of the operand that comes from HBM in flight per CU instead of 48.
template <int WR_, int WM_, int RT_, int OR_ = 32> struct GemmCfg {
    static constexpr int WR = WR_, WM = WM_, RT = RT_, OR = OR_, NW = WR_ * WM_, THREADS = 64 * NW;
    static constexpr int TR = WR_ * RT_ * 32, TN = WM_ * 64;
    static constexpr int NA = TR / 16, NB = TN / 16, NI = NA + NB;   // 1-KiB LDS-DMA instructions per chunk: W, X, all
    static constexpr int PER = (NI + NW - 1) / NW;                   // ... per wave (a wave's last one may not exist)
    static constexpr int A_BYTES = TR * GM_KC * 2, B_BYTES = GM_KC * TN * 2, STAGE = A_BYTES + B_BYTES;
    static constexpr int ROWB = TN * 2;                   // bytes per X tile row
    static constexpr int SLOTS = TN / 8;                  // 16-byte slots per X tile row
    static constexpr int RPI = 64 / SLOTS;                // X tile rows per DMA instruction
    static constexpr int OUT_BYTES = OR_ * 128;           // per-wave output staging: OR rows (r) x 64 m bf16
    static constexpr int PRO_PIECES = TN / (16 * NW);     // PRO: 16-byte X pieces per thread and chunk
    static_assert(SLOTS <= 64 && 64 % SLOTS == 0 && A_BYTES == NA * 1024 && B_BYTES == NB * 1024, "1-KiB DMA pieces");
    static_assert(OR_ == 32 || OR_ == 16, "staging slab");
};
typedef GemmCfg<2, 2, 2> GemmS;
typedef GemmCfg<2, 4, 4> GemmL;
typedef GemmCfg<1, 8, 1, 16> GemmN32;
typedef GemmCfg<1, 8, 2, 16> GemmN64;
typedef GemmCfg<1, 8, 4, 32> GemmN128;
// (Measured and dropped, round 3: M = 2 x 2 waves, RT 4 -> 256 x 128 with THREE 24 KB stages + 2 KB staging slabs = 80 KB,
//  i.e. two workgroups per CU that run out of phase by themselves, one's MFMA chain under the other's epilogue and store
//  drain: 10-15 % SLOWER than L on every stage 2-3 shape, 37.6 vs 34.0 ms over a step's products at 2048 clip-views.)
enum { GM_CFG_S = 0, GM_CFG_L = 1, GM_CFG_N32 = 2, GM_CFG_N64 = 3, GM_CFG_N128 = 4, GM_CFG_XL = 5 };   // XL: gemm_xl.h

struct GemmPlan {
    int cfg;                                     // tile configuration (GM_CFG_*)
    int tr, tn, wm, ns, row_tiles, col_tiles_view, ranges_view, tiles_range, nblocks, P;
    int wgs;                                     // resident-workgroup target the column ranges were cut for
};
// R rows per conv group, M columns, `views` column segments with separate statistics (a range never straddles two)
static GemmPlan gemm_plan(int Rg, int K, int groups, int64_t M, int views) {
    GemmPlan p;
    const int64_t Mg = M / views;
    // measured (tools/gemm_bench.py: every configuration on the forward, data-gradient and concatenated-operand shapes
    // of a step at 256, 512 and 2048 clip-views; profiles/r03_gemm_sweep.txt).  >= 256 rows per group: the large tile
    // for >= 512 rows; for 256 rows when the operand is deep (K >= 512: matrix-heavy), shallow (4K <= R: X is then read
    // once for all 256 rows of a write-bound product) or the rows are long (>= 2^18 columns per view: the tensors no
    // longer fit the Infinity Cache and the small tile's second pass over X goes to HBM).  (Round 2 also sent every
    // shape with >= 2^20 columns per view to the large tile; against the S tile that loses 3-10 % on seven of the ten
    // stage-0 shapes at 2048 clip-views -- 4 x the MFMA work on rows that do not exist -- and is gone.)
    // (with the S tile at two workgroups per CU by launch bounds -- 140-156 VGPRs instead of 200-228, 5-10 % faster -- a
    //  256-row shallow product, K < 128, moves to the large tile only from 2^19 columns per view)
    const bool large = (Rg >= 512 || (Rg >= 256 && (K >= 512 || Mg >= (1 << 19) || (K >= 128 && Mg >= (1 << 18))))) &&
                       Mg % GemmL::TN == 0;
    p.cfg = large ? GM_CFG_L : GM_CFG_S;
    // Round 4: the four-wave 256 x 256 tile (gemm_xl.h) where it measured faster than the eight-wave one at 2048 clip-views
    // (tools/gemm_bench.py --cfgs auto,L,XL; profiles/r04_gemm_xl_2048.txt): ungrouped products with 256 or 512 output rows
    // and >= 512 operand rows, -4 ... -12 % (deep contractions most: its main loop keeps the matrix pipe fed, and its
    // epilogue hides in the next tile).  With >= 1024 output rows or K <= 256 a tile is a short loop in front of 128 KB of
    // stores per workgroup: the product is bound by the vector-memory pipe (LDS-DMA and stores do not overlap) and the two
    // tiles tie or the eight-wave one wins by 3-6 %.
    if (large && groups == 1 && Rg % 256 == 0 && Rg <= 512 && K >= 512) p.cfg = GM_CFG_XL;
    // <= 128 rows per group: the 512-column tiles where they measured faster than both --
    //   N128 (65 ... 128 rows, ungrouped): from 512 operand rows at 2^17 columns per view, from 128 operand rows at 2^19
    //        (-15 ... -25 % on the stage-1 shapes at 2048 clip-views, -10 % on the deep ones at 512; the shallow ones at
    //        512 clip-views and R = 128, K = 64 stay on the S tile);
    //   N64 / N32 (<= 64 / <= 32 rows): from 2^16 columns per view (-10 ... -27 % at 256 and 512 clip-views); beyond
    //        2^20 columns per view only the ungrouped shapes with K = 64 or K >= 256 (the others tie or lose 5 %).
    const int64_t n_from = GRAFP_TUNE_INT("GRAFP_GEMM_N_FROM", 1 << 16);
    if (Rg <= 128 && Mg % GemmN64::TN == 0 && Mg >= n_from) {
        if (Rg > 64) {
            if (groups == 1 && ((K >= 512 && Mg >= 2 * n_from) || (K >= 128 && Mg >= 8 * n_from))) p.cfg = GM_CFG_N128;
        } else if (Mg < (1 << 20) || (groups == 1 && (K <= 64 || K >= 256))) {
            p.cfg = Rg <= 32 ? GM_CFG_N32 : GM_CFG_N64;
        }
    }
    const int force = GRAFP_TUNE_INT("GRAFP_GEMM_CFG", -1);                 // measurement builds only (tuning.h)
    if (force == GM_CFG_S || ((force == GM_CFG_L || force == GM_CFG_XL) && Mg % GemmL::TN == 0) ||
        (force >= GM_CFG_N32 && force <= GM_CFG_N128 && Mg % GemmN64::TN == 0))
        p.cfg = force;
    // the four-wave tile (gemm_xl.h) takes whole 256-row tiles and hides a tile's epilogue in the next tile's first four
    // chunks: rows per group a multiple of 256, at least 128 operand rows
    if (p.cfg == GM_CFG_XL && (Rg % 256 != 0 || K < 4 * GM_KC)) p.cfg = GM_CFG_L;
    static const int trs[6] = {GemmS::TR, GemmL::TR, GemmN32::TR, GemmN64::TR, GemmN128::TR, 256};
    static const int tns[6] = {GemmS::TN, GemmL::TN, GemmN32::TN, GemmN64::TN, GemmN128::TN, 256};
    p.tr = trs[p.cfg];
    p.tn = tns[p.cfg];
    p.wm = p.tn / 64;
    // ring depth: S keeps 2 workgroups per CU (80 KB each: 4 stages); the others are alone on their CU and take what
    // the 160 KB hold beside the staging slabs; 3 stages everywhere beside a PRO table (gemm_dispatch)
    p.ns = p.cfg == GM_CFG_N128 ? 3 : 4;
    const int per_cu = p.cfg == GM_CFG_S ? 2 : 1;
    p.row_tiles = (Rg + p.tr - 1) / p.tr;
    p.col_tiles_view = (int)(Mg / p.tn);
    const int nch = K / GM_KC;
    // two rounds of resident workgroups, but at least ~8 chunks per workgroup to amortise the pipeline fill
    // (four rounds once the launch streams >= 750 MB: -3 % over all layers at 2048 clip-views, nothing below;
    // swept 256 ... 8192 per workgroup-per-CU)
    const int wgs_dflt = (double)(Rg + K) * groups * (double)M * 2.0 >= 750e6 ? 1024 : 512;
    const int wgs = GRAFP_TUNE_INT("GRAFP_GEMM_WGS", 0) > 0 ? GRAFP_TUNE_INT("GRAFP_GEMM_WGS", 0) : wgs_dflt;
    p.wgs = wgs;
    int64_t want = (int64_t)(wgs * per_cu) / ((int64_t)p.row_tiles * groups * views);
    if (want < 1) want = 1;
    int tiles_range = (int)((p.col_tiles_view + want - 1) / want);
    const int min_tiles = (8 + nch - 1) / nch;
    if (tiles_range < min_tiles) tiles_range = min_tiles;
    if (tiles_range > p.col_tiles_view) tiles_range = p.col_tiles_view;
    p.tiles_range = tiles_range;
    p.ranges_view = (p.col_tiles_view + tiles_range - 1) / tiles_range;
    p.nblocks = p.row_tiles * p.ranges_view * views;
    p.P = p.ranges_view;              // one partial per column range: the wave columns of a workgroup are merged in LDS
    return p;
}

// Chan's pairwise combination of (count, mean, M2): set b appended to set a
__device__ __forceinline__ void gm_chan(float &n, float &mean, float &m2, float nb, float mb, float m2b) {
    const float nn = n + nb;
    if (nn > 0.0f) {
        const float delta = mb - mean, f = nb / nn;
        mean = __builtin_fmaf(delta, f, mean);
        m2 += m2b + delta * delta * (n * f);
        n = nn;
    }
}

// CAT: the operand is the row-wise concatenation [X (K1 rows); X2 (K - K1 rows)] of two tensors (never materialised):
// the data gradient of the first layer of a residual block takes the shortcut's gradient as extra operand rows
// against an identity block of the weight, dX = [W^T | I] [dY; dZ] -- the sum is rounded once and the separate
// gradient-accumulation add (two reads, one write of C x M) is gone.
// EPI: inference (eval-mode BatchNorm: scale and shift known before the product): z = act(bf16(W x) * scale + shift) is
// formed in the epilogue -- the same arithmetic on the same rounded value as bn_affine_bf16_kernel, so the result is
// bit-identical to GEMM + normalise pass, without writing and re-reading y (fingerprint generation: generate.py:34-57).
// OUTF32 (round 4): the accumulators leave as f32 (Y points to R x M floats): the product of the f32 "parity" mode, whose
// operands arrive as bf16 hi / lo planes and whose weight is [Wh | Wh | Wl] against the operand rows [Xh; Xl; Xh] (CAT) --
// three exact bf16 products per term, summed in the f32 accumulators: ~2^-16 relative, 2-3 x the library's f32 GEMM rate.
template <typename CFG, int NS, bool PRO, bool STATS, bool CAT = false, bool EPI = false, bool OUTF32 = false>
__global__ __launch_bounds__(CFG::THREADS, CFG::NW == 4 ? 2 : 1) void conv1x1_gemm_kernel(
    const unsigned short *__restrict__ A, int lda, const unsigned short *__restrict__ X, unsigned short *__restrict__ Y,
    int64_t M, int Rg, int K, int row_tiles, int ranges_view, int tiles_range, int col_tiles_view, int views,
    const float2 *__restrict__ pro_tab, int pro_act, float pro_slope, float *__restrict__ part, int P, int nblocks,
    const unsigned short *__restrict__ X2 = nullptr, int K1 = 0, const float2 *__restrict__ epi_tab = nullptr,
    int epi_act = 0, float epi_slope = 0.0f) {
    constexpr int D = NS - 1;                               // chunks in flight
    constexpr int RT = CFG::RT, TN = CFG::TN, ROWB = CFG::ROWB, PER = CFG::PER, OR = CFG::OR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *const s_out = smem + NS * CFG::STAGE;              // NW x OUT_BYTES
    unsigned char *const s_tab = s_out + CFG::NW * CFG::OUT_BYTES;    // PRO: K x float2 (scale, shift) of this view

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: everything derived stays in SGPRs
    const int wr = wave / CFG::WM, wm = wave % CFG::WM;
    const unsigned lds0 = (unsigned)(uintptr_t)(gm_lptr)smem;
    const int logical = xcd_remap(blockIdx.x, nblocks);
    const int rt = logical % row_tiles, range = logical / row_tiles;
    const int view = range / ranges_view, rloc = range - view * ranges_view;
    const int grp = blockIdx.z;
    const int tile0 = rloc * tiles_range;
    const int ntile = (tiles_range < col_tiles_view - tile0) ? tiles_range : col_tiles_view - tile0;
    const int nch = K / GM_KC;
    const int T = ntile * nch;
    const int r0 = rt * CFG::TR;
    const int64_t col0 = (int64_t)view * (M / views) + (int64_t)tile0 * TN;
    A += (size_t)grp * Rg * lda;
    X += (size_t)grp * K * M;
    Y += (size_t)grp * Rg * M;

    if (PRO) {       // this view's (scale, shift) per operand row -> LDS (ordinary loads: before any DMA is in flight)
        const float2 *src = pro_tab + ((size_t)grp * K) * views;
        for (int k = tid; k < K; k += CFG::THREADS) reinterpret_cast<float2 *>(s_tab)[k] = src[(size_t)k * views + view];
    }

    // ---- the chunk's NI LDS-DMA instructions (1 KiB each: 64 lanes x 16 bytes; W rows first, then X rows) are dealt to
    //      the waves round robin: this wave issues q = wave + j * NW.  LDS side lane-linear: stage + q KiB + lane * 16.
    // W piece q < NA: rows 16q .. 16q+15 (64 B each); slot' = lane & 3 holds source slot slot' ^ ((row >> 2) & 3)
    // X piece q - NA: RPI k-rows (ROWB bytes each); 16-byte slot s' = lane % SLOTS of its row holds source segment
    //          (s' >> 2) ^ (row & 3) (low two bits of the 64-byte segment index), piece s' & 3
    const unsigned short *src[PER], *src2[PER];
    bool is_w[PER];
    int my_dma = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int q = wave + j * CFG::NW;                    // scalar
        is_w[j] = q < CFG::NA;
        src2[j] = nullptr;
        if (q < CFG::NI) ++my_dma;
        if (is_w[j]) {
            int row = r0 + 16 * q + (lane >> 2);
            if (row > Rg - 1) row = Rg - 1;                  // rows beyond R: duplicates, never stored
            const int slot = (lane & 3) ^ ((lane >> 4) & 3);
            src[j] = A + (size_t)row * lda + slot * 8;
        } else {
            const int row = CFG::RPI * (q - CFG::NA) + lane / CFG::SLOTS;
            const int sl = lane % CFG::SLOTS;
            const int seg = (sl >> 2) ^ (row & 3);
            src[j] = X + (size_t)row * M + col0 + (seg * 4 + (sl & 3)) * 8;
            if (CAT) src2[j] = X2 + (src[j] - X);
        }
    }
    const int nch1 = CAT ? K1 / GM_KC : nch;
    int is_ch = 0;                                           // chunk-in-tile of the next chunk to issue
    auto issue = [&](int t) {
        const unsigned st = lds0 + (t % NS) * CFG::STAGE + wave * 1024;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (wave + j * CFG::NW < CFG::NI) {              // scalar: false only for the last piece of some waves
                const unsigned short *g;
                if (is_w[j]) g = src[j] + is_ch * GM_KC;
                else if (CAT && is_ch >= nch1) g = src2[j] + (size_t)(is_ch - nch1) * GM_KC * M;
                else g = src[j] + (size_t)is_ch * GM_KC * M;
                gm_dma16(g, st + j * (CFG::NW * 1024));
            }
        }
        if (++is_ch == nch) {
            is_ch = 0;
#pragma unroll
            for (int j = 0; j < PER; ++j)
                if (!is_w[j]) {
                    src[j] += TN;
                    if (CAT) src2[j] += TN;
                }
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
        xoff[mi] = CFG::A_BYTES + (8 * half + (i >> 2)) * ROWB + seg * 64 + (bytecol & 63);
    }
    // W fragment (B operand, j = r): row wr*32*RT + ri*32 + l31, 16-byte slot (2 ks + half) ^ ((row >> 2) & 3)
    int woff[RT][2];
#pragma unroll
    for (int ri = 0; ri < RT; ++ri)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row = wr * 32 * RT + ri * 32 + l31;
            woff[ri][ks] = row * 64 + (((2 * ks + half) ^ ((row >> 2) & 3)) << 4);
        }

    f32x16 acc[2][RT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < RT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    gm_f32x2 sS[RT], sQ[RT];                                 // even / odd elements: packed f32 arithmetic
    float sShift[RT];
    // output tiles of 32 rows this wave really owns (R is a multiple of 32)
    bool rt_valid[RT];
    int stores_per_epi = 0;
#pragma unroll
    for (int ri = 0; ri < RT; ++ri) {
        sS[ri] = sQ[ri] = gm_f32x2{0.0f, 0.0f};
        sShift[ri] = 0.0f;
        rt_valid[ri] = r0 + wr * 32 * RT + ri * 32 < Rg;
        stores_per_epi += rt_valid[ri] ? GM_STORES_PER_RT : 0;
    }

    float2 esc[RT];                                          // EPI: (scale, shift) of the lane's row of each 32-row tile
#pragma unroll
    for (int ri = 0; ri < RT; ++ri) {
        esc[ri] = make_float2(1.0f, 0.0f);
        if (EPI) {
            int r = r0 + wr * 32 * RT + ri * 32 + l31;
            if (r > Rg - 1) r = Rg - 1;
            esc[ri] = epi_tab[((size_t)grp * Rg + r) * views + view];
        }
    }

    // vector-memory operations issued AFTER the DMA of chunk t, by iteration: dma_hist[j] / st_hist[j] = issued in
    // iteration t-1-j (the wait for chunk t may leave exactly those in flight)
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
            if (c >= 1) dma_hist[D - 1 - c] = my_dma;
        }

    unsigned char *const my_out = s_out + wave * CFG::OUT_BYTES;
    // (Measured and dropped, round 3: the upper four waves of an eight-wave tile issuing their DMA pieces AFTER their
    //  MFMAs, so that on every SIMD one wave's issue phase runs under the other's products: 33.65 vs 33.39 ms over a step's
    //  products at 2048 clip-views, the deepest shape 15 % slower.  Likewise W stored chunk-major -- full 128-byte lines
    //  per DMA piece instead of 64-byte row pieces: -2.7 %, not worth a second weight layout.)
    int ch = 0, tile = 0;
    for (int t = 0; t < T; ++t) {
        {   // chunk t landed (this wave's part), then everybody's; the stage of chunk t-1 is free after the barrier
            int allowed = 0;
#pragma unroll
            for (int j = 0; j < D - 1; ++j) allowed += dma_hist[j];         // iterations t-1 .. t-D+1
#pragma unroll
            for (int j = 0; j < D; ++j) allowed += st_hist[j];              // iterations t-1 .. t-D
            gm_wait_allowed(__builtin_amdgcn_readfirstlane(allowed));
            __builtin_amdgcn_s_barrier();
        }
        int issued_now = 0;
        if (t + D < T) {
            issue(t + D);
            issued_now = my_dma;
        }
        unsigned char *const st = smem + (t % NS) * CFG::STAGE;
        // ---- 2 k-steps x (2 x RT) MFMAs ----
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            gm_bf16x8 xa[2], wb[RT];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const gm_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (gm_s16x4 __attribute__((address_space(3))) *)(st + xoff[mi] + ks * 16 * ROWB));
                const gm_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (gm_s16x4 __attribute__((address_space(3))) *)(st + xoff[mi] + ks * 16 * ROWB + 4 * ROWB));
                xa[mi] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            if (PRO) {
                // normalise + activate the fragment IN REGISTERS: its 8 elements are 8 consecutive operand rows
                // (16 ks + 8 half + 0..7) of the lane's column, so the 8 (scale, shift) pairs are the same for a whole
                // half-wave: four broadcast 16-byte table reads.  (Rounds 1-3 did this as an in-place pass over the
                // staged tile -- one more barrier per chunk, and ds_write traffic into a ring that LDS-DMA is filling:
                // on the two-workgroup tile that pass raced with the ring at 2048 clip-views, 0.6 % of the outputs wrong
                // and different from run to run.  Same arithmetic on the same values: the results are bit-identical.)
                const float4 *tp = reinterpret_cast<const float4 *>(s_tab) + (ch * GM_KC + ks * 16 + 8 * half) / 2;
                float4 t4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) t4[q] = tp[q];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    gm_u32x4 w = __builtin_bit_cast(gm_u32x4, xa[mi]);
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        float lo = __uint_as_float(w[d] << 16), hi = __uint_as_float(w[d] & 0xffff0000u);
                        lo = __builtin_fmaf(lo, t4[d].x, t4[d].y);
                        hi = __builtin_fmaf(hi, t4[d].z, t4[d].w);
                        if (pro_act == 1) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
                        else if (pro_act == 2) { lo = lo > 0.f ? lo : lo * pro_slope; hi = hi > 0.f ? hi : hi * pro_slope; }
                        w[d] = gm_pack_bf16(lo, hi);
                    }
                    xa[mi] = __builtin_bit_cast(gm_bf16x8, w);
                }
            }
#pragma unroll
            for (int ri = 0; ri < RT; ++ri) wb[ri] = *reinterpret_cast<const gm_bf16x8 *>(st + woff[ri][ks]);
#pragma unroll
            for (int ri = 0; ri < RT; ++ri)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    acc[mi][ri] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[mi], wb[ri], acc[mi][ri], 0, 0, 0);
        }
        int stored_now = 0;
        if (++ch == nch) {
            ch = 0;
            // ---- epilogue of one output tile: round, statistics, transpose through LDS, 16-byte row stores ----
            const int64_t mcol = col0 + (int64_t)tile * TN + wm * 64;
#pragma unroll
            for (int ri = 0; ri < RT; ++ri) {
                if (OUTF32) {
                    if (rt_valid[ri]) {
                        // lane (l31, half) holds row l31 of the 32, columns mi*32 + 8 rg + 4 half + (0..3): 16-byte pieces
                        float *y32 = reinterpret_cast<float *>(Y) + (size_t)(r0 + wr * 32 * RT + ri * 32 + l31) * M + mcol;
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                            for (int rg = 0; rg < 4; ++rg) {
                                const f32x4 v = {acc[mi][ri][4 * rg + 0], acc[mi][ri][4 * rg + 1], acc[mi][ri][4 * rg + 2],
                                                 acc[mi][ri][4 * rg + 3]};
                                GRAFP_ST_NT(v, reinterpret_cast<f32x4 *>(y32 + mi * 32 + 8 * rg + 4 * half));
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[mi][ri][4 * rg + e] = 0.0f;
                            }
                    }
                } else if (rt_valid[ri]) {
                    if (STATS && tile == 0) {
                        // shift = the row's first rounded output of this wave (lane l31 of the lower half holds it)
                        const unsigned pk = gm_pack_bf16(acc[0][ri][0], 0.f);
                        sShift[ri] = __shfl(__uint_as_float(pk << 16), l31);
                    }
                    // the shift as a REAL register pair, made once per row tile.  Left to itself hipcc keeps the RT shifts in
                    // adjacent registers and splats by operand selection (v_pk_add_f32 d, a, v[72:73] op_sel:[0,1] neg_lo
                    // neg_hi for the odd one); with that form, at 2^20+ columns, ONE accumulate of lanes 48-63 of the odd row
                    // tile took 0 instead of the shift in a few workgroups per launch -- identical y, (mean, M2) partials off
                    // by 1e-6, different from run to run, whichever way the shift had been broadcast (ds_bpermute,
                    // v_permlane32_swap, with or without wait states).  Repeated launches are bit-identical since.
                    gm_f32x2 sh = {sShift[ri], sShift[ri]};
                    if (STATS) asm volatile("" : "+v"(sh));
                    unsigned pk0[2][4], pk1[2][4];
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            unsigned p0 = gm_pack_bf16(acc[mi][ri][4 * rg + 0], acc[mi][ri][4 * rg + 1]);
                            unsigned p1 = gm_pack_bf16(acc[mi][ri][4 * rg + 2], acc[mi][ri][4 * rg + 3]);
                            if (EPI) {
                                float v[4] = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u),
                                              __uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    v[e] = __builtin_fmaf(v[e], esc[ri].x, esc[ri].y);
                                    if (epi_act == 1) v[e] = fmaxf(v[e], 0.f);
                                    else if (epi_act == 2) v[e] = v[e] > 0.f ? v[e] : v[e] * epi_slope;
                                }
                                p0 = gm_pack_bf16(v[0], v[1]);
                                p1 = gm_pack_bf16(v[2], v[3]);
                            }
                            if (STATS) {
                                // two elements per VALU instruction (v_pk_add_f32 / v_pk_fma_f32): the statistics
                                // are VALU work the MFMAs wait for -- 4 instructions per output element cost as
                                // much as the products themselves at K = 128
                                const gm_f32x2 da = gm_f32x2{__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u)} - sh;
                                const gm_f32x2 db = gm_f32x2{__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)} - sh;
                                sS[ri] += da;
                                sQ[ri] = __builtin_elementwise_fma(da, da, sQ[ri]);
                                sS[ri] += db;
                                sQ[ri] = __builtin_elementwise_fma(db, db, sQ[ri]);
                            }
                            pk0[mi][rg] = p0;
                            pk1[mi][rg] = p1;
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[mi][ri][4 * rg + e] = 0.0f;
                        }
                    // transpose through the wave's own slab, OR rows at a time (lane l31 holds row l31 of the 32)
#pragma unroll
                    for (int h = 0; h < 32 / OR; ++h) {
                        if (OR == 32 || (l31 >> 4) == h) {
#pragma unroll
                            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                                for (int rg = 0; rg < 4; ++rg) {
                                    // m = mi*32 + 8 rg + 4 half + (0..3): 16-byte piece mi*4 + rg, 8-byte half `half`
                                    const int p16 = (mi * 4 + rg) ^ (l31 & 7);
                                    *reinterpret_cast<uint2 *>(my_out + (l31 & (OR - 1)) * 128 + p16 * 16 + half * 8) =
                                        make_uint2(pk0[mi][rg], pk1[mi][rg]);
                                }
                        }
                        // convergent no-op: without it hipcc (ROCm 7.2) sinks the reads below INTO the divergent block
                        // above (seen in the .s: ds_read_b128 under the half-wave exec mask, skipped on execz) -- the
                        // rows of the second half came back half garbage
                        if (OR != 32) __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int it = 0; it < OR / 8; ++it) {
                            const int row = it * 8 + (lane >> 3), p16 = lane & 7;
                            const uint4 v = *reinterpret_cast<const uint4 *>(my_out + row * 128 + ((p16 ^ (row & 7)) << 4));
                            const int r = r0 + wr * 32 * RT + ri * 32 + h * OR + row;
                            const gm_u32x4 vv = {v.x, v.y, v.z, v.w};
                            // streaming (nt) stores of Y: -10 ... 15 % on the family against plain stores
                            GRAFP_ST_NT(vv, reinterpret_cast<gm_u32x4 *>(Y + (size_t)r * M + mcol + p16 * 8));
                        }
                    }
                }
            }
            stored_now = OUTF32 ? 2 * stores_per_epi : stores_per_epi;      // (8 stores per 32-row tile instead of 4)
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
        // per row: this wave's (sum, sum of squares, shift) over its 64-column share of every tile of the range ->
        // (count, mean, M2); the WM wave columns of the workgroup are merged through LDS (Chan, fixed order) so that the
        // consumer finds ONE partial per (row, range) -- its combine runs in every workgroup of the normalise pass
        float *s_st = reinterpret_cast<float *>(smem);                  // [WM][TR][2] (mean, M2): the ring is idle now
        const float np = 64.0f * (float)ntile;
        __syncthreads();
#pragma unroll
        for (int ri = 0; ri < RT; ++ri) {
            const float s1 = sS[ri].x + sS[ri].y, q1 = sQ[ri].x + sQ[ri].y;
            const float s = s1 + __shfl_xor(s1, 32), q = q1 + __shfl_xor(q1, 32);
            if (half == 0) {
                const int row = wr * 32 * RT + ri * 32 + l31;
                const float dm = s / np;
                s_st[(wm * CFG::TR + row) * 2 + 0] = sShift[ri] + dm;
                s_st[(wm * CFG::TR + row) * 2 + 1] = fmaxf(q - s * dm, 0.0f);
            }
        }
        __syncthreads();
        if (wm == 0 && half == 0) {
#pragma unroll
            for (int ri = 0; ri < RT; ++ri) {
                const int row = wr * 32 * RT + ri * 32 + l31;
                float n = 0.0f, mean = 0.0f, m2 = 0.0f;
#pragma unroll
                for (int w = 0; w < CFG::WM; ++w)
                    gm_chan(n, mean, m2, np, s_st[(w * CFG::TR + row) * 2], s_st[(w * CFG::TR + row) * 2 + 1]);
                const int r = r0 + row;
                if (rt_valid[ri]) {
                    float *pp = part + ((((size_t)grp * Rg + r) * views + view) * P + rloc) * 3;
                    pp[0] = 0.0f;                                       // (S, Q, shift) with S = 0: mean = shift, M2 = Q
                    pp[1] = m2;
                    pp[2] = mean;
                }
            }
        }
    }
}

}  // namespace grafp
#include "gemm_xl.h"
namespace grafp {

// Statistics of row c, view v from the P partials of the GEMM (n_p columns each): Chan's combination of
// (count, mean, M2) in a fixed order (lane-strided partials, then a 6-step butterfly: deterministic) -> mean, invstd
// saved for backward; (scale, shift) for the affine kernel / the next GEMM's PRO: z = act(y * scale + shift); running
// statistics advance once per view, in order.  One workgroup per row, one wave per view.  f32 is enough here: the
// partials are already centred on a sample of their own row, and the combination never subtracts large numbers.
// (count, mean, M2) of row c, view v from its P partials: every lane of the calling WAVE returns the same values
__device__ __forceinline__ void gm_row_stats(const float *__restrict__ part, int c, int v, int views, int P, int wm,
                                             int tiles_range, int col_tiles_view, int lane, float &n, float &mean,
                                             float &m2) {
    n = 0.0f, mean = 0.0f, m2 = 0.0f;
    for (int p = lane; p < P; p += 64) {
        const float *pp = part + (((size_t)c * views + v) * P + p) * 3;
        const int range = p;                 // one partial per column range (the workgroup merged its wave columns)
        const int tiles = (tiles_range < col_tiles_view - range * tiles_range) ? tiles_range
                                                                               : col_tiles_view - range * tiles_range;
        const float np = (float)wm * 64.0f * tiles, S = pp[0], Q = pp[1], sh = pp[2];
        const float dm = S / np;
        gm_chan(n, mean, m2, np, sh + dm, fmaxf(Q - S * dm, 0.0f));
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float nb = __shfl_xor(n, o), mb = __shfl_xor(mean, o), m2b = __shfl_xor(m2, o);
        if (lane & o) {          // both partners compute the SAME ordered combination (lower lane's set first)
            float n2 = nb, me2 = mb, q2 = m2b;
            gm_chan(n2, me2, q2, n, mean, m2);
            n = n2; mean = me2; m2 = q2;
        } else {
            gm_chan(n, mean, m2, nb, mb, m2b);
        }
    }
}

__global__ __launch_bounds__(512) void bn_finalize_kernel(const float *__restrict__ part, int C, int views, int P,
                                                          int wm, int tiles_range, int col_tiles_view, int64_t Mg,
                                                          const float *__restrict__ pre_bias,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          float eps, float momentum, float *__restrict__ running_mean,
                                                          float *__restrict__ running_var, float *__restrict__ save_mean,
                                                          float *__restrict__ save_invstd, float2 *__restrict__ tab) {
    __shared__ float2 s_stat[8];                 // (mean + bias, unbiased variance) per view
    const int c = blockIdx.x, lane = threadIdx.x & 63, v = threadIdx.x >> 6;
    const float pb = pre_bias ? pre_bias[c] : 0.0f;
    float n, mean, m2;
    gm_row_stats(part, c, v, views, P, wm, tiles_range, col_tiles_view, lane, n, mean, m2);
    if (lane == 0) {
        const float var = fmaxf(m2 / n, 0.0f);
        const float meanf = mean + pb;                              // statistics of y + conv bias
        const float invstd = 1.0f / sqrtf(var + eps);
        save_mean[c * views + v] = meanf;
        save_invstd[c * views + v] = invstd;
        const float g = gamma[c] * invstd;
        tab[(size_t)c * views + v] = make_float2(g, beta[c] + (pb - meanf) * g);
        s_stat[v] = make_float2(meanf, Mg > 1 ? m2 / (n - 1.0f) : var);
    }
    __syncthreads();
    if (threadIdx.x == 0 && running_mean) {
        float rm = running_mean[c], rv = running_var[c];
        for (int u = 0; u < views; ++u) {
            rm = (1.0f - momentum) * rm + momentum * s_stat[u].x;
            rv = (1.0f - momentum) * rv + momentum * s_stat[u].y;
        }
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

struct BnStatsArgs {            // what bn_finalize_kernel takes, for the affine kernel that finalises by itself
    const float *part, *pre_bias, *gamma, *beta;
    float *running_mean, *running_var, *save_mean, *save_invstd;
    float2 *tab;
    int P, wm, tiles_range, col_tiles_view;
    float eps, momentum;
};

// z = act(y * scale + shift) [+ residual] over rows of bf16 (C, M); 4 x 16-byte vectors in flight per thread.
// STATS: no table comes in -- every workgroup combines the GEMM's partial sums of its (row, view) itself (one wave, a
// few microseconds, with the workgroup's first batch of vectors already requested), the first workgroup of a view
// saves mean / invstd / table, the first of a row advances the running statistics: no separate bn_finalize launch.
template <bool STATS>
__global__ __launch_bounds__(256) void bn_affine_bf16_kernel(const unsigned short *__restrict__ y, int64_t M, int64_t Mg,
                                                             int views, int chunks_view, int64_t chunk,
                                                             const float2 *__restrict__ tab, BnStatsArgs st,
                                                             const unsigned short *__restrict__ residual, int act,
                                                             float slope, unsigned short *__restrict__ out, int plain) {
    const int c = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int v = s / chunks_view, sl = s - v * chunks_view;
    const unsigned short *row = y + (size_t)c * M, *rrow = residual ? residual + (size_t)c * M : nullptr;
    unsigned short *orow = out + (size_t)c * M;
    const int64_t vend = (int64_t)(v + 1) * Mg;
    const int64_t lo = (int64_t)v * Mg + (int64_t)sl * chunk, hi = (lo + chunk < vend) ? lo + chunk : vend;
    constexpr int U = 4;
    const int64_t step = 256 * 8;
    int64_t m = lo + (int64_t)tid * 8;
    uint4 rx[U], rr[U];
    bool full = m + (U - 1) * step + 8 <= hi;
    auto fetch = [&]() {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rx[u] = GRAFP_LD_ONCE(2, reinterpret_cast<const uint4 *>(row + m + u * step));
            rr[u] = rrow ? GRAFP_LD_ONCE(8, reinterpret_cast<const uint4 *>(rrow + m + u * step)) : make_uint4(0, 0, 0, 0);
        }
    };
    if (full) fetch();                                        // in flight while the statistics are combined
    float2 ss;
    if (STATS) {
        __shared__ float2 s_ss, s_stat[4];
        const int wave = tid >> 6, lane = tid & 63;
        const bool row_leader = s == 0;                       // view 0, chunk 0: also owns the running statistics
        const int vv = row_leader ? wave : v;                 // the leader's waves take one view each
        if (row_leader ? wave < views : wave == 0) {
            const float pb = st.pre_bias ? st.pre_bias[c] : 0.0f;
            float n, mean, m2;
            gm_row_stats(st.part, c, vv, views, st.P, st.wm, st.tiles_range, st.col_tiles_view, lane, n, mean, m2);
            if (lane == 0) {
                const float var = fmaxf(m2 / n, 0.0f);
                const float meanf = mean + pb;
                const float invstd = 1.0f / sqrtf(var + st.eps);
                const float g = st.gamma[c] * invstd;
                const float2 t = make_float2(g, st.beta[c] + (pb - meanf) * g);
                if (vv == v) s_ss = t;
                if (row_leader) s_stat[vv] = make_float2(meanf, Mg > 1 ? m2 / (n - 1.0f) : var);
                if (sl == 0 && vv == v) {
                    st.save_mean[c * views + v] = meanf;
                    st.save_invstd[c * views + v] = invstd;
                    st.tab[(size_t)c * views + v] = t;
                }
            }
        }
        __syncthreads();
        ss = s_ss;
        if (row_leader && tid == 0 && st.running_mean) {
            float rm = st.running_mean[c], rv = st.running_var[c];
            for (int u = 0; u < views; ++u) {
                rm = (1.0f - st.momentum) * rm + st.momentum * s_stat[u].x;
                rv = (1.0f - st.momentum) * rv + st.momentum * s_stat[u].y;
            }
            st.running_mean[c] = rm;
            st.running_var[c] = rv;
        }
    } else {
        ss = tab[(size_t)c * views + v];
    }
    auto one = [&](const uint4 &ax, const uint4 &ar, unsigned short *dst) {
        const unsigned w[4] = {ax.x, ax.y, ax.z, ax.w}, q[4] = {ar.x, ar.y, ar.z, ar.w};
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
        if (plain) store16_hint(dst, __builtin_bit_cast(st_u32x4, t), true);      // (wave-uniform: bn_affine_launch)
        else GRAFP_ST_NT(t, reinterpret_cast<u4 *>(dst));
    };
    while (full) {
#pragma unroll
        for (int u = 0; u < U; ++u) one(rx[u], rr[u], orow + m + u * step);
        m += U * step;
        full = m + (U - 1) * step + 8 <= hi;
        if (full) fetch();
    }
    for (; m + 8 <= hi; m += step) {
        const uint4 ax = *reinterpret_cast<const uint4 *>(row + m);
        const uint4 ar = rrow ? *reinterpret_cast<const uint4 *>(rrow + m) : make_uint4(0, 0, 0, 0);
        one(ax, ar, orow + m);
    }
}

static bool gemm_shape_ok(int R, int K, int groups, int64_t M, int views) {
    if (R <= 0 || K <= 0 || groups <= 0 || M <= 0 || views <= 0) return false;
    if (R % groups || K % groups) return false;
    const int Rg = R / groups, Kg = K / groups;
    return Rg % 32 == 0 && Kg % GM_KC == 0 && M % views == 0 && (M / views) % GemmS::TN == 0;
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

extern "C" int grafp_conv1x1_gemm_plan(int R, int K, int groups, int64_t M, int views, int *info) {
    using namespace grafp;
    GRAFP_REQUIRE(info, "conv1x1_gemm_plan: null pointer");
    GRAFP_REQUIRE(gemm_shape_ok(R, K, groups, M, views), "conv1x1_gemm_plan: unsupported shape");
    const GemmPlan p = gemm_plan(R / groups, K / groups, groups, M, views);
    info[0] = p.cfg; info[1] = p.nblocks; info[2] = p.tiles_range; info[3] = p.P; info[4] = p.wgs;
    info[5] = p.tr; info[6] = p.tn; info[7] = p.row_tiles;
    return GRAFP_OK;
}

namespace grafp {
struct GemmArgs {
    bool out_f32 = false;                        // y: R x M floats (split-bf16 product of the f32 mode; concatenated operands)
    const unsigned short *w, *x, *x2;
    unsigned short *y;
    int lda, K1;
    int64_t M;
    int Rg, Kg, groups, views;
    const float2 *pro_tab;
    int pro_act;
    float pro_slope;
    float *part;
    const float2 *epi_tab = nullptr;
    int epi_act = 0;
    float epi_slope = 0.0f;
};
template <typename CFG, int NS, bool PRO, bool STATS, bool CAT, bool EPI = false, bool OUTF32 = false>
static void gemm_launch(const GemmPlan &p, const GemmArgs &a, hipStream_t s) {
    const size_t lds = (size_t)NS * CFG::STAGE + CFG::NW * CFG::OUT_BYTES + (PRO ? (size_t)a.Kg * 8 : 0);
    (void)hipFuncSetAttribute((const void *)conv1x1_gemm_kernel<CFG, NS, PRO, STATS, CAT, EPI, OUTF32>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((conv1x1_gemm_kernel<CFG, NS, PRO, STATS, CAT, EPI, OUTF32>), dim3(p.nblocks, 1, a.groups),
                       dim3(CFG::THREADS), lds, s, a.w, a.lda, a.x, a.y, a.M, a.Rg, a.Kg, p.row_tiles, p.ranges_view,
                       p.tiles_range, p.col_tiles_view, a.views, a.pro_tab, a.pro_act, a.pro_slope, a.part, p.P, p.nblocks,
                       a.x2, a.K1, a.epi_tab, a.epi_act, a.epi_slope);
}
template <bool STATS, bool CAT, bool EPI, int ABL = 0>
static void gemm_launch_xl(const GemmPlan &p, const GemmArgs &a, hipStream_t s) {
    (void)hipFuncSetAttribute((const void *)conv1x1_gemm_xl_kernel<STATS, CAT, EPI, ABL>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmXL::LDS);
    hipLaunchKernelGGL((conv1x1_gemm_xl_kernel<STATS, CAT, EPI, ABL>), dim3(p.nblocks, 1, a.groups), dim3(GemmXL::THREADS),
                       GemmXL::LDS, s, a.w, a.lda, a.x, a.y, a.M, a.Rg, a.Kg, p.row_tiles, p.ranges_view, p.tiles_range,
                       p.col_tiles_view, a.views, a.part, p.P, p.nblocks, a.x2, a.K1, a.epi_tab, a.epi_act, a.epi_slope);
}
// plain / statistics / concatenated-operand forms of one tile configuration
template <typename CFG, int NS> static void gemm_launch_cfg(const GemmPlan &p, const GemmArgs &a, hipStream_t s) {
    if (a.out_f32) gemm_launch<CFG, NS, false, false, true, false, true>(p, a, s);
    else if (a.epi_tab) gemm_launch<CFG, NS, false, false, false, true>(p, a, s);
    else if (a.x2) gemm_launch<CFG, NS, false, false, true>(p, a, s);
    else if (a.part) gemm_launch<CFG, NS, false, true, false>(p, a, s);
    else gemm_launch<CFG, NS, false, false, false>(p, a, s);
}
static void gemm_dispatch(const GemmPlan &p, const GemmArgs &a, hipStream_t s) {
    if (a.pro_tab) {                                  // normalise-on-load: three stages beside the table
#define GM_PRO(CFG)                                                        \
    do {                                                                   \
        if (a.part) gemm_launch<CFG, 3, true, true, false>(p, a, s);       \
        else gemm_launch<CFG, 3, true, false, false>(p, a, s);             \
    } while (0)
        switch (p.cfg) {
        case GM_CFG_XL:                                  // (no normalise-on-load form of the four-wave tile: same tile grid)
        case GM_CFG_L: GM_PRO(GemmL); break;
        case GM_CFG_N32: GM_PRO(GemmN32); break;
        case GM_CFG_N64: GM_PRO(GemmN64); break;
        case GM_CFG_N128: GM_PRO(GemmN128); break;
        default: GM_PRO(GemmS); break;
        }
#undef GM_PRO
        return;
    }
    switch (p.cfg) {
    case GM_CFG_XL:
#ifdef GRAFP_MEASURE
        switch (a.part && !a.x2 && !a.epi_tab ? GRAFP_TUNE_INT("GRAFP_XL_ABL", 0) : 0) {      // tools/gemm_bench.py --abl
        case 1: gemm_launch_xl<true, false, false, 1>(p, a, s); return;
        case 2: gemm_launch_xl<true, false, false, 2>(p, a, s); return;
        case 3: gemm_launch_xl<true, false, false, 3>(p, a, s); return;
        case 5: gemm_launch_xl<true, false, false, 5>(p, a, s); return;
        case 7: gemm_launch_xl<true, false, false, 7>(p, a, s); return;
        case 8: gemm_launch_xl<true, false, false, 8>(p, a, s); return;
        case 10: gemm_launch_xl<true, false, false, 10>(p, a, s); return;
        case 11: gemm_launch_xl<true, false, false, 11>(p, a, s); return;
        case 13: gemm_launch_xl<true, false, false, 13>(p, a, s); return;
        case 14: gemm_launch_xl<true, false, false, 14>(p, a, s); return;
        default: break;
        }
#endif
        if (a.epi_tab || a.out_f32) gemm_launch_cfg<GemmL, 4>(p, a, s);   // affine epilogue, f32 output: the eight-wave tile (same grid)
        else if (a.x2) gemm_launch_xl<false, true, false>(p, a, s);
        else if (a.part) gemm_launch_xl<true, false, false>(p, a, s);
        else gemm_launch_xl<false, false, false>(p, a, s);
        break;
    case GM_CFG_L: gemm_launch_cfg<GemmL, 4>(p, a, s); break;       // four stages = all 160 KB (-2.7 % against three)
    case GM_CFG_N32: gemm_launch_cfg<GemmN32, 4>(p, a, s); break;
    case GM_CFG_N64: gemm_launch_cfg<GemmN64, 4>(p, a, s); break;
    case GM_CFG_N128: gemm_launch_cfg<GemmN128, 3>(p, a, s); break;
    default: gemm_launch_cfg<GemmS, 4>(p, a, s); break;
    }
}
}  // namespace grafp

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
    GRAFP_REQUIRE(!pro_tab || p.cfg != GM_CFG_N128 || Kg <= 1024, "conv1x1_gemm: normalise-on-load with <= 128 output rows takes at most 1024 operand rows");
    GemmArgs a;
    a.w = (const unsigned short *)w; a.x = (const unsigned short *)x; a.x2 = nullptr; a.y = (unsigned short *)y;
    a.lda = Kg; a.K1 = 0; a.M = M; a.Rg = Rg; a.Kg = Kg; a.groups = groups; a.views = views;
    a.pro_tab = (const float2 *)pro_tab; a.pro_act = pro_act; a.pro_slope = pro_slope; a.part = stats_part;
    gemm_dispatch(p, a, (hipStream_t)stream);
    GRAFP_CHECK_LAUNCH("conv1x1_gemm_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_conv1x1_gemm_affine_bf16(const void *w, const void *x, int R, int K, int groups, int64_t M, int views,
                                              const float *tab, int act, float slope, void *z, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(w && x && z && tab, "conv1x1_gemm_affine: null pointer");
    GRAFP_REQUIRE(gemm_shape_ok(R, K, groups, M, views), "conv1x1_gemm_affine: unsupported shape R=%d K=%d groups=%d M=%lld "
                  "views=%d", R, K, groups, (long long)M, views);
    GRAFP_REQUIRE((((uintptr_t)w | (uintptr_t)x | (uintptr_t)z) & 15) == 0, "conv1x1_gemm_affine: operands must be 16-byte aligned");
    GRAFP_REQUIRE(act >= 0 && act <= 2, "conv1x1_gemm_affine: bad activation %d", act);
    const int Rg = R / groups, Kg = K / groups;
    const GemmPlan p = gemm_plan(Rg, Kg, groups, M, views);
    GemmArgs a;
    a.w = (const unsigned short *)w; a.x = (const unsigned short *)x; a.x2 = nullptr; a.y = (unsigned short *)z;
    a.lda = Kg; a.K1 = 0; a.M = M; a.Rg = Rg; a.Kg = Kg; a.groups = groups; a.views = views;
    a.pro_tab = nullptr; a.pro_act = 0; a.pro_slope = 0.0f; a.part = nullptr;
    a.epi_tab = (const float2 *)tab; a.epi_act = act; a.epi_slope = slope;
    gemm_dispatch(p, a, (hipStream_t)stream);
    GRAFP_CHECK_LAUNCH("conv1x1_gemm_kernel (affine epilogue)");
    return GRAFP_OK;
}

extern "C" int grafp_conv1x1_gemm_cat_bf16(const void *w, const void *x1, int K1, const void *x2, int K2, int R, int64_t M,
                                           void *y, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(w && x1 && x2 && y, "conv1x1_gemm_cat: null pointer");
    const int K = K1 + K2;
    GRAFP_REQUIRE(K1 > 0 && K2 > 0 && K1 % GM_KC == 0 && gemm_shape_ok(R, K, 1, M, 1),
                  "conv1x1_gemm_cat: unsupported shape R=%d K1=%d K2=%d M=%lld", R, K1, K2, (long long)M);
    GRAFP_REQUIRE((((uintptr_t)w | (uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)y) & 15) == 0,
                  "conv1x1_gemm_cat: operands must be 16-byte aligned");
    const GemmPlan p = gemm_plan(R, K, 1, M, 1);
    GemmArgs a;
    a.w = (const unsigned short *)w; a.x = (const unsigned short *)x1; a.x2 = (const unsigned short *)x2;
    a.y = (unsigned short *)y; a.lda = K; a.K1 = K1; a.M = M; a.Rg = R; a.Kg = K; a.groups = 1; a.views = 1;
    a.pro_tab = nullptr; a.pro_act = 0; a.pro_slope = 0.0f; a.part = nullptr;
    gemm_dispatch(p, a, (hipStream_t)stream);
    GRAFP_CHECK_LAUNCH("conv1x1_gemm_kernel (cat)");
    return GRAFP_OK;
}

// ---- the f32 mode's products on the bf16 matrix cores (round 4) ----
namespace grafp {
// v -> hi = bf16(v), lo = bf16(v - hi): |v - hi - lo| <= 2^-17 |v|; 8 elements per thread, rows stay rows
__global__ __launch_bounds__(256) void split_planes_kernel(const float *__restrict__ x, int64_t n,
                                                           unsigned short *__restrict__ hi, unsigned short *__restrict__ lo) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const float4 a = *reinterpret_cast<const float4 *>(x + i), b = *reinterpret_cast<const float4 *>(x + i + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = gm_pack_bf16(v[2 * e], v[2 * e + 1]);
        const float r0 = v[2 * e] - __uint_as_float(h[e] << 16), r1 = v[2 * e + 1] - __uint_as_float(h[e] & 0xffff0000u);
        l[e] = gm_pack_bf16(r0, r1);
    }
    *reinterpret_cast<uint4 *>(hi + i) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4 *>(lo + i) = make_uint4(l[0], l[1], l[2], l[3]);
}
}  // namespace grafp

extern "C" int grafp_split_bf16_planes(const float *x, int64_t n, void *hi, void *lo, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(x && hi && lo && n > 0 && n % 8 == 0, "split_bf16_planes: null pointer or n %% 8 != 0");
    GRAFP_REQUIRE((((uintptr_t)x | (uintptr_t)hi | (uintptr_t)lo) & 15) == 0, "split_bf16_planes: 16-byte alignment");
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n,
                       (unsigned short *)hi, (unsigned short *)lo);
    GRAFP_CHECK_LAUNCH("split_planes_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_conv1x1_gemm_split_f32(const void *w3, const void *x_planes, int R, int K, int64_t M, float *y,
                                            grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(w3 && x_planes && y, "conv1x1_gemm_split: null pointer");
    GRAFP_REQUIRE(K > 0 && K % GM_KC == 0 && gemm_shape_ok(R, 3 * K, 1, M, 1),
                  "conv1x1_gemm_split: unsupported shape R=%d K=%d M=%lld", R, K, (long long)M);
    GRAFP_REQUIRE((((uintptr_t)w3 | (uintptr_t)x_planes | (uintptr_t)y) & 15) == 0, "conv1x1_gemm_split: 16-byte alignment");
    GemmPlan p = gemm_plan(R, 3 * K, 1, M, 1);
    GemmArgs a;
    a.out_f32 = true;
    a.w = (const unsigned short *)w3;
    a.x = (const unsigned short *)x_planes;                              // rows [Xh; Xl]
    a.x2 = (const unsigned short *)x_planes;                             // ... then Xh again
    a.y = (unsigned short *)y; a.lda = 3 * K; a.K1 = 2 * K; a.M = M; a.Rg = R; a.Kg = 3 * K; a.groups = 1; a.views = 1;
    a.pro_tab = nullptr; a.pro_act = 0; a.pro_slope = 0.0f; a.part = nullptr;
    gemm_dispatch(p, a, (hipStream_t)stream);
    GRAFP_CHECK_LAUNCH("conv1x1_gemm_kernel (split f32)");
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
    GRAFP_REQUIRE(views <= 8, "bn_finalize: at most 8 views");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64 * views), 0, s, stats_part, C, views, p.P, p.wm, p.tiles_range,
                       p.col_tiles_view, M / views, pre_bias, gamma, beta, eps, momentum, running_mean, running_var,
                       save_mean, save_invstd, (float2 *)tab);
    GRAFP_CHECK_LAUNCH("bn_finalize_kernel");
    return GRAFP_OK;
}

static int bn_affine_launch(const void *y, int C, int64_t M, int views, const float *tab, const grafp::BnStatsArgs *st,
                            const void *residual, int act, float slope, void *out, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(y && out && (tab || st), "bn_affine: null pointer");
    GRAFP_REQUIRE(C > 0 && C <= 65535 && M > 0 && views > 0 && M % views == 0 && (M / views) % 8 == 0,
                  "bn_affine: bad shape C=%d M=%lld views=%d", C, (long long)M, views);
    GRAFP_REQUIRE((((uintptr_t)y | (uintptr_t)out | (uintptr_t)residual) & 15) == 0, "bn_affine: 16-byte alignment");
    GRAFP_REQUIRE(act >= 0 && act <= 2, "bn_affine: bad activation %d", act);
    const int64_t Mg = M / views;
    // ~2048 workgroups in total, >= 8192 elements per workgroup
    // (swept 1024 ... 262144, tools/bn_bench.py --affine: the plain form gains 10 % from 64 k small workgroups at 2048
    // clip-views, but the training form re-combines its row's partial statistics in every workgroup and loses 20 %;
    // a separate finalize launch + 64 k workgroups ties with this at 1024 pairs and loses at 256)
    const int wgs = GRAFP_TUNE_INT("GRAFP_AFFINE_WGS", 2048);
    int chunks_view = (int)((wgs + (int64_t)C * views - 1) / ((int64_t)C * views));
    const int64_t max_chunks = (Mg + 8191) / 8192;
    if (chunks_view > max_chunks) chunks_view = (int)max_chunks;
    if (chunks_view < 1) chunks_view = 1;
    int64_t chunk = (Mg + chunks_view - 1) / chunks_view;
    chunk = (chunk + 7) / 8 * 8;
    chunks_view = (int)((Mg + chunk - 1) / chunk);
    const dim3 grid(chunks_view * views, C);
    // store hint by the bytes written, as for the BatchNorm backward (bn.hip: bn_plain_stores): the normalised activation
    // is the operand of the next launches (graph build, max-relative, product).  Until round 6 these were streaming
    // stores like the products' (where they ARE worth 10-15 %); the whole-step A/B (tools/step_env_graph_ab.py,
    // profiles/r06_e_affine_plain_threshold*.txt; tensors up to 70 / 140 / 280 MB / all plain) says otherwise for this
    // kernel at every size: 128 pairs -1.4 / -1.8 / -1.8 %, 256 pairs -1.4 / -1.2 / -1.7 %, 512 pairs 0 / -0.8 / -1.0
    // / -0.9 %, 1024 pairs -0.1 (280 MB) ... -0.3 % (all) -- plain everywhere (the knob stays for measurement builds)
    const int plain = (size_t)C * (size_t)M * 2 <= ((size_t)GRAFP_TUNE_INT("GRAFP_AFFINE_PLAIN_MAX_MB", 1 << 20) << 20) ? 1 : 0;
    if (st)
        hipLaunchKernelGGL(bn_affine_bf16_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short *)y,
                           M, Mg, views, chunks_view, chunk, (const float2 *)nullptr, *st,
                           (const unsigned short *)residual, act, slope, (unsigned short *)out, plain);
    else
        hipLaunchKernelGGL(bn_affine_bf16_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short *)y,
                           M, Mg, views, chunks_view, chunk, (const float2 *)tab, BnStatsArgs{},
                           (const unsigned short *)residual, act, slope, (unsigned short *)out, plain);
    GRAFP_CHECK_LAUNCH("bn_affine_bf16_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_bn_affine_bf16(const void *y, int C, int64_t M, int views, const float *tab, const void *residual,
                                    int act, float slope, void *out, grafp_stream_t stream) {
    return bn_affine_launch(y, C, M, views, tab, nullptr, residual, act, slope, out, stream);
}

extern "C" int grafp_bn_finalize_affine_bf16(const void *y, const float *stats_part, int C, int K, int groups, int64_t M,
                                             int views, const float *pre_bias, const float *gamma, const float *beta,
                                             float eps, float momentum, float *running_mean, float *running_var,
                                             float *save_mean, float *save_invstd, float *tab, const void *residual,
                                             int act, float slope, void *out, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(stats_part && gamma && beta && save_mean && save_invstd && tab, "bn_finalize_affine: null pointer");
    GRAFP_REQUIRE(views >= 1 && views <= 4, "bn_finalize_affine: 1..4 views (one wave each in the row's first workgroup)");
    GRAFP_REQUIRE(gemm_shape_ok(C, K, groups, M, views), "bn_finalize_affine: shape does not match a conv1x1_gemm launch");
    const GemmPlan p = gemm_plan(C / groups, K / groups, groups, M, views);
    BnStatsArgs st;
    st.part = stats_part; st.pre_bias = pre_bias; st.gamma = gamma; st.beta = beta;
    st.running_mean = running_mean; st.running_var = running_var; st.save_mean = save_mean; st.save_invstd = save_invstd;
    st.tab = (float2 *)tab; st.P = p.P; st.wm = p.wm; st.tiles_range = p.tiles_range; st.col_tiles_view = p.col_tiles_view;
    st.eps = eps; st.momentum = momentum;
    return bn_affine_launch(y, C, M, views, nullptr, &st, residual, act, slope, out, stream);
}
