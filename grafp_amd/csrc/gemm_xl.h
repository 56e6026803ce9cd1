// gemm_xl.h -- the 256 x 256 tile of the streaming GEMM (gemm.hip) on FOUR waves, one per SIMD, gfx950.  Round 4.
//
// Why another tile: on the 8-wave 256 x 256 tile (GemmL) the MFMA + LDS-read loop ALONE runs at half the matrix peak
// (stage-2 FFN, 2048 clip-views: 225 us = 1.22 PFLOP/s with the DMA and the epilogue compiled out): a chunk is only two
// k-steps, every k-step starts with fragment reads whose latency nothing covers (both waves of a SIMD stand in the same
// phase behind the one barrier per chunk), and the two waves' MFMAs come out of one pipe.  233 of 256 registers leave no
// room for a second set of fragments.  Here a wave owns 128 x 128 outputs (4 x 4 MFMA tiles: 256 accumulator registers,
// which the 512-register budget of one wave per SIMD lets the compiler keep in AGPRs) and
//   * fragment reads per flop are HALF the 8-wave tile's (a k-step reads 8 KB for 16 MFMAs instead of 6 KB for 8);
//   * the fragments of k-step s + 1 are requested BEFORE the MFMAs of k-step s, also ACROSS the chunk barrier: the
//     wait + barrier for chunk c + 1 sits in the middle of chunk c, between its two k-steps, so every fragment read has
//     16 MFMAs (512 cycles) of cover;
//   * the eight 1-KiB LDS-DMA pieces a wave issues per chunk are spread between the MFMAs of the second k-step (an LDS-DMA
//     issue stalls the wave for 60-200 cycles against the memory pipe's back-pressure; in front of the chunk, as on the
//     8-wave tile, all waves stall together and the matrix pipe idles);
//   * an output tile starts with MFMAs on a zero C operand instead of 256 register clears.
// Ring: four 32-KB stages [W 256 x 32 | X 32 x 256] (same layout and swizzles as GemmL), chunk c + 3 is issued when chunk
// c + 1 has landed; per-wave 8-KB output staging slab (32 rows x 128 columns); 160 KB of LDS, one workgroup per CU.
#pragma once
#include <utility>

namespace grafp {

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}) -- the accumulator
// registers below are NAMED in asm text, and an asm "i" operand must be a constant expression (not an unrolled loop index)
template <int... I, typename F>
__device__ __forceinline__ void xl_static_for_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void xl_static_for(F &&f) {
    xl_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// ---- the 256 accumulator registers: a[16 T .. 16 T + 15] = MFMA tile T = ri * 4 + mi of the wave, by NAME ----
// hipcc 7.2 cannot keep sixteen 16-register accumulators in place when they fill the AGPR file: with the builtin (and
// equally with asm operands tied "+a") it gives the tuples new homes at every control-flow merge and repairs the
// assignment with 200-430 v_accvgpr moves around every 16 MFMAs, 500-750 spilled registers (seen in the .s).  So the
// accumulators never exist as C++ values: every instruction that touches them is an asm statement naming its registers,
// XL_CLAIM_AGPRS makes the kernel descriptor allocate all 256, and tools/check_kernel_regs.py requires of every build
// that the compiler itself emits NO v_accvgpr_* instruction and spills nothing in these kernels (an AGPR is hipcc's
// favourite spill slot).  Hazards hipcc does not pad inside or around an asm statement (cdna_hip_programming.md 5.7):
//   * A / B operands come straight from ds_read: the compiler's own lgkmcnt wait in front of the statement covers them;
//   * the same accumulator is the C operand of the next MFMA on it 16 MFMAs later (whole-tuple accumulate chain: none);
//   * an accumulator is read back (v_accvgpr_read) only behind xl_mfma_drain() -- 16-pass MFMA result -> VALU read.
template <int T> __device__ __forceinline__ void xl_mfma(const gm_bf16x8 &a, const gm_bf16x8 &b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(16 * T), "i"(16 * T + 15));
}
template <int T> __device__ __forceinline__ void xl_mfma0(const gm_bf16x8 &a, const gm_bf16x8 &b) {      // C = 0
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, 0" ::"v"(a), "v"(b), "i"(16 * T), "i"(16 * T + 15));
}
__device__ __forceinline__ void xl_opaque(unsigned &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void xl_keep(const gm_bf16x8 &a, const gm_bf16x8 &b) { asm volatile("" ::"v"(a), "v"(b)); }
__device__ __forceinline__ void xl_mfma_drain() { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); }
// registers 4 G .. 4 G + 3 of tile T (four consecutive m of one row) -> two packed bf16 pairs, round to nearest even.
// ONE statement, so that nothing can be scheduled between the reads and the conversions: with the reads as statements of
// their own hipcc issued all of a tile's 256 v_accvgpr_read first and parked the f32 values -- in AGPRs (seen in the .s).
template <int T, int G> __device__ __forceinline__ void xl_acc_pack4(unsigned &p0, unsigned &p1) {
    unsigned t;
    asm volatile("v_accvgpr_read_b32 %0, a[%c3]\n\tv_accvgpr_read_b32 %2, a[%c4]\n\tv_cvt_pk_bf16_f32 %0, %0, %2\n\t"
                 "v_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\tv_cvt_pk_bf16_f32 %1, %1, %2"
                 : "=&v"(p0), "=&v"(p1), "=&v"(t)
                 : "i"(16 * T + 4 * G), "i"(16 * T + 4 * G + 1), "i"(16 * T + 4 * G + 2), "i"(16 * T + 4 * G + 3));
}
// Keeping hipcc OUT of the accumulator file.  A clobber list is not enough: on gfx90a+ the allocator treats AGPRs as
// ordinary homes for any value that is only copied or spilled ("AV" classes), so between two statements that merely
// CLOBBER a0-a255 it parks its own values there (seen in the .s: v_accvgpr_write of loop-invariant addresses and of the
// statistics sums into accumulators that were still being read).  So the whole file is handed to sixteen 16-register
// placeholder values that are defined by an empty statement at the top of the kernel and consumed by one at its end: for
// the allocator all 256 AGPRs are occupied from the first instruction to the last, whatever tuple it gave each
// placeholder, and the named-register statements above are the only code that ever touches them.  With the file full it
// has nowhere to move a placeholder either.  tools/check_kernel_regs.py still requires: no compiler v_accvgpr_*, no
// scratch, 256 AGPRs in the descriptor.
#define XL_HOLD_AGPRS(h)                                                                                                  \
    asm volatile("; a0-a255 held for the accumulators"                                                                    \
                 : "=a"(h[0]), "=a"(h[1]), "=a"(h[2]), "=a"(h[3]), "=a"(h[4]), "=a"(h[5]), "=a"(h[6]), "=a"(h[7]),        \
                   "=a"(h[8]), "=a"(h[9]), "=a"(h[10]), "=a"(h[11]), "=a"(h[12]), "=a"(h[13]), "=a"(h[14]), "=a"(h[15]))
#define XL_RELEASE_AGPRS(h)                                                                                               \
    asm volatile("; a0-a255 released" ::"a"(h[0]), "a"(h[1]), "a"(h[2]), "a"(h[3]), "a"(h[4]), "a"(h[5]), "a"(h[6]),       \
                 "a"(h[7]), "a"(h[8]), "a"(h[9]), "a"(h[10]), "a"(h[11]), "a"(h[12]), "a"(h[13]), "a"(h[14]), "a"(h[15]))

struct GemmXL {
    static constexpr int NW = 4, THREADS = 256, TR = 256, TN = 256, RT = 4, MT = 4, NS = 4;
    static constexpr int NA = 16, NB = 16;                       // 1-KiB LDS-DMA pieces per chunk: W, X
    static constexpr int PW = NA / NW, PX = NB / NW;             // ... per wave
    static constexpr int A_BYTES = TR * GM_KC * 2, B_BYTES = GM_KC * TN * 2, STAGE = A_BYTES + B_BYTES;
    static constexpr int ROWB = TN * 2, SLOTS = TN / 8, RPI = 64 / SLOTS;
    static constexpr int OUT_BYTES = 32 * 256;                   // per wave: 32 rows (r) x 128 m bf16
    static constexpr int STORES_PER_RT = 8;                      // 16-byte store instructions per wave and 32-row tile
    static constexpr size_t LDS = (size_t)NS * STAGE + NW * OUT_BYTES;
};

// ABL (measurement builds only, results are garbage): 1 = no W pieces, 2 = no stores, 4 = no X pieces, 8 = no MFMAs
template <bool STATS, bool CAT, bool EPI, int ABL = 0>
__global__ __launch_bounds__(GemmXL::THREADS, 1) void conv1x1_gemm_xl_kernel(
    const unsigned short *__restrict__ A, int lda, const unsigned short *__restrict__ X, unsigned short *__restrict__ Y,
    int64_t M, int Rg, int K, int row_tiles, int ranges_view, int tiles_range, int col_tiles_view, int views,
    float *__restrict__ part, int P, int nblocks, const unsigned short *__restrict__ X2, int K1,
    const float2 *__restrict__ epi_tab, int epi_act, float epi_slope) {
    using C = GemmXL;
    static_assert(!EPI, "the eval-mode affine epilogue stays on the eight-wave tile (gemm_dispatch)");
    constexpr int RT = C::RT, MT = C::MT, NS = C::NS, TN = C::TN, ROWB = C::ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *const s_out = smem + NS * C::STAGE;

    f32x16 agpr_hold[16];
    XL_HOLD_AGPRS(agpr_hold);
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int r0 = rt * C::TR;
    const int64_t col0 = (int64_t)view * (M / views) + (int64_t)tile0 * TN;
    A += (size_t)grp * Rg * lda;
    X += (size_t)grp * K * M;
    Y += (size_t)grp * Rg * M;

    // ---- LDS-DMA sources: W pieces q = wave + 4 j (rows 16 q .. 16 q + 15, 64 B each), X pieces p = wave + 4 j
    //      (k-rows 2 p, 2 p + 1, 512 B each); swizzles as in conv1x1_gemm_kernel ----
    //      (Rg is a multiple of 256: no row of a W piece lies beyond the matrix; piece j of a wave = piece 0 + a fixed step)
    const unsigned short *srcw, *srcx, *srcx2 = nullptr;
    {
        const int row = r0 + 16 * wave + (lane >> 2);
        const int slot = (lane & 3) ^ ((lane >> 4) & 3);
        srcw = A + (size_t)row * lda + slot * 8;
    }
    {
        const int row = C::RPI * wave + lane / C::SLOTS;
        const int sl = lane % C::SLOTS;
        const int seg = (sl >> 2) ^ (row & 3);                 // (row + RPI * NW * j) & 3 == row & 3: RPI * NW = 8
        srcx = X + (size_t)row * M + col0 + (seg * 4 + (sl & 3)) * 8;
        if (CAT) srcx2 = X2 + (srcx - X);
    }
    const int64_t w_step = (int64_t)16 * C::NW * lda, x_step = (int64_t)C::RPI * C::NW * M;   // elements per piece index
    const int nch1 = CAT ? K1 / GM_KC : nch;
    int is_ch = 0;                                           // chunk-in-tile of the chunk being issued
    unsigned is_st = lds0 + wave * 1024;                     // ... and its stage (LDS byte address of this wave's piece 0)
    int is_stage = 0;
    // piece j of the chunk being issued (j = 0 .. 7: four W pieces, then four X pieces); call in order
    auto issue_piece = [&](int j) __attribute__((always_inline)) {
        if (j < C::PW) {
            if (!(ABL & 1)) gm_dma16(srcw + (j * w_step + is_ch * GM_KC), is_st + j * (C::NW * 1024));
        } else if (ABL & 4) {
        } else {
            const int jx = j - C::PW;
            const unsigned short *g;
            if (CAT && is_ch >= nch1) g = srcx2 + (jx * x_step + (int64_t)(is_ch - nch1) * GM_KC * M);
            else g = srcx + (jx * x_step + (int64_t)is_ch * GM_KC * M);
            gm_dma16(g, is_st + C::A_BYTES + jx * (C::NW * 1024));
        }
    };
    auto issue_done = [&]() __attribute__((always_inline)) {                                 // after piece 7: advance to the next chunk
        if (++is_ch == nch) {
            is_ch = 0;
            srcx += TN;
            if (CAT) srcx2 += TN;
        }
        if (++is_stage == NS) is_stage = 0;
        is_st = lds0 + is_stage * C::STAGE + wave * 1024;
    };

    // ---- fragment read offsets inside a stage ----
    int xoff[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int i = lane & 15;
        const int bytecol = (wm * 128 + mi * 32 + 16 * ((lane >> 4) & 1) + 4 * (i & 3)) * 2;
        const int seg = (bytecol >> 6) ^ (i >> 2);
        xoff[mi] = C::A_BYTES + (8 * half + (i >> 2)) * ROWB + seg * 64 + (bytecol & 63);
    }
    int woff[2];                                             // row tile ri: + ri * 2048 ((row >> 2) & 3 is the same)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int row = wr * 128 + l31;
        woff[ks] = row * 64 + (((2 * ks + half) ^ ((row >> 2) & 3)) << 4);
    }

    gm_f32x2 sS[RT], sQ[RT];
    float sShift[RT];
#pragma unroll
    for (int ri = 0; ri < RT; ++ri) {
        sS[ri] = sQ[ri] = gm_f32x2{0.0f, 0.0f};
        sShift[ri] = 0.0f;
    }

    // ---- fragments.  X (the A operand): two sets (k-step parity), each fragment two 8-byte transpose reads (lo: k 0-3,
    //      hi: k 4-7 of the lane's half); the eight reads of the NEXT k-step's set go out one per MFMA gap.  W (the B
    //      operand): ONE set -- the MFMAs run row tile by row tile, so wb[ri] is free once its four MFMAs have been issued
    //      (an MFMA reads A / B in its first cycles, an LDS read lands >= 64 cycles after its issue) and the next k-step's
    //      wb[ri] is requested into the same registers right behind them, twelve MFMAs ahead of its use ----
    gm_bf16x8 xa0[MT], xa1[MT], wb[RT];
    auto x_read = [&](auto set_c, auto mi_c, const unsigned char *st) __attribute__((always_inline)) {   // fragment mi of X set SET
        constexpr int SET = decltype(set_c)::value, mi = decltype(mi_c)::value;
        const gm_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (gm_s16x4 __attribute__((address_space(3))) *)(st + xoff[mi] + SET * 16 * ROWB));
        const gm_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (gm_s16x4 __attribute__((address_space(3))) *)(st + xoff[mi] + SET * 16 * ROWB + 4 * ROWB));
        if constexpr (SET == 0) xa0[mi] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        else xa1[mi] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto w_read = [&](auto ks_c, auto ri_c, const unsigned char *st) __attribute__((always_inline)) {
        constexpr int KS = decltype(ks_c)::value, ri = decltype(ri_c)::value;
        wb[ri] = *reinterpret_cast<const gm_bf16x8 *>(st + woff[KS] + ri * 2048);
    };
    auto mfma = [&](auto set_c, auto t_c, auto first_c) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value, TT = decltype(t_c)::value, mi = TT % MT, ri = TT / MT;
        if constexpr (ABL & 8) xl_keep(SET == 0 ? xa0[mi] : xa1[mi], wb[ri]);
        else if constexpr (decltype(first_c)::value) xl_mfma0<TT>(SET == 0 ? xa0[mi] : xa1[mi], wb[ri]);
        else xl_mfma<TT>(SET == 0 ? xa0[mi] : xa1[mi], wb[ri]);
    };

    // ---- the output tile that has LEFT the accumulators: 128 x 128 per wave as packed bf16 pairs, 128 registers.  Phase A
    //      of the epilogue (exposed: v_accvgpr_read, round) fills it when a tile's last MFMA has drained; phase B --
    //      statistics, transpose through the wave's LDS slab, the 16-byte row stores -- rides the MFMA gaps of the NEXT
    //      tile's first four chunks, one 32-row slice per chunk: the VALU work runs beside the matrix pipe instead of in
    //      front of it, and a tile's 32 stores per wave leave spread over four chunks instead of as one burst (with every
    //      CU bursting at once the stores of the old form took ~5 us to drain, and the in-order vmcnt made the wave wait
    //      for that drain before it could see the DMA issued behind them: "no stores" ablation -30 %). ----
    unsigned pk[RT][MT * 4][2];
    int64_t pend_mcol = 0;
    // Slab addressing without a register per address: row l31 of the slab is 256 B = sixteen 16-byte pieces, piece G of row r
    // lives in slot G ^ (r & 15) (conflict-free both ways).  XOR distributes over the shift and the slab is 256-byte aligned,
    // so a group's write address is ONE lane constant XOR an immediate, a fetch address one lane constant XOR (it & 3) * 64
    // plus an immediate offset; the row stores take a wave-uniform 64-bit base (SGPRs) plus one 32-bit lane offset.  The
    // lane constants are made opaque once per chunk (xl_opaque): left visible, hipcc hoists all 16 + 8 + 32 x 2 derived
    // addresses out of the loops and parks them -- in the accumulator registers (seen in the .s).
    unsigned slab_w = (unsigned)(uintptr_t)(gm_lptr)(s_out + wave * C::OUT_BYTES) + l31 * 256 + half * 8 + ((l31 & 15) << 4);
    unsigned slab_r = (unsigned)(uintptr_t)(gm_lptr)(s_out + wave * C::OUT_BYTES) + (lane >> 4) * 256 +
                      (((lane & 15) ^ (lane >> 4)) << 4);
    unsigned y_lane = (unsigned)(((int64_t)(lane >> 4) * M + (lane & 15) * 8) * 2);       // bytes; rows it*4 + (lane >> 4)
    const int y_row0 = __builtin_amdgcn_readfirstlane(r0 + wr * 128);
    gm_f32x2 sh2 = {0.0f, 0.0f};
    // group G = mi * 4 + rg of slice RI: statistics of its four rounded values + its 8 bytes into the slab
    auto slice_write = [&](auto ri_c, auto g_c) __attribute__((always_inline)) {
        constexpr int ri = decltype(ri_c)::value, G = decltype(g_c)::value;
        unsigned p0 = pk[ri][G][0], p1 = pk[ri][G][1];
        if (STATS) {
            // (opaque here: otherwise hipcc unpacks all 128 pairs right behind phase A and parks the 256 floats in AGPRs)
            xl_opaque(p0);
            xl_opaque(p1);
            if constexpr (G == 0) {
                sh2 = gm_f32x2{sShift[ri], sShift[ri]};
                asm volatile("" : "+v"(sh2));                  // a real register pair (see conv1x1_gemm_kernel)
            }
            const gm_f32x2 da = gm_f32x2{__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u)} - sh2;
            const gm_f32x2 db = gm_f32x2{__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)} - sh2;
            sS[ri] += da;
            sQ[ri] = __builtin_elementwise_fma(da, da, sQ[ri]);
            sS[ri] += db;
            sQ[ri] = __builtin_elementwise_fma(db, db, sQ[ri]);
            // pinned HERE, in this MFMA gap: the sums are only needed at the end of the kernel, and hipcc sinks the whole
            // chain of a tile's 64 groups into one block behind phase A (512 live floats, parked in AGPRs -- seen in the .s)
            asm volatile("" : "+v"(sS[ri]), "+v"(sQ[ri]));
        }
        // m = mi*32 + 8 rg + 4 half + (0..3): 16-byte piece G of the row, 8-byte half `half`
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2 __attribute__((address_space(3))) *>(slab_w ^ (G * 16)) = u32x2{p0, p1};
    };
    gm_u32x4 sv[2];                                          // slab rows on their way to HBM (read one even gap ahead)
    // rows it * 4 + (lane >> 4) of the slab (it wave-uniform, 0 .. 7), piece lane & 15
    auto slice_fetch = [&](int slot, int it) __attribute__((always_inline)) {
        sv[slot & 1] = *reinterpret_cast<const gm_u32x4 __attribute__((address_space(3))) *>((slab_r ^ ((it & 3) * 64)) + it * 1024);
    };
    // the row store of slot `slot`: wave-uniform 64-bit base `ub` (rows it * 4 ... of the slice) + the lane's 32-bit offset
    auto slice_store = [&](int slot, int it, char *ub) __attribute__((always_inline)) {
        const gm_u32x4 vv = sv[slot & 1];
        if (!(ABL & 2) || (vv[0] == 0x12345678u && it == 7))
            GRAFP_ST_NT(vv, reinterpret_cast<gm_u32x4 *>(ub + y_lane));
    };
    const int64_t y_rowstep = 8 * M;                         // bytes between the row groups of two consecutive stores

    // ---- ONE uniform stream: every chunk waits for chunk c + 1, reads its first fragments and issues chunk c + 3 -- also
    //      the last three of the range, whose "next chunks" do not exist: their pieces are fetched again from the range's
    //      FIRST tile (valid memory, a few KB per workgroup) into stages nobody reads, and drained before the ring is reused
    //      or the workgroup ends.  No tail variant of the loop body: four instantiations of it (first tile / steady / last /
    //      only tile) made a 17 000-line function whose register assignment hipcc could not hold together -- it parked
    //      values in the accumulator registers at every seam. ----
    const unsigned short *const srcx_first = srcx, *const srcx2_first = srcx2;
    int issued = 0;
    auto issue_next = [&]() __attribute__((always_inline)) {            // after piece 7 of a chunk
        issue_done();
        if (++issued == T) {                                           // beyond the range: again from its first tile
            is_ch = 0;
            srcx = srcx_first;
            if (CAT) srcx2 = srcx2_first;
        }
    };
    // ---- prologue: chunks 0, 1, 2 in flight; chunk 0 landed for everybody; fragments of its first k-step requested ----
#pragma unroll
    for (int c = 0; c < NS - 1; ++c) {
#pragma unroll
        for (int j = 0; j < C::PW + C::PX; ++j) issue_piece(j);
        issue_next();
    }
    if (ABL) gm_wait_vm<0>();
    else gm_wait_vm<16>();
    __builtin_amdgcn_s_barrier();
    xl_static_for<MT>([&](auto i) __attribute__((always_inline)) { x_read(std::integral_constant<int, 0>{}, i, smem); });
    xl_static_for<RT>([&](auto ri) __attribute__((always_inline)) { w_read(std::integral_constant<int, 0>{}, ri, smem); });

    int stage = 0, st_prev = 0;
    bool pend = false;                                       // a tile is waiting in pk (false only during the first tile)
    // A tile's 32 row stores per wave are spread EVENLY over the next tile's chunks: slice s (32 rows) takes `cps` chunks,
    // the first of them also carries the slice's 16 slab writes, each of them `spc` = 8 / cps of its stores.  Why evenly:
    // HBM takes writes at ~4.6 TB/s; issued by every CU in the same few chunks of a tile (or, as the eight-wave tile does, in
    // one burst at its end) they arrive faster than that, the write queues fill, and a CU's vector-memory pipe is in order --
    // the LDS-DMA loads behind a blocked store wait with it ("no MFMA" ablation: stores + DMA 486 us = DMA alone 252 +
    // 234, nothing overlaps; the average write rate of the product is 2.1 TB/s).
    // (measured, 2048 clip-views: cps = nch / 4 -- the stores spread over the WHOLE tile -- is 3-8 % slower on every
    //  shape than one slice per chunk in the tile's first four chunks: the run-time store slots and their scalar address
    //  arithmetic cost more MFMA-gap issue slots than the even write rate gives back; the "no stores" ablation gains the
    //  same ~180 us either way -- 128 KB per CU and tile at the ~5.5 TB/s HBM takes writes, tools/microbench/vmem_pipe_bench)
    constexpr int cps = 1, spc = 8 / cps;
    // One chunk.  SL: the slice of the previous tile's phase B it carries (RT: none); SUB0: the slice's first chunk (slab
    // writes); ZERO: first chunk of the output tile (zero C operand); sub: chunk of the slice (stores sub * spc ...).
    // During the first tile pk is all zeros: the slices run, add zeros to the statistics and skip their stores.
    //   k-step 0: MFMAs on set 0 (requested a k-step ago); gaps: the reads of X set 1 / W, slice writes
    //   lgkmcnt(0) -- everything requested above is 16 MFMAs old; saying so keeps hipcc's own wait-count pass from guessing
    //       across the control flow (without it: lgkmcnt(3..0) in front of the next MFMAs, i.e. a wait for reads issued
    //       just above them, seen in the .s)
    //   wait for this wave's pieces of chunk c + 1, barrier: chunk c + 1 is visible, every wave is past chunk c - 1
    //   k-step 1: MFMAs on set 1; gaps: the reads of chunk c + 1's X set 0 / W, the eight pieces of chunk c + 3 (odd gaps),
    //       slice rows slab -> HBM (fetch in an even gap, store one even gap later)
    auto chunk = [&](auto sl_c, auto sub0_c, auto zero_c, int sub) __attribute__((always_inline)) {
        constexpr int SL = decltype(sl_c)::value;
        constexpr bool SLICE = SL < RT, SUB0 = decltype(sub0_c)::value;
        constexpr int SLI = SLICE ? SL : 0;
        if constexpr (SLICE) {
            xl_opaque(slab_w);
            xl_opaque(slab_r);
            xl_opaque(y_lane);
        }
        unsigned char *const st = smem + stage * C::STAGE;
        const int nstage = stage + 1 == NS ? 0 : stage + 1;
        xl_static_for<RT * MT>([&](auto t) __attribute__((always_inline)) {
            constexpr int TT = decltype(t)::value;
            mfma(std::integral_constant<int, 0>{}, t, zero_c);
            if constexpr (TT < MT) x_read(std::integral_constant<int, 1>{}, t, st);
            if constexpr (TT % MT == MT - 1) w_read(std::integral_constant<int, 1>{}, std::integral_constant<int, TT / MT>{}, st);
            if constexpr (SLICE && SUB0) slice_write(std::integral_constant<int, SLI>{}, t);
            __builtin_amdgcn_sched_barrier(0);
        });
        // everything requested in the gaps above is >= 12 MFMAs old EXCEPT the last gap's: the refresh of wb[3] (and, in a
        // slice's first chunk, the last slab write behind it).  Saying so explicitly keeps hipcc's own wait-count pass from
        // guessing across the loops (it put lgkmcnt(3..0) in front of the next MFMAs: a wait for reads issued just above
        // them); lgkmcnt(0) here would wait for that last read -- a full LDS latency per k-step, in front of the barrier.
        if constexpr (SLICE && SUB0) __builtin_amdgcn_s_waitcnt(0xc27f);      // lgkmcnt(2)
        else __builtin_amdgcn_s_waitcnt(0xc17f);                             // lgkmcnt(1)
        // in flight behind chunk c + 1's pieces: chunk c + 2's (8) and the stores of the previous chunk (the last store of
        // the chunk before that lies between them: counted as older -- conservative).  EXACT counts: a wait that names
        // fewer waits for a store issued a microsecond ago to be acknowledged by HBM.
        if (ABL) gm_wait_vm<0>();
        else if (st_prev == 0) gm_wait_vm<8>();
        else if (st_prev == 1) gm_wait_vm<9>();
        else if (st_prev == 2) gm_wait_vm<10>();
        else if (st_prev == 4) gm_wait_vm<12>();
        else gm_wait_vm<16>();
        __builtin_amdgcn_s_barrier();
        unsigned char *const stn = smem + nstage * C::STAGE;
        const int it0 = sub * spc;
        // (a running pointer, advanced store by store: computed per slot from `it`, hipcc precomputes all eight 64-bit
        //  bases of all four slices and spills scalar registers into VGPR lanes by the hundred)
        char *ub = reinterpret_cast<char *>(Y) + ((int64_t)(y_row0 + SLI * 32 + it0 * 4) * M + pend_mcol) * 2;
        xl_static_for<RT * MT>([&](auto t) __attribute__((always_inline)) {
            constexpr int TT = decltype(t)::value;
            mfma(std::integral_constant<int, 1>{}, t, std::false_type{});
            if constexpr (TT < MT) x_read(std::integral_constant<int, 0>{}, t, stn);
            if constexpr (TT % MT == MT - 1) w_read(std::integral_constant<int, 0>{}, std::integral_constant<int, TT / MT>{}, stn);
            if constexpr (SLICE && !(TT & 1)) {
                constexpr int J = TT / 2;                    // store slot of this even gap: fetch J, store J - 1
                if constexpr (J >= 1) {
                    if (pend && J - 1 < spc) slice_store(J - 1, it0 + J - 1, ub);
                    ub += y_rowstep;
                }
                if (J < spc) slice_fetch(J, it0 + J);
            }
            if constexpr (TT & 1) issue_piece(TT >> 1);
            if constexpr (SLICE && TT == RT * MT - 1) {
                if (pend && spc == 8) slice_store(7, it0 + 7, ub);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        issue_next();
        __builtin_amdgcn_s_waitcnt(0xc17f);                  // lgkmcnt(1): chunk c + 1's first fragments but wb[3] (last gap)
        st_prev = (SLICE && pend && !(ABL & 2)) ? spc : 0;
        stage = nstage;
    };
    // phase A: accumulators -> rounded bf16 pairs in registers
    auto phase_a = [&](int tile) __attribute__((always_inline)) {
        xl_mfma_drain();
        xl_static_for<RT>([&](auto rit) __attribute__((always_inline)) {
            constexpr int ri = decltype(rit)::value;
            xl_static_for<MT * 4>([&](auto gt) __attribute__((always_inline)) {
                constexpr int G = decltype(gt)::value, mi = G / 4, rg = G % 4;
                xl_acc_pack4<ri * MT + mi, rg>(pk[ri][G][0], pk[ri][G][1]);
            });
            if (STATS && tile == 0)      // shift = the row's first rounded output of this wave (lane l31 of the lower half)
                sShift[ri] = __shfl(__uint_as_float(pk[ri][0][0] << 16), l31);
        });
        pend_mcol = col0 + (int64_t)tile * TN + wm * 128;
    };
#pragma unroll
    for (int ri = 0; ri < RT; ++ri)
#pragma unroll
        for (int g = 0; g < MT * 4; ++g) pk[ri][g][0] = pk[ri][g][1] = 0u;
    for (int tile = 0; tile < ntile; ++tile) {
        chunk(std::integral_constant<int, 0>{}, std::true_type{}, std::true_type{}, 0);
        for (int sub = 1; sub < cps; ++sub) chunk(std::integral_constant<int, 0>{}, std::false_type{}, std::false_type{}, sub);
        chunk(std::integral_constant<int, 1>{}, std::true_type{}, std::false_type{}, 0);
        for (int sub = 1; sub < cps; ++sub) chunk(std::integral_constant<int, 1>{}, std::false_type{}, std::false_type{}, sub);
        chunk(std::integral_constant<int, 2>{}, std::true_type{}, std::false_type{}, 0);
        for (int sub = 1; sub < cps; ++sub) chunk(std::integral_constant<int, 2>{}, std::false_type{}, std::false_type{}, sub);
        chunk(std::integral_constant<int, 3>{}, std::true_type{}, std::false_type{}, 0);
        for (int sub = 1; sub < cps; ++sub) chunk(std::integral_constant<int, 3>{}, std::false_type{}, std::false_type{}, sub);
        for (int ch = RT * cps; ch < nch; ++ch) chunk(std::integral_constant<int, RT>{}, std::false_type{}, std::false_type{}, 0);
        phase_a(tile);
        pend = true;
    }
    // the last tile's phase B has no MFMAs to hide behind; and nothing of the three surplus chunks may still be in flight
    // when the ring is reused below or the workgroup's LDS is handed on
    xl_opaque(slab_w);
    xl_opaque(slab_r);
    xl_opaque(y_lane);
    xl_static_for<RT>([&](auto rit) __attribute__((always_inline)) {
        xl_static_for<MT * 4>([&](auto gt) __attribute__((always_inline)) { slice_write(rit, gt); });
        xl_static_for<8>([&](auto it) __attribute__((always_inline)) {
            constexpr int ri = decltype(rit)::value, IT = decltype(it)::value;
            slice_fetch(IT, IT);
            slice_store(IT, IT, reinterpret_cast<char *>(Y) + ((int64_t)(y_row0 + ri * 32 + IT * 4) * M + pend_mcol) * 2);
        });
    });
    gm_wait_vm<0>();
    if (STATS) {
        float *s_st = reinterpret_cast<float *>(smem);                  // [2][TR][2] (mean, M2): the ring is idle now
        const float np = 128.0f * (float)ntile;
        __syncthreads();
#pragma unroll
        for (int ri = 0; ri < RT; ++ri) {
            const float s1 = sS[ri].x + sS[ri].y, q1 = sQ[ri].x + sQ[ri].y;
            const float s = s1 + __shfl_xor(s1, 32), q = q1 + __shfl_xor(q1, 32);
            if (half == 0) {
                const int row = wr * 128 + ri * 32 + l31;
                const float dm = s / np;
                s_st[(wm * C::TR + row) * 2 + 0] = sShift[ri] + dm;
                s_st[(wm * C::TR + row) * 2 + 1] = fmaxf(q - s * dm, 0.0f);
            }
        }
        __syncthreads();
        if (wm == 0 && half == 0) {
#pragma unroll
            for (int ri = 0; ri < RT; ++ri) {
                const int row = wr * 128 + ri * 32 + l31;
                float n = 0.0f, mean = 0.0f, m2 = 0.0f;
#pragma unroll
                for (int w = 0; w < 2; ++w)
                    gm_chan(n, mean, m2, np, s_st[(w * C::TR + row) * 2], s_st[(w * C::TR + row) * 2 + 1]);
                {
                    float *pp = part + ((((size_t)grp * Rg + r0 + row) * views + view) * P + rloc) * 3;
                    pp[0] = 0.0f;                                       // (S, Q, shift) with S = 0: mean = shift, M2 = Q
                    pp[1] = m2;
                    pp[2] = mean;
                }
            }
        }
    }
    XL_RELEASE_AGPRS(agpr_hold);
}

}  // namespace grafp
