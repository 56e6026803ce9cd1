// knn_split.hip -- the dynamic k-NN graph (K3-K5) through a SPLIT-bf16 Gram matrix with certified results, gfx950.
//
// Same result as knn_graph.hip (= oracle/csrc/knn_graph.c = /root/reference/encoder/gcn_lib/torch_edge.py:7-18,70-103,
// 270-284), bit for bit, at a fraction of the matrix time.  knn_topk_kernel forms every Gram entry with the exact-f32
// MFMA (64 cycles per 2 channels of a 32 x 32 tile) because the neighbour INDICES must equal the oracle's.  But the
// indices only depend on the ORDER of the distances, and an approximation with a rigorous error bound m decides that
// order wherever two distances differ by more than 2 m:
//   * every normalised feature is split into hi = bf16(v), lo = bf16(v - hi) (knn_normalize_split_kernel; v - hi is
//     exact, |v - hi - lo| <= 2^-18 |v|), and g~ = <xh,yh> + <xh,yl> + <xl,yh> runs on the bf16 matrix cores (three
//     32-cycle MFMAs per 16 channels: 5.3 x fewer matrix cycles); the dropped terms are <= 3.01 * 2^-18 |x||y|;
//   * each lane keeps the k smallest approximate distances of its query as integer KEYS (distance bits with the
//     candidate index in the low mantissa bits: one v_min_u32 / v_max_u32 pair per slot instead of a compare and four
//     selects) plus the smallest key that did not make the list (the (k+1)-th);
//   * a query is CERTIFIED when consecutive entries of its (k+1)-list are more than 2 m (+ the key truncation) apart:
//     then the oracle's f32 distances have the same strict order and no candidate outside the list can enter it.  The
//     few uncertified queries (near-ties, duplicates: 1-3 % on encoder features) are flagged and knn_exact_clip_kernel
//     recomputes them, clip by clip, with the oracle's exact arithmetic (c-ordered fmaf chains, (sq_i + (-2 g)) + sq_j, ties to the
//     lowest index).
// Error budget for unit-norm rows (|x| = |y| = 1 up to rounding; the entry requires normalize = 1), C channels:
//   representation            3.01 * 2^-18                       = 1.15e-5
//   MFMA accumulation         3 C additions, each <= 2^-23 of a partial sum <= 1.004 (truncation assumed)
//   => |g~ - g| <= e_g(C) = 1.15e-5 + 3.6e-7 C ;   d~ = (sq_q + 2^-10) - 2 g~ + sq_j adds two roundings (<= 5e-7)
//   oracle's own rounding     |d_oracle - d| <= 2 C 2^-24 + 5e-7
//   m(C) = 2 e_g(C) + 2^-23 C + 1e-6         (C = 64: 7.7e-5, C = 512: 4.5e-4; typical errors are 10-30 x smaller)
// The constant 2^-10 keeps every d~ positive (m < 2^-10 is checked), so the keys order as unsigned integers.
#include <math.h>

#include "common.h"
#include "dma_ring.h"
#include "tuning.h"

namespace grafp {

constexpr int KS_TQ = 128;                  // query nodes per workgroup (32 per wave)
constexpr int KS_TR = 128;                  // candidate nodes per block
constexpr int KS_KC = 32;                   // channels per chunk
constexpr int KS_PLANE = KS_KC * KS_TR * 2; // one bf16 tile: 32 channel rows x 256 B
constexpr int KS_STAGE = 4 * KS_PLANE;      // candidates hi | candidates lo | queries hi | queries lo
constexpr int KS_LDS = 2 * KS_STAGE + 3 * KS_TR * 4;
constexpr float KS_SHIFT = 9.765625e-4f;    // 2^-10

__device__ __forceinline__ float ks_ld(const float *p) { return *p; }
__device__ __forceinline__ float ks_ld(const unsigned short *p) { return __uint_as_float(((unsigned)*p) << 16); }

// pass 1: channel-L2 normalisation exactly as knn_normalize_kernel (same chains, same bits) -- but the normalised f32
// values are not written: only their two bf16 halves, the squared norms and the denominators (pass 3 re-derives any f32
// value it needs as x / den, the same IEEE division)
template <typename T>
__global__ __launch_bounds__(256) void knn_normalize_split_kernel(const T *__restrict__ x, int64_t sb, int64_t sc,
                                                                  float *__restrict__ den_out, float *__restrict__ sq,
                                                                  unsigned short *__restrict__ xh,
                                                                  unsigned short *__restrict__ xl, int C, int N,
                                                                  int *__restrict__ counters) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    // the counters of this call (all uncertified queries; diagnostics) start at zero: done here, in front of pass 2 in
    // stream order, rather than by a hipMemsetAsync -- a memset NODE inside a captured graph faulted on replay once
    // other work had run in between (HIP 7.0)
    if (blockIdx.x == 0 && b == 0 && threadIdx.x < 2) counters[threadIdx.x] = 0;
    if (n >= N) return;
    const T *xb = x + (size_t)b * sb + n;
    const size_t o = (size_t)b * C * N + n;
    float ss = 0.0f;
    int c = 0;
    for (; c + 8 <= C; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ks_ld(xb + (size_t)(c + u) * sc);
#pragma unroll
        for (int u = 0; u < 8; ++u) ss = __builtin_fmaf(v[u], v[u], ss);
    }
    for (; c < C; ++c) {
        const float v = ks_ld(xb + (size_t)c * sc);
        ss = __builtin_fmaf(v, v, ss);
    }
    const float den = fmaxf(sqrtf(ss), 1e-12f);       // sqrtf: correctly rounded (see knn_graph.hip)
    float q = 0.0f;
    for (c = 0; c + 8 <= C; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ks_ld(xb + (size_t)(c + u) * sc);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v[u] = __fdiv_rn(v[u], den);
            q = __builtin_fmaf(v[u], v[u], q);
            const unsigned h = gm_pack_bf16(v[u], 0.0f) & 0xffffu;
            const float r = v[u] - __uint_as_float(h << 16);          // exact
            xh[o + (size_t)(c + u) * N] = (unsigned short)h;
            xl[o + (size_t)(c + u) * N] = (unsigned short)(gm_pack_bf16(r, 0.0f) & 0xffffu);
        }
    }
    for (; c < C; ++c) {
        const float v = __fdiv_rn(ks_ld(xb + (size_t)c * sc), den);
        q = __builtin_fmaf(v, v, q);
        const unsigned h = gm_pack_bf16(v, 0.0f) & 0xffffu;
        const float r = v - __uint_as_float(h << 16);
        xh[o + (size_t)c * N] = (unsigned short)h;
        xl[o + (size_t)c * N] = (unsigned short)(gm_pack_bf16(r, 0.0f) & 0xffffu);
    }
    sq[(size_t)b * N + n] = q;
    den_out[(size_t)b * N + n] = den;
}

// (distance, index) as one order-preserving 64-bit key (every finite float)
__device__ __forceinline__ unsigned long long ks_key64(float d, int i) {
    unsigned u = __float_as_uint(d);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned)i;
}

template <bool V> struct KsBool { static constexpr bool value = V; };

// K smallest keys, ascending, + the smallest key that left (or never entered) the list
template <int K>
struct KeyList {
    unsigned k[K], next;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int t = 0; t < K; ++t) k[t] = 0x7f7fffffu;       // the largest finite float: an empty slot
        next = 0x7f7fffffu;
    }
    static __device__ __forceinline__ unsigned med3(unsigned a, unsigned b, unsigned c) {      // -> v_med3_u32
        const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
        const unsigned m = hi < c ? hi : c;
        return lo < m ? m : lo;
    }
    // With k ascending and next >= k[K-1], slot t of the new list is the MEDIAN of (its lower neighbour, itself, v): v
    // below both -> the neighbour moves up, v between -> v, v above -> unchanged; the key that leaves is the median of
    // (k[K-1], v, next).  K + 1 instructions per candidate instead of the 2 K + 1 of a min / max bubble (the scans are
    // bound by exactly these VALU instructions), evaluated from the top so that every slot still sees its OLD neighbour.
    __device__ __forceinline__ void push(unsigned v) {
        next = med3(k[K - 1], v, next);
#pragma unroll
        for (int t = K - 1; t > 0; --t) k[t] = med3(k[t - 1], k[t], v);
        k[0] = v < k[0] ? v : k[0];
    }
};

// pass 2.  Workgroup = 128 queries of one clip x all candidates, 4 waves x 32 queries; a lane owns ONE query (MFMA column
// j = lane & 31) and, per candidate block, the 64 candidates of its rows (mfma_row).  Per chunk of 32 channels the four
// bf16 tiles arrive by LDS-DMA (wave w moves plane w: 8 x 1 KiB), one chunk ahead, one barrier per chunk; the fragments
// need 8 consecutive channels of a node from tiles whose rows are channels: ds_read_b64_tr_b16, 64-byte segments of a
// row XOR-swizzled by (channel & 3) on the DMA source side and on the read side (as conv1x1_gemm_kernel).
template <int K, typename I>
__global__ __launch_bounds__(256, 2) void knn_topk_split_kernel(const unsigned short *__restrict__ xh,
                                                                const unsigned short *__restrict__ xl,
                                                                const float *__restrict__ sq, I *__restrict__ idx,
                                                                int *__restrict__ unc_count,
                                                                unsigned char *__restrict__ unc_flag,
                                                                int *__restrict__ extra,
                                                                int C, int N, int tiles_per_clip, int nblocks,
                                                                float margin2, unsigned key_mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *const sSq = reinterpret_cast<float *>(smem + 2 * KS_STAGE);
    const unsigned lds0 = (unsigned)(uintptr_t)(gm_lptr)smem;

    const int bid = xcd_remap(blockIdx.x, nblocks);
    const int b = bid / tiles_per_clip;
    const int q0 = (bid % tiles_per_clip) * KS_TQ;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *sqb = sq + (size_t)b * N;
    const int myq = q0 + wave * 32 + l31;
    const float dq = sqb[myq] + KS_SHIFT;

    const int nch = C / KS_KC, nblk = N / KS_TR, T = nblk * nch;

    // ---- DMA: wave w moves plane w of every chunk (0 candidates hi, 1 candidates lo, 2 queries hi, 3 queries lo) ----
    // instruction i covers channel rows 4i .. 4i+3 (256 B each); LDS slot s' = lane & 15 of row lane >> 4 holds source
    // segment (s' >> 2) ^ (row & 3), piece s' & 3
    const unsigned short *plane = ((wave & 1) ? xl : xh) + (size_t)b * C * N;
    const int rowl = lane >> 4, sl = lane & 15;
    const int scol = (((sl >> 2) ^ (rowl & 3)) * 4 + (sl & 3)) * 8;           // element offset inside the 128-node row
    const unsigned short *src0 = plane + (size_t)rowl * N + scol + (wave >= 2 ? q0 : 0);
    auto dma_chunk = [&](int t) {
        const int blk = t / nch, ch = t - blk * nch;
        const unsigned short *s = src0 + (size_t)(ch * KS_KC) * N + (wave >= 2 ? 0 : blk * KS_TR);
        const unsigned st = lds0 + (t & 1) * KS_STAGE + wave * KS_PLANE;
#pragma unroll
        for (int i = 0; i < 8; ++i) gm_dma16(s + (size_t)(4 * i) * N, st + i * 1024);
        // the block's 128 squared norms: also by DMA (an ordinary load here would make hipcc drain vmcnt(0) -- the DMAs
        // above included -- in front of its LDS write)
        if (ch == 0 && wave < 2)
            gm_dma4(sqb + blk * KS_TR + wave * 64 + lane, lds0 + 2 * KS_STAGE + ((blk % 3) * KS_TR + wave * 64) * 4);
    };

    // ---- fragment offsets inside a plane: node tile tt (32 nodes), k-step ks (16 channels) ----
    // lane i = lane & 15 of group (lane >> 4) & 1: row 16 ks + 8 half + (i >> 2) (+ 4), nodes tt*32 + 16 grp + 4 (i & 3)
    int foff[4];
    {
        const int i = lane & 15, grp = (lane >> 4) & 1;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int bytecol = (tt * 32 + 16 * grp + 4 * (i & 3)) * 2;
            const int seg = (bytecol >> 6) ^ (i >> 2);
            foff[tt] = (8 * half + (i >> 2)) * 256 + seg * 64 + (bytecol & 63);
        }
    }
    int qoff;                                                  // this wave's 32 queries: node tile `wave` of the query planes
    {
        const int i = lane & 15, grp = (lane >> 4) & 1;
        const int bytecol = (wave * 32 + 16 * grp + 4 * (i & 3)) * 2;
        const int seg = (bytecol >> 6) ^ (i >> 2);
        qoff = (8 * half + (i >> 2)) * 256 + seg * 64 + (bytecol & 63);
    }
    auto frag = [&](const unsigned char *pl, int off) -> gm_bf16x8 {
        const gm_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gm_s16x4 __attribute__((address_space(3))) *)(pl + off));
        const gm_s16x4 hi =
            __builtin_amdgcn_ds_read_tr16_b64_v4i16((gm_s16x4 __attribute__((address_space(3))) *)(pl + off + 4 * 256));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    f32x16 acc[4], prev[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.0f; prev[t][r] = 0.0f; }
    KeyList<K + 1> best;                                       // the k + 1 smallest keys, + the next one
    best.init();

    // key of element e (tile e / 16, register e % 16) of a finished block: the index bits of a lane never overlap --
    // (r & 3) | 4 half | 8 (r >> 2) | 32 tile | 128 block
    auto insert = [&](const f32x16 (&a)[4], int e, int sq_base, unsigned lane_bits) {
        const int loc_c = (e >> 4) * 32 + ((e & 15) & 3) + 8 * ((e & 15) >> 2);        // compile-time part of the index
        const float d = __builtin_fmaf(-2.0f, a[e >> 4][e & 15], dq) + sSq[sq_base + loc_c + 4 * half];
        best.push((__float_as_uint(d) & key_mask) | lane_bits | (unsigned)loc_c);
    };

    dma_chunk(0);
    gm_wait_vm<0>();
    __syncthreads();
    for (int blk = 0; blk < nblk; ++blk) {
        const int sq_prev = ((blk + 2) % 3) * KS_TR;
        const unsigned bits_prev = (unsigned)((blk - 1) * KS_TR) | (unsigned)(4 * half);
        for (int ch = 0; ch < nch; ++ch) {
            const int t = blk * nch + ch;
            if (t + 1 < T) dma_chunk(t + 1);
            const unsigned char *st = smem + (t & 1) * KS_STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const gm_bf16x8 qh = frag(st + 2 * KS_PLANE, qoff + ks * 16 * 256);
                const gm_bf16x8 ql = frag(st + 3 * KS_PLANE, qoff + ks * 16 * 256);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const gm_bf16x8 ah = frag(st, foff[tt] + ks * 16 * 256);
                    const gm_bf16x8 al = frag(st + KS_PLANE, foff[tt] + ks * 16 * 256);
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh, acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql, acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh, acc[tt], 0, 0, 0);
                }
                // the 64 keys of the PREVIOUS block, half of them per k-step of this block's first chunk: VALU work
                // beside the matrix work
                if (ch == 0 && blk > 0) {
#pragma unroll
                    for (int e = 0; e < 32; ++e) insert(prev, ks * 32 + e, sq_prev, bits_prev);
                }
            }
            gm_wait_vm<0>();
            __syncthreads();
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            prev[tt] = acc[tt];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tt][r] = 0.0f;
        }
    }
    {
        const unsigned bits_last = (unsigned)((nblk - 1) * KS_TR) | (unsigned)(4 * half);
#pragma unroll
        for (int e = 0; e < 64; ++e) insert(prev, e, ((nblk + 2) % 3) * KS_TR, bits_last);
    }
    // the two half-waves saw disjoint candidate subsets of the same query
    unsigned ok[K + 1], onext = (unsigned)__shfl_xor((int)best.next, 32);
#pragma unroll
    for (int t = 0; t <= K; ++t) ok[t] = (unsigned)__shfl_xor((int)best.k[t], 32);
#pragma unroll
    for (int t = 0; t <= K; ++t) best.push(ok[t]);
    best.next = onext < best.next ? onext : best.next;
    if (half == 0) {
        const unsigned imask = ~key_mask;
        const size_t row = (size_t)b * N + myq;
        I *o = idx + row * K;
        // CERTIFIED: consecutive entries of the (k+1)-list further apart than 2 m (+ the truncation of the lower key):
        // the oracle's distances have the same strict order and nothing outside the first k can enter.
        // LIGHT: not certified, but the (k+2)-th smallest key is beyond 2 m of the k-th: the exact top k are among the
        // k + 1 listed candidates (every unlisted one is strictly worse than the first k listed) -- pass 3a recomputes
        // those k + 1 distances exactly.  HEAVY (>= 3 candidates inside the band of the k-th: clusters, duplicates):
        // pass 3b rescans the whole clip for the query.
        const float trunc = __uint_as_float(0x3f800000u + imask) - 1.0f;       // relative size of the dropped mantissa bits
        bool certified = true;
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const float lo = __uint_as_float(best.k[t] & key_mask), hi = __uint_as_float(best.k[t + 1] & key_mask);
            certified = certified && (hi - lo * (1.0f + trunc) > margin2);
            o[t] = (I)(best.k[t] & imask);
        }
        const bool light = __uint_as_float(best.next & key_mask) -
                               __uint_as_float(best.k[K - 1] & key_mask) * (1.0f + trunc) > margin2;
        unc_flag[row] = certified ? 0 : (light ? 1 : 2);
        if (!certified) {
            atomicAdd(unc_count, 1);                                           // diagnostics: all uncertified queries
            if (light) extra[row] = (int)(best.k[K] & imask);          // the (k+1)-th listed candidate (pass 3a reads it)
        }
    }
}

// ---- bf16 inputs: the RAW form (round 3b) -------------------------------------------------------------------------
// A bf16 feature is its own exact bf16 operand.  G = <x_q, x_j> on the RAW features needs ONE bf16 MFMA per 16 channels
// (no hi / lo planes, nothing written by pass 1 but the norms), and the normalisation moves behind the product:
//   g~ = G r_j / den_q   (r_j = fl(1 / den_j)),      d'' = den_q d~ = fma(G, -2 r_j, fma(sq_j, den_q, (sq_q + 2^-10) den_q)).
// For a fixed query the positive factor den_q does not change the order, so the keys are built from d'' (two VALU
// instructions per candidate, as before) and the gaps are compared against 2 m den_q.  Error budget, unit rows:
//   MFMA accumulation         C additions, each <= 2^-23 of a partial sum <= |x_q||x_j| (the products are exact in f32)
//   r_j, the division         2 roundings, <= 2^-23 together
//   oracle: fl(x / den) twice + its c-ordered fmaf chain            <= 2 * 2^-24 + C 2^-24
//   => |g~ - g_oracle| <= e_r(C) = 1.5 C 2^-23 + 3 * 2^-23 ;   m_r(C) = 2 e_r(C) + 2e-6   (C = 64: 2.5e-5, 3 x tighter
//   than the split form; C = 512: 1.9e-4)
// Ranges: den >= 1e-12 by construction; a clip with a node norm above 2^40 (G could overflow) sends all its queries to
// the exact pass.  Same outputs and tiers as knn_topk_split_kernel.
constexpr int KR_NS = 4;                    // ring stages of [candidates | queries] (16 KB each)
constexpr int KR_STAGE = 2 * KS_PLANE;
constexpr int KR_NT = 8;                    // per-block candidate tables (cj = -2 r_j, sq_j) in flight
constexpr int KR_LDS = KR_NS * KR_STAGE + KR_NT * KS_TR * 8;
constexpr float KR_TINY = 1.8189894e-12f;   // 2^-39: |cj| below it <=> den_j > 2^40

// pass 1 (RAW): den, sq exactly as knn_normalize_kernel (same chains, same bits) + the candidate table; VE nodes per thread
template <int VE>
__global__ __launch_bounds__(256) void knn_norms_kernel(const unsigned short *__restrict__ x, int64_t sb, int64_t sc,
                                                        float *__restrict__ den_out, float *__restrict__ sq,
                                                        float2 *__restrict__ cs, int B, int C, int N,
                                                        int *__restrict__ counters) {
    const int per = N / VE;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid < 2) counters[gid] = 0;                            // see knn_normalize_split_kernel
    if (gid >= (int64_t)B * per) return;
    const int b = (int)(gid / per), n0 = (int)(gid - (int64_t)b * per) * VE;
    const unsigned short *xb = x + (size_t)b * sb + n0;
    float ss[VE], den[VE], q[VE];
#pragma unroll
    for (int u = 0; u < VE; ++u) ss[u] = 0.0f;
    // raw bf16 pairs of channel c (VE / 2 dwords); 8 channels of loads in flight per thread
    auto load = [&](int c, unsigned (&w)[VE / 2]) {
        if (VE == 8) {
            const uint4 t = *reinterpret_cast<const uint4 *>(xb + (size_t)c * sc);
            w[0] = t.x; w[1 % (VE / 2)] = t.y; w[2 % (VE / 2)] = t.z; w[3 % (VE / 2)] = t.w;
        } else if (VE == 4) {
            const uint2 t = *reinterpret_cast<const uint2 *>(xb + (size_t)c * sc);
            w[0] = t.x; w[1 % (VE / 2)] = t.y;
        } else {
            w[0] = *reinterpret_cast<const unsigned *>(xb + (size_t)c * sc);
        }
    };
    for (int c = 0; c < C; c += 8) {                            // C % 32 == 0
        unsigned w[8][VE / 2];
#pragma unroll
        for (int j = 0; j < 8; ++j) load(c + j, w[j]);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < VE / 2; ++e) {
                const float lo = __uint_as_float(w[j][e] << 16), hi = __uint_as_float(w[j][e] & 0xffff0000u);
                ss[2 * e] = __builtin_fmaf(lo, lo, ss[2 * e]);
                ss[2 * e + 1] = __builtin_fmaf(hi, hi, ss[2 * e + 1]);
            }
    }
#pragma unroll
    for (int u = 0; u < VE; ++u) {
        den[u] = fmaxf(sqrtf(ss[u]), 1e-12f);
        q[u] = 0.0f;
    }
    for (int c = 0; c < C; c += 8) {
        unsigned w[8][VE / 2];
#pragma unroll
        for (int j = 0; j < 8; ++j) load(c + j, w[j]);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < VE / 2; ++e) {
                const float lo = __fdiv_rn(__uint_as_float(w[j][e] << 16), den[2 * e]);
                const float hi = __fdiv_rn(__uint_as_float(w[j][e] & 0xffff0000u), den[2 * e + 1]);
                q[2 * e] = __builtin_fmaf(lo, lo, q[2 * e]);
                q[2 * e + 1] = __builtin_fmaf(hi, hi, q[2 * e + 1]);
            }
    }
    const size_t o = (size_t)b * N + n0;
#pragma unroll
    for (int u = 0; u < VE; ++u) {
        sq[o + u] = q[u];
        den_out[o + u] = den[u];
    }
    // the candidate table, PAIR-interleaved: nodes 2p, 2p + 1 -> (cj_2p, cj_2p+1, sq_2p, sq_2p+1), so that the scan reads
    // two candidates' factors as two aligned register pairs (one v_pk_fma_f32 per pair and term; knn_topk_raw_kernel)
    f32x4 *cs4 = reinterpret_cast<f32x4 *>(cs) + (o >> 1);                 // o = b * N + n0 is a multiple of VE (even)
#pragma unroll
    for (int u = 0; u < VE; u += 2)
        cs4[u >> 1] = f32x4{-2.0f * __frcp_rn(den[u]), -2.0f * __frcp_rn(den[u + 1]), q[u], q[u + 1]};
}

template <int K, typename I>
__global__ __launch_bounds__(256, 2) void knn_topk_raw_kernel(const unsigned short *__restrict__ x, int64_t sb, int64_t sc,
                                                              const float *__restrict__ sq, const float *__restrict__ den,
                                                              const float2 *__restrict__ cs, I *__restrict__ idx,
                                                              int *__restrict__ unc_count,
                                                              unsigned char *__restrict__ unc_flag,
                                                              int *__restrict__ extra,
                                                              int C, int N, int tiles_per_clip, int nblocks,
                                                              float margin2, unsigned key_mask) {
    constexpr int D = KR_NS - 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const float2 *const sCS = reinterpret_cast<const float2 *>(smem + KR_NS * KR_STAGE);
    const unsigned lds0 = (unsigned)(uintptr_t)(gm_lptr)smem;

    const int bid = xcd_remap(blockIdx.x, nblocks);
    const int b = bid / tiles_per_clip;
    const int q0 = (bid % tiles_per_clip) * KS_TQ;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int myq = q0 + wave * 32 + l31;
    const float iq = den[(size_t)b * N + myq];
    const float dqi = (sq[(size_t)b * N + myq] + KS_SHIFT) * iq;
    const float2 *csb = cs + (size_t)b * N;

    const int nch = C / KS_KC, nblk = N / KS_TR, T = nblk * nch;

    // ---- DMA: waves 0, 1 move the candidate tile of every chunk (channel rows 0-15 / 16-31), waves 2, 3 the query tile;
    // instruction i covers 4 channel rows (256 B each); LDS slot s' = lane & 15 of row lane >> 4 holds source segment
    // (s' >> 2) ^ (row & 3), piece s' & 3
    const int pl = wave >> 1, hw = wave & 1;
    const int rowl = lane >> 4, sl = lane & 15;
    const int scol = (((sl >> 2) ^ (rowl & 3)) * 4 + (sl & 3)) * 8;
    const unsigned short *src0 = x + (size_t)b * sb + (size_t)(hw * 16 + rowl) * sc + scol + (pl ? q0 : 0);
    auto dma_chunk = [&](int t) {
        const int blk = t / nch, ch = t - blk * nch;
        const unsigned short *s = src0 + (size_t)(ch * KS_KC) * sc + (pl ? 0 : blk * KS_TR);
        const unsigned st = lds0 + (t % KR_NS) * KR_STAGE + pl * KS_PLANE + hw * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) gm_dma16(s + (size_t)(4 * i) * sc, st + i * 1024);
        // the block's candidate table: 128 x (cj, sq_j) = 1 KiB, by DMA as well (see knn_topk_split_kernel)
        if (ch == 0 && wave == 0)
            gm_dma16(csb + blk * KS_TR + lane * 2, lds0 + KR_NS * KR_STAGE + (blk % KR_NT) * (KS_TR * 8));
    };

    int foff[4];
    {
        const int i = lane & 15, grp = (lane >> 4) & 1;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int bytecol = (tt * 32 + 16 * grp + 4 * (i & 3)) * 2;
            const int seg = (bytecol >> 6) ^ (i >> 2);
            foff[tt] = (8 * half + (i >> 2)) * 256 + seg * 64 + (bytecol & 63);
        }
    }
    int qoff;
    {
        const int i = lane & 15, grp = (lane >> 4) & 1;
        const int bytecol = (wave * 32 + 16 * grp + 4 * (i & 3)) * 2;
        const int seg = (bytecol >> 6) ^ (i >> 2);
        qoff = (8 * half + (i >> 2)) * 256 + seg * 64 + (bytecol & 63);
    }
    auto frag = [&](const unsigned char *p, int off) -> gm_bf16x8 {
        const gm_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gm_s16x4 __attribute__((address_space(3))) *)(p + off));
        const gm_s16x4 hi =
            __builtin_amdgcn_ds_read_tr16_b64_v4i16((gm_s16x4 __attribute__((address_space(3))) *)(p + off + 4 * 256));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    f32x16 acc[4];
    KeyList<K + 1> best;
    best.init();
    bool bad = false;

    // four candidates at once (accumulator registers e0 .. e0 + 3 of one MFMA tile = candidates loc_c .. loc_c + 3): their
    // table entries are two 16-byte broadcast reads [cj0 cj1 sq0 sq1][cj2 cj3 sq2 sq3] (pair-interleaved by pass 1), and
    // d'' = fma(G, cj, fma(sq, den_q, (sq_q + 2^-10) den_q)) runs as v_pk_fma_f32 on aligned register pairs -- two FMAs per
    // instruction, the same IEEE result per element as the scalar form (the scan is bound by exactly these VALU instructions)
    const gm_f32x2 iq2 = {iq, iq}, dqi2 = {dqi, dqi};
    unsigned kmv = key_mask;
    asm volatile("" : "+v"(kmv));
    auto insert4 = [&](const f32x16 (&a)[4], int e0, int tb, unsigned lane_bits) {
        const int tt = e0 >> 4, r0 = e0 & 15;
        const int loc_c = tt * 32 + 8 * (r0 >> 2);
        const f32x4 *tp = reinterpret_cast<const f32x4 *>(sCS + tb + loc_c + 4 * half);
        const f32x4 t01 = tp[0], t23 = tp[1];
        const gm_f32x2 d01 = __builtin_elementwise_fma(gm_f32x2{a[tt][r0], a[tt][r0 + 1]}, gm_f32x2{t01[0], t01[1]},
                                                    __builtin_elementwise_fma(gm_f32x2{t01[2], t01[3]}, iq2, dqi2));
        const gm_f32x2 d23 = __builtin_elementwise_fma(gm_f32x2{a[tt][r0 + 2], a[tt][r0 + 3]}, gm_f32x2{t23[0], t23[1]},
                                                    __builtin_elementwise_fma(gm_f32x2{t23[2], t23[3]}, iq2, dqi2));
        // key = distance bits under key_mask, candidate index in the freed mantissa bits: ONE v_and_or_b32 with the mask in
        // a VGPR and the wave-uniform index in an SGPR (mask and index both in SGPRs do not fit one VOP3: v_and + v_or)
        const unsigned c0 = lane_bits | (unsigned)loc_c;
        best.push((__float_as_uint(d01[0]) & kmv) | c0);
        best.push((__float_as_uint(d01[1]) & kmv) | (c0 + 1));
        best.push((__float_as_uint(d23[0]) & kmv) | (c0 + 2));
        best.push((__float_as_uint(d23[1]) & kmv) | (c0 + 3));
    };
    // cj of candidate c of the table block at tb (pair-interleaved layout)
    auto cj_of = [&](int tb, int c) -> float {
        return reinterpret_cast<const float *>(sCS + tb)[4 * (c >> 1) + (c & 1)];
    };

#pragma unroll
    for (int c = 0; c < D; ++c)
        if (c < T) dma_chunk(c);
    // One candidate block: its products accumulate in `acc` -- the first k-step starts from a zero C OPERAND, no register
    // clears -- and when its last chunk is done the 64 finished values of a lane become keys straight out of the
    // accumulators.  (Until round 5 the finished block was first copied to a second register set and its keys were built
    // "in the MFMA shadow" of the next block: VALU and MFMA instructions of one SIMD do not overlap (tools/microbench/
    // mfma_valu_bench), so the shadow bought nothing, and the 128 v_mov per block of "prev = acc; acc = 0" were a fifth of
    // the kernel's VALU instructions at 64 channels.)
    for (int blk = 0; blk < nblk; ++blk) {
        const int tb = (blk % KR_NT) * KS_TR;
        const unsigned bits = (unsigned)(blk * KS_TR);                  // wave-uniform: (bits | loc_c) stays in SGPRs
        auto chunk = [&](int ch, auto first_c) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_c)::value;
            const int t = blk * nch + ch;
            {   // chunk t landed (this wave's pieces; wave 0's table pieces only make its wait stricter), then everybody's
                const int newer = T - 1 - t;
                if (newer >= D - 1) gm_wait_vm<(D - 1) * 4>();
                else if (newer == 1) gm_wait_vm<4>();
                else gm_wait_vm<0>();
                __builtin_amdgcn_s_barrier();
            }
            if (t + D < T) dma_chunk(t + D);
            const unsigned char *st = smem + (t % KR_NS) * KR_STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const gm_bf16x8 qf = frag(st + KS_PLANE, qoff + ks * 16 * 256);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const gm_bf16x8 af = frag(st, foff[tt] + ks * 16 * 256);
                    if (FIRST && ks == 0) {
                        const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, qf, zero, 0, 0, 0);
                    } else {
                        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, qf, acc[tt], 0, 0, 0);
                    }
                }
            }
        };
        chunk(0, KsBool<true>{});
        for (int ch = 1; ch < nch; ++ch) chunk(ch, KsBool<false>{});
        // in batches of 8: hipcc otherwise hoists all 32 table reads (64 registers) over the inserts and spills
#pragma unroll
        for (int e8 = 0; e8 < 64; e8 += 8) {
            insert4(acc, e8, tb, bits);
            insert4(acc, e8 + 4, tb, bits);
            __builtin_amdgcn_sched_barrier(0);
        }
        bad = bad || fabsf(cj_of(tb, lane)) < KR_TINY || fabsf(cj_of(tb, 64 + lane)) < KR_TINY;
    }
    const bool clip_bad = __any(bad ? 1 : 0) != 0;
    // index bit 2 (which half-wave's rows) is the same for every candidate a lane saw: set once, here
#pragma unroll
    for (int t = 0; t <= K; ++t) best.k[t] |= (unsigned)(4 * half);
    best.next |= (unsigned)(4 * half);
    unsigned ok[K + 1], onext = (unsigned)__shfl_xor((int)best.next, 32);
#pragma unroll
    for (int t = 0; t <= K; ++t) ok[t] = (unsigned)__shfl_xor((int)best.k[t], 32);
#pragma unroll
    for (int t = 0; t <= K; ++t) best.push(ok[t]);
    best.next = onext < best.next ? onext : best.next;
    if (half == 0) {
        const unsigned imask = ~key_mask;
        const size_t row = (size_t)b * N + myq;
        I *o = idx + row * K;
        const float trunc = __uint_as_float(0x3f800000u + imask) - 1.0f;
        const float band = margin2 * iq;                                       // the keys are den_q times the distances
        bool certified = !clip_bad;
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const float lo = __uint_as_float(best.k[t] & key_mask), hi = __uint_as_float(best.k[t + 1] & key_mask);
            certified = certified && (hi - lo * (1.0f + trunc) > band);
            o[t] = (I)(best.k[t] & imask);
        }
        const bool light = !clip_bad && __uint_as_float(best.next & key_mask) -
                                                __uint_as_float(best.k[K - 1] & key_mask) * (1.0f + trunc) > band;
        unc_flag[row] = certified ? 0 : (light ? 1 : 2);
        if (!certified) {
            atomicAdd(unc_count, 1);
            if (light) extra[row] = (int)(best.k[K] & imask);          // the (k+1)-th listed candidate (pass 3a reads it)
        }
    }
}

// pass 3a (LIGHT queries: exact distances to the k + 1 listed candidates only) lives at the head of knn_exact_clip_kernel.
// pass 3b: the HEAVY queries with the oracle's exact arithmetic against every candidate.  One workgroup per CLIP (its feature slab is read
// once per group of 16 uncertified queries instead of once per query): the queries' normalised vectors sit in LDS, thread t
// takes candidates t, t + 256, ... (ascending, so the strict-< insert keeps the lower index on ties), re-derives the
// candidate's normalised features as x / den (the division of pass 1), runs the c-ordered fmaf chains of the 16 queries
// side by side, and the block then takes K rounds of a (distance, index) minimum per query.
template <int V> struct KsInt { static constexpr int value = V; };
constexpr int KX_QG = 16;        // uncertified queries per pass over the clip's features
constexpr int KX_NU = 2;         // candidates per thread and pass (256 threads: column ranges of 512 nodes)
constexpr int KX_CHUNK = 16384;  // bytes of one staged feature chunk (double buffered)

// bytes of LDS in front of the (N)-entry query list of knn_exact_clip_kernel: max of [2 chunks | KX_QG x C floats] (heavy
// pass) and [4 waves x (K + 2) x C floats] (light pass); the same expression sizes the launch (ks_launch_exact)
__host__ __device__ constexpr size_t kx_front_bytes(int K, int C) {
    const size_t heavy = (size_t)2 * KX_CHUNK + (size_t)KX_QG * C * 4, light = (size_t)4 * (K + 2) * C * 4;
    return heavy > light ? heavy : light;
}
template <int K, typename I, typename T>
__global__ __launch_bounds__(256) void knn_exact_clip_kernel(const T *__restrict__ x, int64_t sb, int64_t sc,
                                                             const float *__restrict__ den, const float *__restrict__ sq,
                                                             const unsigned char *__restrict__ unc_flag,
                                                             I *__restrict__ idx, int C, int N, int chc,
                                                             const int *__restrict__ counters,
                                                             int *__restrict__ n_uncertified,
                                                             const int *__restrict__ extra, int stop) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_x[];
    if (n_uncertified && blockIdx.x == 0 && threadIdx.x == 0) *n_uncertified = counters[0];   // diagnostics (last pass)
    T *sX = reinterpret_cast<T *>(sm_x);                                       // [2][chc][W] feature chunks (W columns)
    float *sQ = reinterpret_cast<float *>(sm_x + 2 * KX_CHUNK);                 // [KX_QG][C]
    // the query lists sit behind BOTH uses of the staging area in front of them: the chunk ring + query group of the heavy
    // pass, and the light pass's per-wave (K + 2) x C quotient blocks (sL below), whichever is larger (kx_front_bytes)
    int *s_list = reinterpret_cast<int *>(sm_x + kx_front_bytes(K, C));          // [N]
    __shared__ int s_n;
    __shared__ unsigned long long s_red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ int s_nl;
    if (tid == 0) { s_n = 0; s_nl = 0; }
    __syncthreads();
    // HEAVY queries from the front of s_list, LIGHT ones from its back (a query is one or the other)
    for (int q = tid; q < N; q += 256) {
        const unsigned char f = unc_flag[(size_t)b * N + q];
        if (f == 2) s_list[atomicAdd(&s_n, 1)] = q;
        else if (f == 1) s_list[N - 1 - atomicAdd(&s_nl, 1)] = q;
    }
    __syncthreads();
    const int n = s_n, nl = s_nl;
    if (n == 0 && nl == 0) return;
    if (stop == 1) return;                                     // measurement builds only (tools/knn_prof.sh): after the flag scan
    const T *xb = x + (size_t)b * sb;
    const float *denb = den + (size_t)b * N, *sqb = sq + (size_t)b * N;
    // ---- LIGHT queries of this clip (pass 3a): exact distances to the k + 1 listed candidates only, the oracle's c-ordered
    // fmaf chain on x / den, ranked by (distance, index).  One query per WAVE at a time: all 64 lanes first fetch the k + 2
    // feature columns involved (the query's and its candidates': (k + 2) C / 64 independent 2-byte loads per lane, all in
    // flight together), divide by the column's norm and park the quotients in LDS; then lane s runs candidate s's chain out
    // of LDS.  (Until round 5 eight lanes per query walked the channels with their own dependent global loads, 8 channels
    // at a time: C / 8 load latencies in a row per query -- 50 of the kernel's 100 us at 256 channels, and the whole of its
    // 47 us at 256 clip-views.)  The quotients and the chain are the same IEEE operations in the same order.
    {
        float *sL = reinterpret_cast<float *>(sm_x) + (size_t)wave * (K + 2) * C;      // [K + 2][C]: the staging area is idle
        for (int e = wave; e < nl; e += 4) {                                           // wave-uniform
            const int q = s_list[N - 1 - e];
            const size_t row = (size_t)b * N + q;
            int col[K + 2];
            col[0] = q;
#pragma unroll
            for (int t = 0; t < K; ++t) col[1 + t] = (int)idx[row * K + t];
            col[K + 1] = extra[row];
#pragma unroll
            for (int v = 0; v < K + 2; ++v) {
                const float dv = denb[col[v]];
                for (int c = lane; c < C; c += 64) sL[v * C + c] = __fdiv_rn(ks_ld(xb + (size_t)c * sc + col[v]), dv);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            unsigned long long key = ~0ull;
            int j = 0;
            if (lane <= K) {
                j = col[1];
#pragma unroll
                for (int t = 1; t <= K; ++t) j = lane == t ? col[1 + t] : j;
                const f32x4 *vq4 = reinterpret_cast<const f32x4 *>(sL);
                const f32x4 *vj4 = reinterpret_cast<const f32x4 *>(sL + (size_t)(1 + lane) * C);
                float g = 0.0f;
                for (int c4 = 0; c4 < C / 4; ++c4) {                                    // C % 32 == 0
                    const f32x4 a = vj4[c4], bq = vq4[c4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) g = __builtin_fmaf(a[u], bq[u], g);
                }
                const float d = __builtin_fmaf(-2.0f, g, sqb[q]) + sqb[j];            // (sq_i + (-2 g)) + sq_j
                key = ks_key64(d, j);
            }
            int rank = 0;                                          // among lanes 0 .. K (keys are distinct: the indices are)
#pragma unroll
            for (int t = 0; t <= K; ++t) {
                const unsigned long long other = __shfl(key, t);
                rank += other < key ? 1 : 0;
            }
            if (lane <= K && rank < K) idx[row * K + rank] = (I)j;
            __builtin_amdgcn_wave_barrier();                       // the chains are done with sL before the next query's quotients land
        }
    }
    if (n == 0 || stop == 2) return;                           // (stop == 2: measurement builds, after the light queries)
    const int W = N < 256 * KX_NU ? N : 256 * KX_NU;          // columns per range (N % 128 == 0)
    constexpr int VE = 16 / (int)sizeof(T);                    // elements per 16-byte vector
    const int vrow = W / VE, nvec = chc * vrow;                // vectors per row / per chunk
    // a clip's heavy queries in groups of QG, each group one pass over the clip's features; most clips hold one to four
    // of them, and the pass is VALU work of (one division + QG fma) per candidate and channel: the narrow group for those
    auto passes = [&](auto qg) {
    constexpr int QG = decltype(qg)::value;
    for (int g0 = 0; g0 < n; g0 += QG) {
        const int ng = n - g0 < QG ? n - g0 : QG;
        __syncthreads();
        for (int i = tid; i < ng * C; i += 256) {
            const int qi = i / C, c = i - qi * C, q = s_list[g0 + qi];
            sQ[qi * C + c] = __fdiv_rn(ks_ld(xb + (size_t)c * sc + q), denb[q]);
        }
        float sqq[QG], bd[QG][K];
        int bi[QG][K];
#pragma unroll
        for (int qi = 0; qi < QG; ++qi) {
            sqq[qi] = qi < ng ? sqb[s_list[g0 + qi]] : 0.0f;
#pragma unroll
            for (int t = 0; t < K; ++t) { bd[qi][t] = INFINITY; bi[qi][t] = 0x7fffffff; }
        }
        for (int j0 = 0; j0 < N; j0 += W) {                    // candidate column ranges, ascending
            float g[KX_NU][QG], dj[KX_NU];
#pragma unroll
            for (int u = 0; u < KX_NU; ++u) {
                const int j = j0 + tid + 256 * u;
                dj[u] = (tid + 256 * u < W) ? denb[j] : 1.0f;
#pragma unroll
                for (int qi = 0; qi < QG; ++qi) g[u][qi] = 0.0f;
            }
            // chunks of chc channels x W columns through LDS by LDS-DMA (1 KiB per wave instruction, lane-linear: vector
            // v of the chunk lands at byte 16 v), the next chunk in flight while this one is consumed
            const int nchunk = C / chc;
            auto dma = [&](int ck) {
                const unsigned dst = (unsigned)(uintptr_t)(gm_lptr)sm_x + (ck & 1) * KX_CHUNK;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int v0 = (r * 4 + wave) * 64;        // wave-uniform; nvec is a multiple of 64
                    if (v0 < nvec) {
                        const int v = v0 + lane, row = v / vrow, cv = v - row * vrow;
                        gm_dma16(xb + (size_t)(ck * chc + row) * sc + j0 + cv * VE, dst + v0 * 16);
                    }
                }
            };
            dma(0);
            gm_wait_vm<0>();
            __syncthreads();                                   // chunk 0 landed; sQ is complete
            for (int ck = 0; ck < nchunk; ++ck) {
                const T *buf = sX + (size_t)(ck & 1) * (KX_CHUNK / sizeof(T));
                if (ck + 1 < nchunk) dma(ck + 1);
                // four channels per step: their LDS reads and IEEE divisions are independent of each other (only the fmaf
                // chains are ordered), so issuing them together hides three of every four read + division latencies -- with
                // one channel per step the pass was a chain of C such latencies (36 us per clip, the kernel's critical path)
                auto channels = [&](int cc0, auto nc_c) __attribute__((always_inline)) {
                    constexpr int NC = decltype(nc_c)::value;
                    float v[NC][KX_NU];
#pragma unroll
                    for (int k2 = 0; k2 < NC; ++k2)
#pragma unroll
                        for (int u = 0; u < KX_NU; ++u)
                            v[k2][u] = (tid + 256 * u < W)
                                           ? __fdiv_rn(ks_ld(buf + (size_t)(cc0 + k2) * W + tid + 256 * u), dj[u]) : 0.0f;
#pragma unroll
                    for (int k2 = 0; k2 < NC; ++k2) {
                        const int c = ck * chc + cc0 + k2;
                        float a[QG];
#pragma unroll
                        for (int qi = 0; qi < QG; ++qi) a[qi] = sQ[qi * C + c];
#pragma unroll
                        for (int u = 0; u < KX_NU; ++u)
                            if (tid + 256 * u < W)
#pragma unroll
                                for (int qi = 0; qi < QG; ++qi) g[u][qi] = __builtin_fmaf(v[k2][u], a[qi], g[u][qi]);
                    }
                };
                int cc = 0;
                for (; cc + 4 <= chc; cc += 4) channels(cc, KsInt<4>{});
                for (; cc < chc; ++cc) channels(cc, KsInt<1>{});
                gm_wait_vm<0>();
                __syncthreads();                               // next chunk landed; this buffer may be overwritten
            }
#pragma unroll
            for (int u = 0; u < KX_NU; ++u) {
                const int j = j0 + tid + 256 * u;
                if (tid + 256 * u < W) {
                    const float sqj = sqb[j];
#pragma unroll
                    for (int qi = 0; qi < QG; ++qi) {
                        if (qi < ng) {
                            float v = __builtin_fmaf(-2.0f, g[u][qi], sqq[qi]) + sqj;      // (sq_i + (-2 g)) + sq_j
                            int vi = j;
                            const float v0 = v;
#pragma unroll
                            for (int t = 0; t < K; ++t) {                              // TopK::push_ascending
                                const bool take = v0 < bd[qi][t];
                                const float od = bd[qi][t];
                                const int oi = bi[qi][t];
                                bd[qi][t] = take ? v : od;
                                bi[qi][t] = take ? vi : oi;
                                v = take ? od : v;
                                vi = take ? oi : vi;
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int qi = 0; qi < QG; ++qi) {
            if (qi < ng) {                                                     // block-uniform
                I *o = idx + ((size_t)b * N + s_list[g0 + qi]) * K;
#pragma unroll
                for (int t = 0; t < K; ++t) {
                    const unsigned long long mine = bi[qi][0] == 0x7fffffff ? ~0ull : ks_key64(bd[qi][0], bi[qi][0]);
                    unsigned long long m = mine;
#pragma unroll
                    for (int s2 = 1; s2 < 64; s2 <<= 1) {
                        const unsigned long long other = __shfl_xor(m, s2);
                        m = other < m ? other : m;
                    }
                    if (lane == 0) s_red[wave] = m;
                    __syncthreads();
                    m = s_red[0];
#pragma unroll
                    for (int w = 1; w < 4; ++w) m = s_red[w] < m ? s_red[w] : m;
                    __syncthreads();
                    if (tid == 0) o[t] = (I)(unsigned)(m & 0xffffffffull);
                    if (mine == m) {                                           // the winner pops its head
#pragma unroll
                        for (int u = 0; u + 1 < K; ++u) { bd[qi][u] = bd[qi][u + 1]; bi[qi][u] = bi[qi][u + 1]; }
                        bd[qi][K - 1] = INFINITY;
                        bi[qi][K - 1] = 0x7fffffff;
                    }
                }
            }
        }
    }
    };
    if (n <= 4) passes(KsInt<4>{});
    else passes(KsInt<KX_QG>{});
}

static int ks_index_bits(int N) {
    int b = 1;
    while ((1 << b) < N) ++b;
    return b;
}
static float ks_margin(int C) {
    const float e_g = 1.15e-5f + 3.6e-7f * (float)C;
    return 2.0f * e_g + 1.1920929e-7f * (float)C + 1e-6f;
}
static float ks_margin_raw(int C) { return 2.0f * (1.7881393e-7f * (float)C + 3.6e-7f) + 2e-6f; }
static bool ks_supported(int C, int N, int k) {
    return C > 0 && C % KS_KC == 0 && N >= KS_TR && N % KS_TR == 0 && N <= 4096 && k >= 1 && k <= 4 && k <= N &&
           ks_margin(C) < 0.9f * KS_SHIFT && kx_front_bytes(k, C) + (size_t)N * 4 <= 160 * 1024;   // (exact pass's LDS)
}
static size_t ks_align(size_t v) { return (v + 255) & ~(size_t)255; }

struct KsArgs {
    const void *x;
    bool f32, raw;
    const float2 *cs;
    int64_t sb, sc;
    const unsigned short *xh, *xl;
    const float *sq, *den;
    int *count, *extra, *n_unc;
    unsigned char *flag;
    void *idx;
    int B, C, N;
    float margin2;
    unsigned key_mask;
};
template <int K, typename I, typename T> static void ks_launch_exact(const KsArgs &a, hipStream_t s) {
    // pass 3b stages chunks of `chc` channels x min(N, 512) columns (<= 16 KB; a power of two, C % 32 == 0)
    const int wcols = a.N < 256 * KX_NU ? a.N : 256 * KX_NU;
    int chc = 32;
    while (chc > 1 && (size_t)chc * wcols * sizeof(T) > KX_CHUNK) chc >>= 1;
    const size_t lds_x = kx_front_bytes(K, a.C) + (size_t)a.N * 4;
    (void)hipFuncSetAttribute((const void *)knn_exact_clip_kernel<K, I, T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_x);
    hipLaunchKernelGGL((knn_exact_clip_kernel<K, I, T>), dim3(a.B), dim3(256), lds_x, s, (const T *)a.x, a.sb, a.sc, a.den,
                       a.sq, a.flag, (I *)a.idx, a.C, a.N, chc, a.count, a.n_unc, a.extra, GRAFP_TUNE_INT("GRAFP_KX_STOP", 0));
}
template <int K, typename I> static void ks_launch(const KsArgs &a, hipStream_t s) {
    const int tiles = a.N / KS_TQ, nblocks = a.B * tiles;
    if (a.raw) {
        (void)hipFuncSetAttribute((const void *)knn_topk_raw_kernel<K, I>, hipFuncAttributeMaxDynamicSharedMemorySize, KR_LDS);
        hipLaunchKernelGGL((knn_topk_raw_kernel<K, I>), dim3(nblocks), dim3(256), KR_LDS, s, (const unsigned short *)a.x,
                           a.sb, a.sc, a.sq, a.den, a.cs, (I *)a.idx, a.count, a.flag, a.extra, a.C, a.N, tiles,
                           nblocks, a.margin2, a.key_mask);
    } else {
        (void)hipFuncSetAttribute((const void *)knn_topk_split_kernel<K, I>, hipFuncAttributeMaxDynamicSharedMemorySize, KS_LDS);
        hipLaunchKernelGGL((knn_topk_split_kernel<K, I>), dim3(nblocks), dim3(256), KS_LDS, s, a.xh, a.xl, a.sq, (I *)a.idx,
                           a.count, a.flag, a.extra, a.C, a.N, tiles, nblocks, a.margin2, a.key_mask);
    }
    if (a.f32) ks_launch_exact<K, I, float>(a, s);
    else ks_launch_exact<K, I, unsigned short>(a, s);
}
template <typename I> static void ks_launch_k(int k, const KsArgs &a, hipStream_t s) {
    switch (k) {
    case 1: ks_launch<1, I>(a, s); break;
    case 2: ks_launch<2, I>(a, s); break;
    case 3: ks_launch<3, I>(a, s); break;
    default: ks_launch<4, I>(a, s); break;
    }
}

}  // namespace grafp

extern "C" int grafp_knn_split_supported(int C, int N, int k) { return grafp::ks_supported(C, N, k) ? 1 : 0; }

// Where the split path measured faster than the exact-f32 MFMA kernel (tools/knn_bench.py, random unit features at 2048
// clip-views and the encoder's own features at 256): C = 64: 1.69 vs 3.62 ms, C = 128: 0.98 vs 1.70 ms; C = 256 ties
// (0.87 vs 0.90) and C = 512 loses (0.97 vs 0.60 ms: the error bound grows with C, 3-8 % of the queries take the exact
// passes, which cost more than the 128-candidate scan they replace).
extern "C" int grafp_knn_split_preferred(int C, int N, int k) { return grafp::ks_supported(C, N, k) && C <= 128 ? 1 : 0; }

// ... for inputs of `dtype`: bf16 features take the RAW form (one MFMA per 16 channels, no planes; see knn_topk_raw_kernel),
// measured at 2048 clip-views against the exact-f32 kernel incl. both normalisation passes (profiles/r03_knn_bench.txt)
extern "C" int grafp_knn_split_preferred_for(int dtype, int C, int N, int k) {
    if (dtype != GRAFP_BF16) return grafp_knn_split_preferred(C, N, k);
    return grafp::ks_supported(C, N, k) && C <= GRAFP_TUNE_INT("GRAFP_KNN_RAW_MAXC", 256) ? 1 : 0;
}

static size_t ks_workspace(int dtype, int B, int C, int N) {
    using namespace grafp;
    if (B <= 0 || C <= 0 || N <= 0) return 0;
    const size_t e = (size_t)B * C * N;
    // sq, den | planes (f32 inputs) or the candidate table (bf16 inputs) | counters | flags | extra
    const size_t mid = dtype == GRAFP_BF16 ? ks_align((size_t)B * N * 8) : 2 * ks_align(e * 2);
    return 2 * ks_align((size_t)B * N * 4) + mid + 256 + ks_align((size_t)B * N) + ks_align((size_t)B * N * 4);
}
extern "C" size_t grafp_knn_split_workspace(int B, int C, int N) { return ks_workspace(GRAFP_F32, B, C, N); }
extern "C" size_t grafp_knn_split_workspace_for(int dtype, int B, int C, int N) { return ks_workspace(dtype, B, C, N); }

extern "C" int grafp_knn_graph_split(const void *x, int dtype, int64_t stride_b, int64_t stride_c, int B, int C, int N,
                                     int k, void *idx, int idx_is_i32, void *ws, size_t ws_bytes, int32_t *n_uncertified,
                                     grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(x && idx, "knn_graph_split: null pointer");
    GRAFP_REQUIRE(B > 0 && ks_supported(C, N, k),
                  "knn_graph_split: unsupported shape B=%d C=%d N=%d k=%d (C %% 32, N %% 128, N <= 4096, k <= 4)", B, C, N, k);
    GRAFP_REQUIRE(dtype == GRAFP_F32 || dtype == GRAFP_BF16, "knn_graph_split: dtype %d not in {f32, bf16}", dtype);
    GRAFP_REQUIRE((int64_t)B * N < (1ll << 31), "knn_graph_split: too many nodes");
    const size_t need = ks_workspace(dtype, B, C, N);
    if (!ws || ws_bytes < need) {
        set_error("knn_graph_split: workspace %zu bytes < required %zu", ws_bytes, need);
        return GRAFP_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const bool raw = dtype == GRAFP_BF16;
    const size_t e = (size_t)B * C * N;
    char *p = (char *)ws;
    float *sq = (float *)p;                       p += ks_align((size_t)B * N * 4);
    float *den = (float *)p;                      p += ks_align((size_t)B * N * 4);
    unsigned short *xh = nullptr, *xl = nullptr;
    float2 *cs = nullptr;
    if (raw) {
        cs = (float2 *)p;                         p += ks_align((size_t)B * N * 8);
    } else {
        xh = (unsigned short *)p;                 p += ks_align(e * 2);
        xl = (unsigned short *)p;                 p += ks_align(e * 2);
    }
    int *count = (int *)p;                        p += 256;
    unsigned char *flag = (unsigned char *)p;     p += ks_align((size_t)B * N);
    int *extra = (int *)p;                        p += ks_align((size_t)B * N * 4);
    GRAFP_REQUIRE((((uintptr_t)ws | (uintptr_t)x) & 15) == 0 && (stride_b * (dtype == GRAFP_F32 ? 4 : 2)) % 16 == 0 &&
                      (stride_c * (dtype == GRAFP_F32 ? 4 : 2)) % 16 == 0,
                  "knn_graph_split: input rows and workspace must be 16-byte aligned");
    if (raw) {
        // nodes per thread: 16-byte loads while that leaves >= 2^18 threads, narrower ones for the short rows of the
        // late stages (a thread walks its channels in order -- the oracle's chains -- so nodes are the only parallelism)
        const int64_t nodes = (int64_t)B * N;
        const int ve = nodes >= (1 << 21) ? 8 : nodes >= (1 << 20) ? 4 : 2;
        const int64_t threads = nodes / ve;
        const dim3 gn((unsigned)((threads + 255) / 256));
        if (ve == 8)
            hipLaunchKernelGGL(knn_norms_kernel<8>, gn, dim3(256), 0, s, (const unsigned short *)x, stride_b, stride_c, den,
                               sq, cs, B, C, N, count);
        else if (ve == 4)
            hipLaunchKernelGGL(knn_norms_kernel<4>, gn, dim3(256), 0, s, (const unsigned short *)x, stride_b, stride_c, den,
                               sq, cs, B, C, N, count);
        else
            hipLaunchKernelGGL(knn_norms_kernel<2>, gn, dim3(256), 0, s, (const unsigned short *)x, stride_b, stride_c, den,
                               sq, cs, B, C, N, count);
        GRAFP_CHECK_LAUNCH("knn_norms_kernel");
    } else {
        const dim3 gn((N + 255) / 256, B);
        hipLaunchKernelGGL(knn_normalize_split_kernel<float>, gn, dim3(256), 0, s, (const float *)x, stride_b, stride_c,
                           den, sq, xh, xl, C, N, count);
        GRAFP_CHECK_LAUNCH("knn_normalize_split_kernel");
    }
    KsArgs a;
    a.x = x; a.f32 = dtype == GRAFP_F32; a.raw = raw; a.cs = cs; a.sb = stride_b; a.sc = stride_c; a.xh = xh; a.xl = xl; a.sq = sq; a.den = den;
    a.count = count; a.n_unc = (int *)n_uncertified; a.flag = flag; a.extra = extra; a.idx = idx; a.B = B; a.C = C; a.N = N;
    a.margin2 = 2.0f * (raw ? ks_margin_raw(C) : ks_margin(C));
    a.key_mask = ~((1u << ks_index_bits(N)) - 1u);
    if (idx_is_i32) ks_launch_k<int32_t>(k, a, s);
    else ks_launch_k<int64_t>(k, a, s);
    GRAFP_CHECK_LAUNCH("knn_topk_split_kernel");
    return GRAFP_OK;
}
