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
//     few uncertified queries (near-ties, duplicates: ~1 % on encoder features) go to a list and knn_exact_rows_kernel
//     recomputes them with the oracle's exact arithmetic (c-ordered fmaf chains, (sq_i + (-2 g)) + sq_j, ties to the
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

// pass 1: channel-L2 normalisation exactly as knn_normalize_kernel (same chains, same bits in xn / sq) + the split planes
template <typename T>
__global__ __launch_bounds__(256) void knn_normalize_split_kernel(const T *__restrict__ x, int64_t sb, int64_t sc,
                                                                  float *__restrict__ xn, float *__restrict__ sq,
                                                                  unsigned short *__restrict__ xh,
                                                                  unsigned short *__restrict__ xl, int C, int N) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
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
            xn[o + (size_t)(c + u) * N] = v[u];
            q = __builtin_fmaf(v[u], v[u], q);
            const unsigned h = gm_pack_bf16(v[u], 0.0f) & 0xffffu;
            const float r = v[u] - __uint_as_float(h << 16);          // exact
            xh[o + (size_t)(c + u) * N] = (unsigned short)h;
            xl[o + (size_t)(c + u) * N] = (unsigned short)(gm_pack_bf16(r, 0.0f) & 0xffffu);
        }
    }
    for (; c < C; ++c) {
        const float v = __fdiv_rn(ks_ld(xb + (size_t)c * sc), den);
        xn[o + (size_t)c * N] = v;
        q = __builtin_fmaf(v, v, q);
        const unsigned h = gm_pack_bf16(v, 0.0f) & 0xffffu;
        const float r = v - __uint_as_float(h << 16);
        xh[o + (size_t)c * N] = (unsigned short)h;
        xl[o + (size_t)c * N] = (unsigned short)(gm_pack_bf16(r, 0.0f) & 0xffffu);
    }
    sq[(size_t)b * N + n] = q;
}

// K smallest keys, ascending, + the smallest key that left (or never entered) the list
template <int K>
struct KeyList {
    unsigned k[K], next;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int t = 0; t < K; ++t) k[t] = 0x7f7fffffu;       // the largest finite float: an empty slot
        next = 0x7f7fffffu;
    }
    __device__ __forceinline__ void push(unsigned v) {
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const unsigned lo = v < k[t] ? v : k[t];           // v_min_u32
            v = v < k[t] ? k[t] : v;                           // v_max_u32
            k[t] = lo;
        }
        next = v < next ? v : next;
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
                                                                int *__restrict__ unc_count, int *__restrict__ unc_list,
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
    KeyList<K> best;
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
    unsigned ok[K], onext = (unsigned)__shfl_xor((int)best.next, 32);
#pragma unroll
    for (int t = 0; t < K; ++t) ok[t] = (unsigned)__shfl_xor((int)best.k[t], 32);
#pragma unroll
    for (int t = 0; t < K; ++t) best.push(ok[t]);
    best.next = onext < best.next ? onext : best.next;
    if (half == 0) {
        const unsigned imask = ~key_mask;
        I *o = idx + ((size_t)b * N + myq) * K;
        bool certified = true;
        // consecutive entries (the (k+1)-th included) further apart than 2 m + the truncation of the lower one's key
        const float trunc = __uint_as_float(0x3f800000u + imask) - 1.0f;       // relative size of the dropped mantissa bits
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const float lo = __uint_as_float(best.k[t] & key_mask);
            const float hi = __uint_as_float((t + 1 < K ? best.k[t + 1] : best.next) & key_mask);
            certified = certified && (hi - lo * (1.0f + trunc) > margin2);
            o[t] = (I)(best.k[t] & imask);
        }
        if (!certified) unc_list[atomicAdd(unc_count, 1)] = b * N + myq;
    }
}

// pass 3: the uncertified queries with the oracle's exact arithmetic.  One wave per query; lane l takes candidates l, l + 64,
// ... (ascending, so the strict-< insert keeps the lower index on ties), c-ordered fmaf chains, then K rounds of a
// wave-wide (distance, index) minimum.
__device__ __forceinline__ unsigned long long ks_key64(float d, int i) {
    unsigned u = __float_as_uint(d);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                           // order-preserving for every finite float
    return ((unsigned long long)u << 32) | (unsigned)i;
}

template <int K, typename I>
__global__ __launch_bounds__(256) void knn_exact_rows_kernel(const float *__restrict__ xn, const float *__restrict__ sq,
                                                             I *__restrict__ idx, const int *__restrict__ unc_count,
                                                             const int *__restrict__ unc_list, int C, int N) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int count = *unc_count;
    for (int e = gw; e < count; e += nw) {
        const int row = unc_list[e];
        const int b = row / N, q = row - b * N;
        const float *xb = xn + (size_t)b * C * N;
        const float *sqb = sq + (size_t)b * N;
        const float sq_q = sqb[q];
        float bd[K];
        int bi[K];
#pragma unroll
        for (int t = 0; t < K; ++t) { bd[t] = INFINITY; bi[t] = 0x7fffffff; }
        for (int j0 = 0; j0 < N; j0 += 256) {
            float g[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int c = 0; c < C; ++c) {
                const float a = xb[(size_t)c * N + q];
                const float *r = xb + (size_t)c * N + j0 + lane;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + lane + 64 * u;
                    const float v = j < N ? r[64 * u] : 0.0f;
                    g[u] = __builtin_fmaf(v, a, g[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + lane + 64 * u;
                if (j < N) {
                    float v = __builtin_fmaf(-2.0f, g[u], sq_q) + sqb[j];      // (sq_i + (-2 g)) + sq_j
                    int vi = j;
                    const float v0 = v;
#pragma unroll
                    for (int t = 0; t < K; ++t) {                              // TopK::push_ascending
                        const bool take = v0 < bd[t];
                        const float od = bd[t];
                        const int oi = bi[t];
                        bd[t] = take ? v : od;
                        bi[t] = take ? vi : oi;
                        v = take ? od : v;
                        vi = take ? oi : vi;
                    }
                }
            }
        }
        I *o = idx + (size_t)row * K;
#pragma unroll
        for (int t = 0; t < K; ++t) {
            unsigned long long mine = bi[0] == 0x7fffffff ? ~0ull : ks_key64(bd[0], bi[0]), m = mine;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const unsigned long long other = __shfl_xor(m, s);
                m = other < m ? other : m;
            }
            if (lane == 0) o[t] = (I)(unsigned)(m & 0xffffffffull);
            if (mine == m) {                                                   // the winner pops its head
#pragma unroll
                for (int u = 0; u + 1 < K; ++u) { bd[u] = bd[u + 1]; bi[u] = bi[u + 1]; }
                bd[K - 1] = INFINITY;
                bi[K - 1] = 0x7fffffff;
            }
        }
    }
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
static bool ks_supported(int C, int N, int k) {
    return C > 0 && C % KS_KC == 0 && N >= KS_TR && N % KS_TR == 0 && N <= 4096 && k >= 1 && k <= 4 && k <= N &&
           ks_margin(C) < 0.9f * KS_SHIFT;
}
static size_t ks_align(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace grafp

extern "C" int grafp_knn_split_supported(int C, int N, int k) { return grafp::ks_supported(C, N, k) ? 1 : 0; }

extern "C" size_t grafp_knn_split_workspace(int B, int C, int N) {
    using namespace grafp;
    if (B <= 0 || C <= 0 || N <= 0) return 0;
    const size_t e = (size_t)B * C * N;
    return ks_align(e * 4) + ks_align((size_t)B * N * 4) + 2 * ks_align(e * 2) + 256 + ks_align((size_t)B * N * 4);
}

extern "C" int grafp_knn_graph_split(const void *x, int dtype, int64_t stride_b, int64_t stride_c, int B, int C, int N,
                                     int k, void *idx, int idx_is_i32, void *ws, size_t ws_bytes, int32_t *n_uncertified,
                                     grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(x && idx, "knn_graph_split: null pointer");
    GRAFP_REQUIRE(B > 0 && ks_supported(C, N, k),
                  "knn_graph_split: unsupported shape B=%d C=%d N=%d k=%d (C %% 32, N %% 128, N <= 4096, k <= 4)", B, C, N, k);
    GRAFP_REQUIRE(dtype == GRAFP_F32 || dtype == GRAFP_BF16, "knn_graph_split: dtype %d not in {f32, bf16}", dtype);
    GRAFP_REQUIRE((int64_t)B * N < (1ll << 31), "knn_graph_split: too many nodes");
    const size_t need = grafp_knn_split_workspace(B, C, N);
    if (!ws || ws_bytes < need) {
        set_error("knn_graph_split: workspace %zu bytes < required %zu", ws_bytes, need);
        return GRAFP_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t e = (size_t)B * C * N;
    char *p = (char *)ws;
    float *xn = (float *)p;                       p += ks_align(e * 4);
    float *sq = (float *)p;                       p += ks_align((size_t)B * N * 4);
    unsigned short *xh = (unsigned short *)p;     p += ks_align(e * 2);
    unsigned short *xl = (unsigned short *)p;     p += ks_align(e * 2);
    int *count = (int *)p;                        p += 256;
    int *list = (int *)p;
    GRAFP_REQUIRE((((uintptr_t)xh | (uintptr_t)xl) & 15) == 0, "knn_graph_split: workspace must be 16-byte aligned");
    if (hipMemsetAsync(count, 0, sizeof(int), s) != hipSuccess) {
        set_error("knn_graph_split: hipMemsetAsync failed");
        return GRAFP_ERR_LAUNCH;
    }
    const dim3 gn((N + 255) / 256, B);
    if (dtype == GRAFP_F32)
        hipLaunchKernelGGL(knn_normalize_split_kernel<float>, gn, dim3(256), 0, s, (const float *)x, stride_b, stride_c,
                           xn, sq, xh, xl, C, N);
    else
        hipLaunchKernelGGL(knn_normalize_split_kernel<unsigned short>, gn, dim3(256), 0, s, (const unsigned short *)x,
                           stride_b, stride_c, xn, sq, xh, xl, C, N);
    GRAFP_CHECK_LAUNCH("knn_normalize_split_kernel");
    const int tiles = N / KS_TQ, nblocks = B * tiles;
    const unsigned key_mask = ~((1u << ks_index_bits(N)) - 1u);
    const float margin2 = 2.0f * ks_margin(C);
#define KS_LAUNCH(K, I)                                                                                                 \
    do {                                                                                                                \
        (void)hipFuncSetAttribute((const void *)knn_topk_split_kernel<K, I>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  KS_LDS);                                                                              \
        hipLaunchKernelGGL((knn_topk_split_kernel<K, I>), dim3(nblocks), dim3(256), KS_LDS, s, xh, xl, sq, (I *)idx,     \
                           count, list, C, N, tiles, nblocks, margin2, key_mask);                                       \
        hipLaunchKernelGGL((knn_exact_rows_kernel<K, I>), dim3(512), dim3(256), 0, s, xn, sq, (I *)idx, count, list, C,  \
                           N);                                                                                          \
    } while (0)
#define KS_LAUNCH_K(I)                                  \
    switch (k) {                                        \
    case 1: KS_LAUNCH(1, I); break;                     \
    case 2: KS_LAUNCH(2, I); break;                     \
    case 3: KS_LAUNCH(3, I); break;                     \
    default: KS_LAUNCH(4, I); break;                    \
    }
    if (idx_is_i32) { KS_LAUNCH_K(int32_t) } else { KS_LAUNCH_K(int64_t) }
#undef KS_LAUNCH_K
#undef KS_LAUNCH
    GRAFP_CHECK_LAUNCH("knn_topk_split_kernel");
    if (n_uncertified &&
        hipMemcpyAsync(n_uncertified, count, sizeof(int), hipMemcpyDeviceToDevice, s) != hipSuccess) {
        set_error("knn_graph_split: hipMemcpyAsync failed");
        return GRAFP_ERR_LAUNCH;
    }
    return GRAFP_OK;
}
