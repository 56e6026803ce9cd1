// knn_pre.hip -- dynamic k-NN graph through a bf16 pre-filter with exact f32 rescoring (K3-K5), gfx950.
//
// Same contract and the SAME indices as knn_graph.hip (the reference's DenseDilatedKnnGraph,
// /root/reference/encoder/gcn_lib/torch_edge.py:7-18,70-103,270-284; arithmetic order of oracle/csrc/knn_graph.c),
// for the shapes the encoder produces (C % 64 == 0, N % 128 == 0, k <= 4).  knn_graph.hip evaluates all N^2 distances
// with the exact-f32 MFMA (157 TFLOP/s) and pays ~15 VALU instructions per pair for the sorted insert, which do not
// overlap with the f32 MFMA.  Here:
//   pass 1  Gram tiles on the bf16 matrix cores (16x the f32 rate); per lane only the running MAXIMUM of <q^,x^>
//           over 4 disjoint candidate groups (one v_max3 per two pairs).  The 8 group maxima of a query (2 half-
//           lanes x 4) give 8 upper bounds on true distances of 8 different candidates; the k-th smallest of them
//           bounds the k-th best distance from above.
//   pass 2  the same tiles again; a candidate is kept iff its LOWER bound does not exceed that bound -- one
//           compare per pair against a per-lane constant -- and its index goes to a lane-private list in LDS
//           (4-8 survivors per query on the encoder's features).
//   rescore the exact distance (c-ordered fmaf chain over the f32 rows, (sq_i + (-2 g)) + sq_j) of the survivors and
//           the top-k by (distance, index): bit-identical to the all-f32 kernel.  A lane whose list overflows falls
//           back to exact distances for ALL its candidates.
// Bounds: with x^ = round_bf16(x), |<q^,x^> - <q,x>| <= (2u + u^2)|q||x| <= 0.0039139 (sq_q + sq_j), u = 2^-8, so
//   d in [ (sq_q + sq_j)(1 - S) - 2<q^,x^>, (sq_q + sq_j)(1 + S) - 2<q^,x^> ],  S = 0.008 (spare 1.7e-4 for the MFMA's
//   f32 accumulation and the rounding of the tests); sq_j is replaced by the clip's max / min squared norm (both 1
//   to rounding for normalised features), which makes the tests per-lane constants.
// Data: knn_normalize_rows_kernel writes node-major rows -- xnf (B,N,C) f32 for the rescoring, xnh (B,N,C) bf16 for the
// MFMA fragments (8 consecutive channels = 16 contiguous bytes) -- and sq (B,N).
#include <math.h>

#include "common.h"

namespace grafp {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int KP_TQ = 128;        // query nodes per workgroup (32 per wave)
constexpr int KP_TR = 128;        // candidate nodes per block
constexpr int KP_KC = 64;         // channels per LDS chunk: 128 bytes per node row
constexpr int KP_CAP = 24;        // survivors a lane can hold (per half-lane: half of the candidates)
constexpr float KP_SLACK = 0.008f;
constexpr int KP_TILE_BYTES = KP_TR * KP_KC * 2;                        // 16 KB
constexpr int KP_LDS_BYTES = 4 * KP_TILE_BYTES + 256 * KP_CAP * 2 + 64;  // A[2] | B[2] | lists | reductions

__device__ __forceinline__ unsigned short kp_bf16_rne(float f) {
    unsigned int u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float kp_ld(const float *p) { return *p; }
__device__ __forceinline__ float kp_ld(const unsigned short *p) { return __uint_as_float(((unsigned)*p) << 16); }

// ---- pass 0: channel-L2 normalisation (torch_edge.py:281), squared norms, node-major f32 + bf16 rows -----------------
// One thread per node for the arithmetic (the oracle's c-ascending chains); 32-channel slabs go through an LDS
// transpose so that the node-major rows are written in 128-byte (f32) / 64-byte (bf16) pieces.
template <typename T>
__global__ __launch_bounds__(256) void knn_normalize_rows_kernel(const T *__restrict__ x, int64_t sb, int64_t sc,
                                                                 float *__restrict__ xnf,
                                                                 unsigned short *__restrict__ xnh,
                                                                 float *__restrict__ sq, int C, int N, int normalize) {
    __shared__ float tile[256][33];
    const int tid = threadIdx.x, b = blockIdx.y, n0 = blockIdx.x * 256, n = n0 + tid;
    const bool valid = n < N;
    const T *xb = x + (size_t)b * sb + (valid ? n : 0);
    float den = 1.0f;
    if (normalize && valid) {
        float ss = 0.0f;
        int c = 0;
        for (; c + 8 <= C; c += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = kp_ld(xb + (size_t)(c + u) * sc);
#pragma unroll
            for (int u = 0; u < 8; ++u) ss = __builtin_fmaf(v[u], v[u], ss);
        }
        for (; c < C; ++c) {
            const float v = kp_ld(xb + (size_t)c * sc);
            ss = __builtin_fmaf(v, v, ss);
        }
        // sqrtf, not __fsqrt_rn: only the former is correctly rounded here (with
        // -fhip-fp32-correctly-rounded-divide-sqrt); the intrinsic is 1 ulp off for ~15 % of arguments
        den = fmaxf(sqrtf(ss), 1e-12f);
    }
    float q = 0.0f;
    for (int c0 = 0; c0 < C; c0 += 32) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = (valid && c0 + u < C) ? kp_ld(xb + (size_t)(c0 + u) * sc) : 0.0f;
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            if (normalize) v[u] = __fdiv_rn(v[u], den);
            if (c0 + u < C) q = __builtin_fmaf(v[u], v[u], q);
            tile[tid][u] = v[u];
        }
        __syncthreads();
        float *of = xnf + ((size_t)b * N + n0) * C + c0;
        unsigned short *oh = xnh + ((size_t)b * N + n0) * C + c0;
#pragma unroll
        for (int it = 0; it < 8; ++it) {            // 256 nodes x 8 float4
            const int i = it * 256 + tid, node = i >> 3, p = i & 7;
            if (n0 + node < N && c0 + p * 4 < C) {
                f32x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = tile[node][p * 4 + e];
                *reinterpret_cast<f32x4 *>(of + (size_t)node * C + p * 4) = w;
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {            // 256 nodes x 4 pieces of 8 bf16
            const int i = it * 256 + tid, node = i >> 2, p = i & 3;
            if (n0 + node < N && c0 + p * 8 < C) {
                u32x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    w[e] = (unsigned)kp_bf16_rne(tile[node][p * 8 + 2 * e]) |
                           ((unsigned)kp_bf16_rne(tile[node][p * 8 + 2 * e + 1]) << 16);
                *reinterpret_cast<u32x4 *>(oh + (size_t)node * C + p * 8) = w;
            }
        }
        __syncthreads();
    }
    if (valid) sq[(size_t)b * N + n] = q;
}

// (distance, index) lists of K, lexicographic insert (arbitrary arrival order)
template <int K>
struct KpTop {
    float d[K];
    int i[K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int t = 0; t < K; ++t) {
            d[t] = INFINITY;
            i[t] = 0x7fffffff;
        }
    }
    __device__ __forceinline__ void push_lex(float v, int vi) {
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const bool lt = v < d[t] || (v == d[t] && vi < i[t]);
            const float lo = lt ? v : d[t], hi = lt ? d[t] : v;
            const int ilo = lt ? vi : i[t], ihi = lt ? i[t] : vi;
            d[t] = lo; i[t] = ilo;
            v = hi; vi = ihi;
        }
    }
};

// exact squared distance of nodes (q, j): c-ascending fmaf chain over the f32 rows, then (sq_q + (-2 g)) + sq_j
__device__ __forceinline__ float kp_exact(const float *__restrict__ xf, int C, int q, int j, float sq_q, float sq_j) {
    const f32x4 *qr = reinterpret_cast<const f32x4 *>(xf + (size_t)q * C);
    const f32x4 *jr = reinterpret_cast<const f32x4 *>(xf + (size_t)j * C);
    float g = 0.0f;
    for (int c4 = 0; c4 < C / 4; c4 += 4) {         // C % 16 == 0: four float4 of each row in flight
        f32x4 a[4], bq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = jr[c4 + u];
            bq[u] = qr[c4 + u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) g = __builtin_fmaf(a[u][e], bq[u][e], g);
    }
    return __builtin_fmaf(-2.0f, g, sq_q) + sq_j;
}

template <int K, typename I>
__global__ __launch_bounds__(256, 2) void knn_pre_kernel(const float *__restrict__ xnf,
                                                         const unsigned short *__restrict__ xnh,
                                                         const float *__restrict__ sq, I *__restrict__ idx, int C,
                                                         int N, int tiles_per_clip, int nblocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char *sA = smem_b, *sB = smem_b + 2 * KP_TILE_BYTES;
    unsigned short *lists = reinterpret_cast<unsigned short *>(smem_b + 4 * KP_TILE_BYTES);
    float *sred = reinterpret_cast<float *>(smem_b + 4 * KP_TILE_BYTES + 256 * KP_CAP * 2);

    const int bid = xcd_remap(blockIdx.x, nblocks);
    const int b = bid / tiles_per_clip;
    const int q0 = (bid % tiles_per_clip) * KP_TQ;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const unsigned short *xh = xnh + (size_t)b * N * C;
    const float *xf = xnf + (size_t)b * N * C;
    const float *sqb = sq + (size_t)b * N;
    const int myq = q0 + wave * 32 + l31;
    const float sq_q = sqb[myq];

    // squared-norm range of the clip (the per-candidate norm is replaced by it in both bounds)
    {
        float mn = INFINITY, mx = 0.0f;
        for (int n = tid; n < N; n += 256) {
            const float v = sqb[n];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            mn = fminf(mn, __shfl_xor(mn, s));
            mx = fmaxf(mx, __shfl_xor(mx, s));
        }
        if (lane == 0) {
            sred[wave] = mn;
            sred[4 + wave] = mx;
        }
        __syncthreads();
    }
    const float sq_min = fminf(fminf(sred[0], sred[1]), fminf(sred[2], sred[3]));
    const float sq_max = fmaxf(fmaxf(sred[4], sred[5]), fmaxf(sred[6], sred[7]));

    const int nch = C / KP_KC, nblk = N / KP_TR, T = nblk * nch;

    // LDS-DMA of chunk t1 (candidate block t1 / nch, channel chunk t1 % nch) into buffer u & 1.  One wave instruction
    // moves 8 node rows x 128 bytes to a lane-linear destination; the 16-byte slot a lane fetches is XOR-swizzled with
    // (row >> 1) & 7 so that the ds_read_b128 of 16 different rows (one MFMA lane group) hits 64 distinct banks.
    typedef const void __attribute__((address_space(1))) *gptr_t;
    typedef void __attribute__((address_space(3))) *lptr_t;
    auto dma_chunk = [&](int t1, int u) {
        const int blk1 = t1 / nch, ch1 = t1 - blk1 * nch;
        int off = (u & 1) * KP_TILE_BYTES;
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rowbase = (wave * 4 + i) * 8;
            const int row = rowbase + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            const unsigned short *ga = xh + ((size_t)(blk1 * KP_TR + row) * C + ch1 * KP_KC + c * 8);
            const unsigned short *gb = xh + ((size_t)(q0 + row) * C + ch1 * KP_KC + c * 8);
            __builtin_amdgcn_global_load_lds((gptr_t)ga, (lptr_t)(sA + off + rowbase * 128), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)gb, (lptr_t)(sB + off + rowbase * 128), 16, 0, 0);
        }
    };
    auto frag = [&](const unsigned char *base, int row, int chunk) -> bf16x8 {
        return *reinterpret_cast<const bf16x8 *>(base + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
    };

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float gmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    float a_q = INFINITY;              // pass 2: keep iff <q^,x^> >= a_q
    int cnt = 0;
    bool overflow = false;
    unsigned short *mylist = lists + (size_t)tid * KP_CAP;

    dma_chunk(0, 0);
    __syncthreads();
    for (int u = 0; u < 2 * T; ++u) {
        const int t = u < T ? u : u - T;
        if (u + 1 < 2 * T) dma_chunk(u + 1 < T ? u + 1 : u + 1 - T, u + 1);
        int buf_off = (u & 1) * KP_TILE_BYTES;
        asm volatile("" : "+v"(buf_off));
        const unsigned char *a = sA + buf_off, *bq_ = sB + buf_off;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 bq = frag(bq_, wave * 32 + l31, 2 * s + half);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(a, tt * 32 + l31, 2 * s + half), bq, acc[tt], 0, 0, 0);
        }
        const int blk = t / nch, ch = t - blk * nch;
        if (ch == nch - 1) {            // block complete: lane holds <q^,x^> of its query with 64 candidates
            if (u < T) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) gmax[tt] = fmaxf(gmax[tt], fmaxf(acc[tt][r], acc[tt][r + 1]));
            } else {
                unsigned long long any = 0;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) any |= __ballot(acc[tt][r] >= a_q);
                if (any != 0) {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            if (acc[tt][r] >= a_q) {
                                if (cnt < KP_CAP) mylist[cnt] = (unsigned short)(blk * KP_TR + tt * 32 + mfma_row(r, half));
                                else overflow = true;
                                ++cnt;
                            }
                        }
                }
            }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tt][r] = 0.0f;
        }
        if (u == T - 1) {
            // end of pass 1: 8 upper bounds per query (4 groups x 2 half-lanes), the K-th smallest bounds the K-th best
            const float tplus = (sq_q + sq_max) * (1.0f + KP_SLACK);
            float hi[8];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                hi[g] = __builtin_fmaf(-2.0f, gmax[g], tplus);
                hi[4 + g] = __shfl_xor(hi[g], 32);
            }
            float kth[K];
#pragma unroll
            for (int t2 = 0; t2 < K; ++t2) kth[t2] = INFINITY;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                float v = hi[g];
#pragma unroll
                for (int t2 = 0; t2 < K; ++t2) {
                    const float lo = fminf(v, kth[t2]);
                    v = fmaxf(v, kth[t2]);
                    kth[t2] = lo;
                }
            }
            const float bound = fmaxf(kth[K - 1], 0.0f);
            a_q = 0.5f * ((sq_q + sq_min) * (1.0f - KP_SLACK) - bound);
        }
        __syncthreads();
    }

    // exact rescoring of the survivors (or of every candidate of this half-lane when its list overflowed)
    KpTop<K> best;
    best.init();
    const unsigned long long ovm = __ballot(overflow);          // either half-lane of the query overflowed
    const bool ov = (((ovm >> l31) | (ovm >> (l31 + 32))) & 1ull) != 0;
    if (!ov) {
        const int m = cnt;
        for (int e = 0; e < m; ++e) {
            const int j = (int)mylist[e];
            best.push_lex(kp_exact(xf, C, myq, j, sq_q, sqb[j]), j);
        }
    } else {
        for (int j = half * (N / 2); j < (half + 1) * (N / 2); ++j)
            best.push_lex(kp_exact(xf, C, myq, j, sq_q, sqb[j]), j);
    }
    // the two half-lanes hold disjoint candidate subsets of the same query: merge them
    float od[K];
    int oi[K];
#pragma unroll
    for (int t2 = 0; t2 < K; ++t2) {
        od[t2] = __shfl_xor(best.d[t2], 32);
        oi[t2] = __shfl_xor(best.i[t2], 32);
    }
#pragma unroll
    for (int t2 = 0; t2 < K; ++t2) best.push_lex(od[t2], oi[t2]);
    if (half == 0) {
        I *o = idx + ((size_t)b * N + myq) * K;
#pragma unroll
        for (int t2 = 0; t2 < K; ++t2) o[t2] = (I)best.i[t2];
    }
}

template <int K, typename I>
static void launch_pre(const float *xnf, const unsigned short *xnh, const float *sq, I *idx, int B, int C, int N,
                       hipStream_t s) {
    const int tiles = N / KP_TQ;
    const int nblocks = B * tiles;
    (void)hipFuncSetAttribute((const void *)knn_pre_kernel<K, I>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)KP_LDS_BYTES);
    hipLaunchKernelGGL((knn_pre_kernel<K, I>), dim3(nblocks), dim3(256), KP_LDS_BYTES, s, xnf, xnh, sq, idx, C, N, tiles,
                       nblocks);
}

template <typename I>
static int pre_dispatch(const float *xnf, const unsigned short *xnh, const float *sq, int B, int C, int N, int k, I *idx,
                        hipStream_t s) {
    switch (k) {
        case 1: launch_pre<1, I>(xnf, xnh, sq, idx, B, C, N, s); break;
        case 2: launch_pre<2, I>(xnf, xnh, sq, idx, B, C, N, s); break;
        case 3: launch_pre<3, I>(xnf, xnh, sq, idx, B, C, N, s); break;
        default: launch_pre<4, I>(xnf, xnh, sq, idx, B, C, N, s); break;
    }
    GRAFP_CHECK_LAUNCH("knn_pre_kernel");
    return GRAFP_OK;
}

}  // namespace grafp

extern "C" int grafp_knn_pre_supported(int C, int N, int k) {
    return (C > 0 && N > 0 && C % grafp::KP_KC == 0 && N % grafp::KP_TR == 0 && N <= 65536 && k >= 1 && k <= 4 &&
            k <= N) ? 1 : 0;
}

extern "C" size_t grafp_knn_pre_workspace(int B, int C, int N) {
    if (B <= 0 || C <= 0 || N <= 0) return 0;
    const size_t f = ((size_t)B * C * N * sizeof(float) + 255) & ~(size_t)255;
    const size_t h = ((size_t)B * C * N * sizeof(unsigned short) + 255) & ~(size_t)255;
    const size_t sq = ((size_t)B * N * sizeof(float) + 255) & ~(size_t)255;
    return f + h + sq;
}

extern "C" int grafp_knn_graph_pre(const void *x, int dtype, int64_t stride_b, int64_t stride_c, int B, int C, int N,
                                   int k, int normalize, void *idx, int idx_is_i32, void *ws, size_t ws_bytes,
                                   grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(x && idx, "knn_graph_pre: null pointer");
    GRAFP_REQUIRE(B > 0 && grafp_knn_pre_supported(C, N, k), "knn_graph_pre: unsupported shape C=%d N=%d k=%d "
                  "(C %% 64 == 0, N %% 128 == 0, k <= 4; use grafp_knn_graph_f32 otherwise)", C, N, k);
    GRAFP_REQUIRE(dtype == GRAFP_F32 || dtype == GRAFP_BF16, "knn_graph_pre: dtype %d not in {f32, bf16}", dtype);
    const size_t need = grafp_knn_pre_workspace(B, C, N);
    if (!ws || ws_bytes < need) {
        set_error("knn_graph_pre: workspace %zu bytes < required %zu", ws_bytes, need);
        return GRAFP_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    char *w = (char *)ws;
    float *xnf = (float *)w;
    w += ((size_t)B * C * N * sizeof(float) + 255) & ~(size_t)255;
    unsigned short *xnh = (unsigned short *)w;
    w += ((size_t)B * C * N * sizeof(unsigned short) + 255) & ~(size_t)255;
    float *sq = (float *)w;
    const dim3 grid((N + 255) / 256, B);
    if (dtype == GRAFP_F32)
        hipLaunchKernelGGL(knn_normalize_rows_kernel<float>, grid, dim3(256), 0, s, (const float *)x, stride_b, stride_c,
                           xnf, xnh, sq, C, N, normalize);
    else
        hipLaunchKernelGGL(knn_normalize_rows_kernel<unsigned short>, grid, dim3(256), 0, s, (const unsigned short *)x,
                           stride_b, stride_c, xnf, xnh, sq, C, N, normalize);
    GRAFP_CHECK_LAUNCH("knn_normalize_rows_kernel");
    if (idx_is_i32) return pre_dispatch<int>(xnf, xnh, sq, B, C, N, k, (int *)idx, s);
    return pre_dispatch<int64_t>(xnf, xnh, sq, B, C, N, k, (int64_t *)idx, s);
}
