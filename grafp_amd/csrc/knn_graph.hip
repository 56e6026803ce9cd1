// knn_graph.hip -- dynamic k-NN graph build (K3+K4+K5 of SURVEY.md section 2a), gfx950.
//
// Replaces DenseDilatedKnnGraph.forward + dense_knn_matrix + pairwise_distance
// (/root/reference/encoder/gcn_lib/torch_edge.py:270-284, 70-103, 7-18): the reference materialises
// the (B,N,N) distance matrix (1.07 GB at B=256, N=1024) and runs torch.topk over it; here a
// workgroup owns 128 query nodes of one clip, streams the clip's candidates through LDS in
// 32-channel x 128-node tiles, forms the Gram tile with exact-f32 MFMA (v_mfma_f32_32x32x2_f32:
// bitwise a c-ordered fmaf chain, which is the order oracle/csrc/knn_graph.c fixes) and keeps a
// per-lane top-k in registers.  Only (B,N,k) indices are written.
//
// Roofline: 2*N^2*C flops per clip against 4*C*N + 8*k*N bytes (63-468 flop/B) -> bound by the f32
// matrix rate (157.3 TFLOP/s), not HBM.  See DESIGN.md "knn_topk_kernel".
#include <math.h>

#include "common.h"

namespace grafp {

constexpr int TQ = 128;  // query nodes per workgroup (32 per wave)
constexpr int TR = 128;  // candidate nodes per pass
constexpr int KC = 32;   // channels per LDS chunk

// ---- pass 1: channel-L2 normalisation (torch_edge.py:281) and squared norms -------------------
// One thread per node; lanes run over consecutive nodes so every load/store is coalesced.  The input may be
// any (b, c) strided view with N contiguous -- (B,C,N) or the GEMM-friendly (C,B,N) -- in f32 or bf16; xn/sq are
// always written as contiguous (B,C,N)/(B,N) f32 for pass 2.  Loads are issued 8 channels ahead of the
// dependent fmaf chain (the chain order is still c ascending).
__device__ __forceinline__ float ld_as_f32(const float *p) { return *p; }
__device__ __forceinline__ float ld_as_f32(const unsigned short *p) { return __uint_as_float(((unsigned)*p) << 16); }

template <typename T>
__global__ __launch_bounds__(256) void knn_normalize_kernel(const T *__restrict__ x, int64_t sb, int64_t sc,
                                                            float *__restrict__ xn, float *__restrict__ sq, int C,
                                                            int N, int normalize) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= N) return;
    const T *xb = x + (size_t)b * sb + n;
    float *ob = xn + (size_t)b * C * N + n;
    float den = 1.0f;
    if (normalize) {
        float ss = 0.0f;
        int c = 0;
        for (; c + 8 <= C; c += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ld_as_f32(xb + (size_t)(c + u) * sc);
#pragma unroll
            for (int u = 0; u < 8; ++u) ss = __builtin_fmaf(v[u], v[u], ss);
        }
        for (; c < C; ++c) {
            const float v = ld_as_f32(xb + (size_t)c * sc);
            ss = __builtin_fmaf(v, v, ss);
        }
        den = fmaxf(__fsqrt_rn(ss), 1e-12f);
    }
    float q = 0.0f;
    int c = 0;
    for (; c + 8 <= C; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld_as_f32(xb + (size_t)(c + u) * sc);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (normalize) v[u] = __fdiv_rn(v[u], den);
            ob[(size_t)(c + u) * N] = v[u];
            q = __builtin_fmaf(v[u], v[u], q);
        }
    }
    for (; c < C; ++c) {
        float v = ld_as_f32(xb + (size_t)c * sc);
        if (normalize) v = __fdiv_rn(v, den);
        ob[(size_t)c * N] = v;
        q = __builtin_fmaf(v, v, q);
    }
    sq[(size_t)b * N + n] = q;
}

// ---- pass 2: Gram tiles + top-k ----------------------------------------------------------------
template <int K>
struct TopK {
    float d[K];
    int i[K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int t = 0; t < K; ++t) {
            d[t] = INFINITY;
            i[t] = 0x7fffffff;
        }
    }
    // Sorted insert as a carry chain of plain selects (branch-free).  `take` compares the ORIGINAL new
    // value with each OLD slot: in a sorted list that predicate is monotone (false...false,true...true),
    // so the first true slot receives the new element and every later slot receives its predecessor.
    // Candidates arrive in ascending index order within a lane, so strict '<' keeps the lower index on
    // ties, and a displaced (older) element always moves down regardless of ties.
    __device__ __forceinline__ void push_ascending(float v, int vi) {
        const float v0 = v;
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const bool take = v0 < d[t];
            const float od = d[t];
            const int oi = i[t];
            d[t] = take ? v : od;
            i[t] = take ? vi : oi;
            v = take ? od : v;
            vi = take ? oi : vi;
        }
    }
    // arbitrary order: full (distance, index) lexicographic comparison
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

// Stage a KC x 128 tile of xn (channels c0.., nodes n0..) into LDS; zero-fill outside (C, N).
__device__ __forceinline__ void stage_tile(float (*dst)[TR], const float *__restrict__ xb, int C, int N, int c0,
                                           int n0, int tid, bool vec_ok) {
#pragma unroll
    for (int it = 0; it < (KC * TR / 4) / 256; ++it) {
        const int i = tid + it * 256;
        const int row = i / (TR / 4), c4 = i % (TR / 4);
        const int c = c0 + row, n = n0 + c4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C) {
            const float *src = xb + (size_t)c * N + n;
            if (vec_ok && n + 3 < N) {
                v = *reinterpret_cast<const float4 *>(src);
            } else {
                if (n + 0 < N) v.x = src[0];
                if (n + 1 < N) v.y = src[1];
                if (n + 2 < N) v.z = src[2];
                if (n + 3 < N) v.w = src[3];
            }
        }
        *reinterpret_cast<float4 *>(&dst[row][c4 * 4]) = v;
    }
}

template <int K, typename I>
__global__ __launch_bounds__(256, 2) void knn_topk_kernel(const float *__restrict__ xn, const float *__restrict__ sq,
                                                          I *__restrict__ idx, int C, int N,
                                                          int tiles_per_clip, int nblocks) {
    __shared__ __attribute__((aligned(16))) float sA[KC][TR];  // candidates
    __shared__ __attribute__((aligned(16))) float sB[KC][TQ];  // queries
    __shared__ float sSq[TR];

    const int bid = xcd_remap(blockIdx.x, nblocks);
    const int b = bid / tiles_per_clip;
    const int q0 = (bid % tiles_per_clip) * TQ;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const float *xb = xn + (size_t)b * C * N;
    const float *sqb = sq + (size_t)b * N;
    const bool vec_ok = (N & 3) == 0;

    const int myq = q0 + wave * 32 + l31;
    const float sq_q = myq < N ? sqb[myq] : 0.0f;
    TopK<K> best;
    best.init();

    for (int r0 = 0; r0 < N; r0 += TR) {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

        for (int c0 = 0; c0 < C; c0 += KC) {
            __syncthreads();  // previous chunk (and previous pass's sSq) fully consumed
            stage_tile(sA, xb, C, N, c0, r0, tid, vec_ok);
            stage_tile(sB, xb, C, N, c0, q0, tid, vec_ok);
            if (c0 == 0 && tid < TR) sSq[tid] = (r0 + tid < N) ? sqb[r0 + tid] : INFINITY;
            __syncthreads();
#pragma unroll 4
            for (int kk = 0; kk < KC; kk += 2) {
                const float bq = sB[kk + half][wave * 32 + l31];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = mfma32x32x2(sA[kk + half][t * 32 + l31], bq, acc[t]);
            }
        }
        // lane holds G[cand][query = myq] for 64 candidates, visited in ascending candidate order
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int loc = t * 32 + mfma_row(r, half);
                const float d = (sq_q + (-2.0f * acc[t][r])) + sSq[loc];  // +inf beyond N: never inserted
                best.push_ascending(d, r0 + loc);
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // keep the 64 distances from being hoisted
            }
        }
    }
    // the two half-waves saw disjoint candidate subsets of the same query: merge them
    float od[K];
    int oi[K];
#pragma unroll
    for (int t = 0; t < K; ++t) {
        od[t] = __shfl_xor(best.d[t], 32);
        oi[t] = __shfl_xor(best.i[t], 32);
    }
#pragma unroll
    for (int t = 0; t < K; ++t) best.push_lex(od[t], oi[t]);
    if (half == 0 && myq < N) {
        I *o = idx + ((size_t)b * N + myq) * K;
#pragma unroll
        for (int t = 0; t < K; ++t) o[t] = (I)best.i[t];
    }
}

template <int K, typename I>
static void launch_topk(const float *xn, const float *sq, I *idx, int B, int C, int N, hipStream_t s) {
    const int tiles = (N + TQ - 1) / TQ;
    const int nblocks = B * tiles;
    hipLaunchKernelGGL((knn_topk_kernel<K, I>), dim3(nblocks), dim3(256), 0, s, xn, sq, idx, C, N, tiles, nblocks);
}

}  // namespace grafp

extern "C" size_t grafp_knn_graph_workspace(int B, int C, int N) {
    if (B <= 0 || C <= 0 || N <= 0) return 0;
    const size_t xn = ((size_t)B * C * N * sizeof(float) + 255) & ~(size_t)255;
    const size_t sq = ((size_t)B * N * sizeof(float) + 255) & ~(size_t)255;
    return xn + sq;
}

template <typename I>
static int knn_topk_launch(const float *xn, const float *sq, int B, int C, int N, int k, I *idx, hipStream_t s) {
    using namespace grafp;
    switch (k) {
        case 1: launch_topk<1, I>(xn, sq, idx, B, C, N, s); break;
        case 2: launch_topk<2, I>(xn, sq, idx, B, C, N, s); break;
        case 3: launch_topk<3, I>(xn, sq, idx, B, C, N, s); break;
        case 4: launch_topk<4, I>(xn, sq, idx, B, C, N, s); break;
        case 5: launch_topk<5, I>(xn, sq, idx, B, C, N, s); break;
        case 6: launch_topk<6, I>(xn, sq, idx, B, C, N, s); break;
        case 7: launch_topk<7, I>(xn, sq, idx, B, C, N, s); break;
        default: launch_topk<8, I>(xn, sq, idx, B, C, N, s); break;
    }
    GRAFP_CHECK_LAUNCH("knn_topk_kernel");
    return GRAFP_OK;
}

static bool knn_args_ok(const void *a, const void *b, int B, int C, int N, int k) {
    using namespace grafp;
    if (!a || !b) { set_error("knn_graph: null pointer"); return false; }
    if (B <= 0 || C <= 0 || N <= 0) { set_error("knn_graph: bad shape B=%d C=%d N=%d", B, C, N); return false; }
    if (k < 1 || k > GRAFP_KNN_MAX_K || k > N) {
        set_error("knn_graph: k=%d must be in [1, min(N=%d, %d)]", k, N, GRAFP_KNN_MAX_K);
        return false;
    }
    return true;
}

extern "C" int grafp_knn_normalize_strided(const void *x, int dtype, int64_t stride_b, int64_t stride_c, int B, int C,
                                           int N, int normalize, float *xn, float *sq, grafp_stream_t stream) {
    using namespace grafp;
    if (!knn_args_ok(x, xn, B, C, N, 1) || !knn_args_ok(x, sq, B, C, N, 1)) return GRAFP_ERR_ARG;
    GRAFP_REQUIRE(dtype == GRAFP_F32 || dtype == GRAFP_BF16, "knn_normalize: dtype %d not in {f32, bf16}", dtype);
    const dim3 grid((N + 255) / 256, B);
    if (dtype == GRAFP_F32)
        hipLaunchKernelGGL(knn_normalize_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)x,
                           stride_b, stride_c, xn, sq, C, N, normalize);
    else
        hipLaunchKernelGGL(knn_normalize_kernel<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short *)x, stride_b, stride_c, xn, sq, C, N, normalize);
    GRAFP_CHECK_LAUNCH("knn_normalize_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_knn_normalize_f32(const float *x, int B, int C, int N, int normalize, float *xn, float *sq,
                                       grafp_stream_t stream) {
    return grafp_knn_normalize_strided(x, GRAFP_F32, (int64_t)C * N, N, B, C, N, normalize, xn, sq, stream);
}

extern "C" int grafp_knn_topk_f32(const float *xn, const float *sq, int B, int C, int N, int k, int64_t *idx,
                                  grafp_stream_t stream) {
    if (!knn_args_ok(xn, idx, B, C, N, k) || !knn_args_ok(sq, idx, B, C, N, k)) return GRAFP_ERR_ARG;
    return knn_topk_launch<int64_t>(xn, sq, B, C, N, k, idx, (hipStream_t)stream);
}

extern "C" int grafp_knn_topk_i32(const float *xn, const float *sq, int B, int C, int N, int k, int32_t *idx,
                                  grafp_stream_t stream) {
    if (!knn_args_ok(xn, idx, B, C, N, k) || !knn_args_ok(sq, idx, B, C, N, k)) return GRAFP_ERR_ARG;
    return knn_topk_launch<int32_t>(xn, sq, B, C, N, k, idx, (hipStream_t)stream);
}

extern "C" int grafp_knn_graph_f32(const float *x, int B, int C, int N, int k, int normalize, int64_t *idx, void *ws,
                                   size_t ws_bytes, grafp_stream_t stream) {
    using namespace grafp;
    if (!knn_args_ok(x, idx, B, C, N, k)) return GRAFP_ERR_ARG;
    const size_t need = grafp_knn_graph_workspace(B, C, N);
    if (!ws || ws_bytes < need) {
        set_error("knn_graph: workspace %zu bytes < required %zu", ws_bytes, need);
        return GRAFP_ERR_WORKSPACE;
    }
    float *xn = (float *)ws;
    float *sq = (float *)((char *)ws + (((size_t)B * C * N * sizeof(float) + 255) & ~(size_t)255));
    const int rc = grafp_knn_normalize_f32(x, B, C, N, normalize, xn, sq, stream);
    if (rc != GRAFP_OK) return rc;
    return grafp_knn_topk_f32(xn, sq, B, C, N, k, idx, stream);
}
