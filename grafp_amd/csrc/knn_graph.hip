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

#include <type_traits>

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
        // sqrtf, not __fsqrt_rn: only the former is correctly rounded here (with
        // -fhip-fp32-correctly-rounded-divide-sqrt); the intrinsic is 1 ulp off for ~15 % of arguments
        den = fmaxf(sqrtf(ss), 1e-12f);
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

// One KC x 128 tile of xn (channels c0.., nodes n0..), 4 float4 per thread, zero-filled outside (C, N):
// fetched to registers first (so the loads fly under the MFMAs of the previous chunk), written to LDS later.
struct TileRegs {
    float4 v[(KC * TR / 4) / 256];
};
__device__ __forceinline__ void fetch_tile(TileRegs &t, const float *__restrict__ xb, int C, int N, int c0, int n0,
                                           int tid, bool vec_ok) {
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
        t.v[it] = v;
    }
}
__device__ __forceinline__ void store_tile(float *dst, const TileRegs &t, int tid) {
#pragma unroll
    for (int it = 0; it < (KC * TR / 4) / 256; ++it) {
        const int i = tid + it * 256;
        *reinterpret_cast<float4 *>(dst + (i / (TR / 4)) * TR + (i % (TR / 4)) * 4) = t.v[it];
    }
}

// dynamic LDS: sA[2][KC][TR] | sB[2][KC][TQ] | sSq[3][TR]   (66 KB: two workgroups per CU)
constexpr int KNN_LDS_FLOATS = 2 * KC * TR + 2 * KC * TQ + 3 * TR;

// PIPE (C a multiple of 2*KC = 64, i.e. every stage of the encoder): chunks are processed in unrolled PAIRS and the
// top-k insertion of candidate block i-1 (64 distances per lane, ~30 % of the work at C = 64) is spread over the
// 32 MFMA steps of the first chunk pair of block i, so the VALU work runs in the shadow of the matrix pipe.
// !PIPE: any C, insertion after each block.
template <int K, typename I, bool PIPE>
__global__ __launch_bounds__(256, 2) void knn_topk_kernel(const float *__restrict__ xn, const float *__restrict__ sq,
                                                          I *__restrict__ idx, int C, int N, int tiles_per_clip,
                                                          int nblocks) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float *sA = smem_f, *sB = smem_f + 2 * KC * TR, *sSq = smem_f + 2 * KC * TR + 2 * KC * TQ;

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

    const int nch = (C + KC - 1) / KC;
    const int nblk = (N + TR - 1) / TR;
    const int T = nblk * nch;

    f32x16 acc[4], prev[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.0f; prev[t][r] = 0.0f; }

    // distance of element e (tile e/16, register e%16) of a finished block; +inf beyond N: never inserted.
    // fmaf(-2, g, sq_q) == sq_q + (-2*g) exactly (the product is exact), i.e. the oracle's (sq_i + (-2 g)) + sq_j.
    auto insert = [&](const f32x16 (&a)[4], int e, int sq_base, int idx_base) {
        const int loc = (e >> 4) * 32 + mfma_row(e & 15, half);
        const float d = __builtin_fmaf(-2.0f, a[e >> 4][e & 15], sq_q) + sSq[sq_base + loc];
        best.push_ascending(d, idx_base + loc);
    };
    auto stage_next = [&](TileRegs &ra, TileRegs &rb, int t1) {   // write chunk t1 (already in registers) to LDS
        const int blk1 = t1 / nch, ch1 = t1 - blk1 * nch;
        int off = (t1 & 1) * KC * TR;
        asm volatile("" : "+v"(off));
        store_tile(sA + off, ra, tid);
        store_tile(sB + off, rb, tid);
        if (ch1 == 0 && tid < TR) {
            const int n = blk1 * TR + tid;
            sSq[(blk1 % 3) * TR + tid] = (n < N) ? sqb[n] : INFINITY;
        }
    };
    auto fetch_next = [&](TileRegs &ra, TileRegs &rb, int t1) {
        const int blk1 = t1 / nch, ch1 = t1 - blk1 * nch;
        // opaque offsets: otherwise unrolled copies keep their own pre-computed addresses live (spills)
        int c_off = ch1 * KC, n_off = blk1 * TR;
        asm volatile("" : "+v"(c_off), "+v"(n_off));
        fetch_tile(ra, xb, C, N, c_off, n_off, tid, vec_ok);
        fetch_tile(rb, xb, C, N, c_off, q0, tid, vec_ok);
    };

    if (PIPE) {
        // Tiles arrive by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass).  One wave
        // instruction moves 1 KiB = two 128-float channel rows (lanes 0-31 row r, lanes 32-63 row r+1) to a
        // wave-uniform, lane-linear LDS destination -- exactly the [KC][128] tile layout.  Shapes are exact
        // multiples here (C % 64 == 0, N % 128 == 0), so no bounds are needed.  The DMA of chunk t+1 is issued
        // before chunk t is consumed and is drained by the barrier that ends chunk t.
        typedef const void __attribute__((address_space(1))) *gptr_t;
        typedef void __attribute__((address_space(3))) *lptr_t;
        auto dma_chunk = [&](int t1) {
            const int blk1 = t1 / nch, ch1 = t1 - blk1 * nch;
            int off = (t1 & 1) * KC * TR;
            asm volatile("" : "+v"(off));
            const float *src = xb + (size_t)(ch1 * KC + half) * N + l31 * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wave * 8 + 2 * i;
                __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)row * N + blk1 * TR), (lptr_t)(sA + off + row * TR),
                                                 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)row * N + q0), (lptr_t)(sB + off + row * TR), 16, 0,
                                                 0);
            }
            if (ch1 == 0 && tid < TR) sSq[(blk1 % 3) * TR + tid] = sqb[blk1 * TR + tid];
        };
        // One unrolled pair of chunks starting at flat chunk index t0.  INS / HAS are compile-time so that every
        // MFMA step is straight-line code: HAS = this block exists (issue MFMAs + next DMA), INS = spread the 64
        // inserts of the previous block (prev) over the pair's 32 MFMA steps, two per step.
        auto chunk_pair = [&](auto INS, auto HAS, int t0, int sq_prev, int idx_prev) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const int t = t0 + c2;
                if (HAS.value && (t + 1 < T)) dma_chunk(t + 1);
                int buf_off = (t & 1) * KC * TR;
                asm volatile("" : "+v"(buf_off));
                const float *a = sA + buf_off, *bq_ = sB + buf_off;
#pragma unroll
                for (int kk = 0; kk < KC; kk += 2) {
                    if (HAS.value) {
                        const float bq = bq_[(kk + half) * TQ + wave * 32 + l31];
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt)
                            acc[tt] = mfma32x32x2(a[(kk + half) * TR + tt * 32 + l31], bq, acc[tt]);
                    }
                    if (INS.value) {
                        const int step = c2 * 16 + (kk >> 1);          // 0..31
                        insert(prev, step * 2, sq_prev, idx_prev);
                        insert(prev, step * 2 + 1, sq_prev, idx_prev);
                    }
                }
                __syncthreads();
            }
        };
        using std::true_type;
        using std::false_type;
        dma_chunk(0);
        __syncthreads();
        for (int blk = 0; blk < nblk; ++blk) {
            const int sq_prev = ((blk + 2) % 3) * TR, idx_prev = (blk - 1) * TR;
            if (blk > 0) chunk_pair(true_type{}, true_type{}, blk * nch, sq_prev, idx_prev);
            else chunk_pair(false_type{}, true_type{}, blk * nch, 0, 0);
            for (int cp = 1; cp < nch / 2; ++cp) chunk_pair(false_type{}, true_type{}, blk * nch + cp * 2, 0, 0);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                prev[tt] = acc[tt];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tt][r] = 0.0f;
            }
        }
        // drain: the last block's distances (no MFMAs left to hide behind)
#pragma unroll
        for (int e = 0; e < 64; ++e) {
            insert(prev, e, ((nblk + 2) % 3) * TR, (nblk - 1) * TR);
            if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        TileRegs ra, rb;
        fetch_next(ra, rb, 0);
        stage_next(ra, rb, 0);
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            const int blk = t / nch, ch = t - blk * nch;
            const bool more = t + 1 < T;
            if (more) fetch_next(ra, rb, t + 1);
            const float *a = sA + (t & 1) * KC * TR, *bq_ = sB + (t & 1) * KC * TQ;
#pragma unroll 4
            for (int kk = 0; kk < KC; kk += 2) {
                const float bq = bq_[(kk + half) * TQ + wave * 32 + l31];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) acc[tt] = mfma32x32x2(a[(kk + half) * TR + tt * 32 + l31], bq, acc[tt]);
            }
            if (more) stage_next(ra, rb, t + 1);
            if (ch == nch - 1) {      // block complete: lane holds G[cand][query = myq] for 64 candidates, ascending
#pragma unroll
                for (int e = 0; e < 64; ++e) {
                    insert(acc, e, (blk % 3) * TR, blk * TR);
                    if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep the 64 distances from being hoisted
                }
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tt][r] = 0.0f;
            }
            __syncthreads();
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
    const size_t lds = (size_t)KNN_LDS_FLOATS * sizeof(float);
#define KNN_LAUNCH(PIPE)                                                                                             \
    (void)hipFuncSetAttribute((const void *)knn_topk_kernel<K, I, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds);                                                                             \
    hipLaunchKernelGGL((knn_topk_kernel<K, I, PIPE>), dim3(nblocks), dim3(256), lds, s, xn, sq, idx, C, N, tiles,    \
                       nblocks)
    if (K <= 4 && C % (2 * KC) == 0 && N % TR == 0 && (((uintptr_t)xn) & 15) == 0) { KNN_LAUNCH(true); }
    else { KNN_LAUNCH(false); }
#undef KNN_LAUNCH
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
