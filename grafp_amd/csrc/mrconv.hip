// mrconv.hip -- edge gather + max-relative aggregation (K6+K7 of SURVEY.md section 2a), gfx950.
//
// Replaces, for the live 'mr' graph conv (/root/reference/encoder/gcn_lib/torch_vertex.py:19-34):
//   x_i = batched_index_select(x, centre)   torch_nn.py:79-98   (a (B,C,N,k) copy of x itself)
//   x_j = batched_index_select(x, nn_idx)                       (a (B,C,N,k) gather)
//   rel = max_k(x_j - x_i) ; out = interleave(x, rel)           torch_vertex.py:29-32
// The reference moves ~4 full-tensor copies per call through HBM (201 MB twice at B=256 stage 0); here a
// workgroup stages a slab of channel rows of one clip in LDS, gathers neighbours from LDS and writes
// the interleaved (B,2C,N) result once.  HBM-bound: 4CN + 8kN read, 8CN written per clip.
//
// Backward recomputes the arg-max neighbour (first maximum, as torch.max on CPU) and scatter-adds into
// an LDS accumulator row, so dx is written once, coalesced, with no global atomics.
#include <math.h>

#include "common.h"

namespace grafp {

constexpr int MR_THREADS = 256;

__device__ __forceinline__ int clampi(int64_t v, int n) {
    return v < 0 ? 0 : (v >= n ? n - 1 : (int)v);
}

// f32 / bf16 element access (bf16 carried as unsigned short; round-to-nearest-even on store)
__device__ __forceinline__ float mr_ld(const float *p) { return *p; }
__device__ __forceinline__ float mr_ld(const unsigned short *p) { return __uint_as_float(((unsigned)*p) << 16); }
__device__ __forceinline__ void mr_st(float *p, float v) { *p = v; }
__device__ __forceinline__ void mr_st(unsigned short *p, float v) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) { *p = (unsigned short)((u >> 16) | 0x40); return; }   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    *p = (unsigned short)(u >> 16);
}

// Activations are addressed as base + b*sb + c*sc + n (N contiguous): (B,C,N) has sb = C*N, sc = N; the
// GEMM-friendly (C,B,N) has sb = N, sc = B*N.  dynamic LDS: rows[CC*N] f32 | (bwd: acc[CC*N] f32) | sidx[K*N] i32
template <typename T>
__device__ __forceinline__ void stage_rows(float *dst, const T *__restrict__ src, int64_t sc, int cc, int N, int tid) {
    for (int i = tid; i < cc * N; i += MR_THREADS) {
        const int c = i / N, n = i - c * N;
        dst[i] = mr_ld(src + (size_t)c * sc + n);
    }
}
template <>
__device__ __forceinline__ void stage_rows<float>(float *dst, const float *__restrict__ src, int64_t sc, int cc, int N,
                                                  int tid) {
    if ((N & 3) == 0 && (sc & 3) == 0 && (((uintptr_t)src) & 15) == 0) {
        const int n4 = N >> 2;
        for (int i = tid; i < cc * n4; i += MR_THREADS) {
            const int c = i / n4, j = i - c * n4;
            reinterpret_cast<float4 *>(dst)[i] = reinterpret_cast<const float4 *>(src + (size_t)c * sc)[j];
        }
    } else {
        for (int i = tid; i < cc * N; i += MR_THREADS) {
            const int c = i / N, n = i - c * N;
            dst[i] = src[(size_t)c * sc + n];
        }
    }
}

__device__ __forceinline__ void stage_idx(int *sidx, const int64_t *__restrict__ idxb, int N, int K, int tid) {
    for (int i = tid; i < N * K; i += MR_THREADS) {
        const int n = i / K, k = i - n * K;
        sidx[k * N + n] = clampi(idxb[i], N);
    }
}

template <typename T>
__global__ __launch_bounds__(MR_THREADS) void mrconv_fwd_kernel(const T *__restrict__ x, int64_t x_sb, int64_t x_sc,
                                                                const int64_t *__restrict__ idx, T *__restrict__ out,
                                                                int64_t o_sb, int64_t o_sc, int C, int N, int K,
                                                                int CC) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y, c0 = blockIdx.x * CC, tid = threadIdx.x;
    const int cc = min(CC, C - c0);
    float *rows = reinterpret_cast<float *>(smem);
    int *sidx = reinterpret_cast<int *>(rows + (size_t)CC * N);
    stage_rows<T>(rows, x + (size_t)b * x_sb + (size_t)c0 * x_sc, x_sc, cc, N, tid);
    stage_idx(sidx, idx + (size_t)b * N * K, N, K, tid);
    __syncthreads();

    T *ob = out + (size_t)b * o_sb + (size_t)(2 * c0) * o_sc;
    int c = 0, n = tid;
    while (n >= N) { n -= N; ++c; }
    while (c < cc) {
        const float *row = rows + c * N;
        const float xi = row[n];
        float m = -INFINITY;
        for (int k = 0; k < K; ++k) m = fmaxf(m, row[sidx[k * N + n]] - xi);
        mr_st(ob + (size_t)(2 * c) * o_sc + n, xi);
        mr_st(ob + (size_t)(2 * c + 1) * o_sc + n, m);
        n += MR_THREADS;
        while (n >= N) { n -= N; ++c; }
    }
}

template <typename T>
__global__ __launch_bounds__(MR_THREADS) void mrconv_bwd_kernel(const T *__restrict__ x, int64_t x_sb, int64_t x_sc,
                                                                const int64_t *__restrict__ idx,
                                                                const T *__restrict__ gout, int64_t g_sb, int64_t g_sc,
                                                                T *__restrict__ dx, int C, int N, int K, int CC) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y, c0 = blockIdx.x * CC, tid = threadIdx.x;
    const int cc = min(CC, C - c0);
    float *rows = reinterpret_cast<float *>(smem);
    float *acc = rows + (size_t)CC * N;
    int *sidx = reinterpret_cast<int *>(acc + (size_t)CC * N);
    stage_rows<T>(rows, x + (size_t)b * x_sb + (size_t)c0 * x_sc, x_sc, cc, N, tid);
    stage_idx(sidx, idx + (size_t)b * N * K, N, K, tid);
    const T *gb = gout + (size_t)b * g_sb + (size_t)(2 * c0) * g_sc;
    {   // acc <- g_even - g_odd  (identity branch, minus the centre term of every relative difference)
        int c = 0, n = tid;
        while (n >= N) { n -= N; ++c; }
        while (c < cc) {
            acc[c * N + n] = mr_ld(gb + (size_t)(2 * c) * g_sc + n) - mr_ld(gb + (size_t)(2 * c + 1) * g_sc + n);
            n += MR_THREADS;
            while (n >= N) { n -= N; ++c; }
        }
    }
    __syncthreads();
    {   // route g_odd[c][m] to the arg-max neighbour of m (first maximum)
        int c = 0, n = tid;
        while (n >= N) { n -= N; ++c; }
        while (c < cc) {
            const float *row = rows + c * N;
            const float xi = row[n];
            float best = -INFINITY;
            int bj = sidx[n];
            for (int k = 0; k < K; ++k) {
                const int j = sidx[k * N + n];
                const float v = row[j] - xi;
                if (v > best) { best = v; bj = j; }
            }
            atomicAdd(&acc[c * N + bj], mr_ld(gb + (size_t)(2 * c + 1) * g_sc + n));
            n += MR_THREADS;
            while (n >= N) { n -= N; ++c; }
        }
    }
    __syncthreads();
    T *db = dx + (size_t)b * x_sb + (size_t)c0 * x_sc;
    for (int i = tid; i < cc * N; i += MR_THREADS) {
        const int c = i / N, n = i - c * N;
        mr_st(db + (size_t)c * x_sc + n, acc[i]);
    }
}

static int pick_cc(int C, int N, int target_elems) {
    int cc = target_elems / N;
    if (cc < 1) cc = 1;
    if (cc > C) cc = C;
    return cc;
}

}  // namespace grafp

extern "C" int grafp_mrconv_fwd_strided(const void *x, int dtype, int64_t x_sb, int64_t x_sc, const int64_t *idx, int B,
                                        int C, int N, int K, void *out, int64_t o_sb, int64_t o_sc,
                                        grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(x && idx && out, "mrconv_fwd: null pointer");
    GRAFP_REQUIRE(B > 0 && C > 0 && N > 0 && K > 0, "mrconv_fwd: bad shape B=%d C=%d N=%d K=%d", B, C, N, K);
    GRAFP_REQUIRE(dtype == GRAFP_F32 || dtype == GRAFP_BF16, "mrconv_fwd: dtype %d not in {f32, bf16}", dtype);
    const int CC = pick_cc(C, N, 8192);
    const size_t lds = ((size_t)CC * N + (size_t)K * N) * 4;
    GRAFP_REQUIRE(lds <= 160 * 1024, "mrconv_fwd: N=%d K=%d needs %zu B of LDS (> 160 KiB)", N, K, lds);
    const dim3 grid((C + CC - 1) / CC, B);
    if (dtype == GRAFP_F32) {
        (void)hipFuncSetAttribute((const void *)mrconv_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(mrconv_fwd_kernel<float>, grid, dim3(MR_THREADS), lds, (hipStream_t)stream, (const float *)x,
                           x_sb, x_sc, idx, (float *)out, o_sb, o_sc, C, N, K, CC);
    } else {
        (void)hipFuncSetAttribute((const void *)mrconv_fwd_kernel<unsigned short>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(mrconv_fwd_kernel<unsigned short>, grid, dim3(MR_THREADS), lds, (hipStream_t)stream,
                           (const unsigned short *)x, x_sb, x_sc, idx, (unsigned short *)out, o_sb, o_sc, C, N, K, CC);
    }
    GRAFP_CHECK_LAUNCH("mrconv_fwd_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_mrconv_bwd_strided(const void *x, int dtype, int64_t x_sb, int64_t x_sc, const int64_t *idx,
                                        const void *grad_out, int64_t g_sb, int64_t g_sc, int B, int C, int N, int K,
                                        void *dx, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(x && idx && grad_out && dx, "mrconv_bwd: null pointer");
    GRAFP_REQUIRE(B > 0 && C > 0 && N > 0 && K > 0, "mrconv_bwd: bad shape B=%d C=%d N=%d K=%d", B, C, N, K);
    GRAFP_REQUIRE(dtype == GRAFP_F32 || dtype == GRAFP_BF16, "mrconv_bwd: dtype %d not in {f32, bf16}", dtype);
    const int CC = pick_cc(C, N, 4096);
    const size_t lds = ((size_t)2 * CC * N + (size_t)K * N) * 4;
    GRAFP_REQUIRE(lds <= 160 * 1024, "mrconv_bwd: N=%d K=%d needs %zu B of LDS (> 160 KiB)", N, K, lds);
    const dim3 grid((C + CC - 1) / CC, B);
    if (dtype == GRAFP_F32) {
        (void)hipFuncSetAttribute((const void *)mrconv_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(mrconv_bwd_kernel<float>, grid, dim3(MR_THREADS), lds, (hipStream_t)stream, (const float *)x,
                           x_sb, x_sc, idx, (const float *)grad_out, g_sb, g_sc, (float *)dx, C, N, K, CC);
    } else {
        (void)hipFuncSetAttribute((const void *)mrconv_bwd_kernel<unsigned short>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(mrconv_bwd_kernel<unsigned short>, grid, dim3(MR_THREADS), lds, (hipStream_t)stream,
                           (const unsigned short *)x, x_sb, x_sc, idx, (const unsigned short *)grad_out, g_sb, g_sc,
                           (unsigned short *)dx, C, N, K, CC);
    }
    GRAFP_CHECK_LAUNCH("mrconv_bwd_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_mrconv_fwd_f32(const float *x, const int64_t *idx, int B, int C, int N, int K, float *out,
                                    grafp_stream_t stream) {
    return grafp_mrconv_fwd_strided(x, GRAFP_F32, (int64_t)C * N, N, idx, B, C, N, K, out, (int64_t)2 * C * N, N, stream);
}

extern "C" int grafp_mrconv_bwd_f32(const float *x, const int64_t *idx, const float *grad_out, int B, int C, int N,
                                    int K, float *dx, grafp_stream_t stream) {
    return grafp_mrconv_bwd_strided(x, GRAFP_F32, (int64_t)C * N, N, idx, grad_out, (int64_t)2 * C * N, N, B, C, N, K, dx,
                                    stream);
}
