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

// dynamic LDS: rows[cc*N] f32 | (bwd: acc[cc*N] f32) | sidx[K*N] i32
__device__ __forceinline__ void stage_rows(float *dst, const float *__restrict__ src, int count, int tid) {
    if ((((uintptr_t)src) & 15) == 0 && (count & 3) == 0) {
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        for (int i = tid; i < count / 4; i += MR_THREADS) d4[i] = s4[i];
    } else {
        for (int i = tid; i < count; i += MR_THREADS) dst[i] = src[i];
    }
}

__device__ __forceinline__ void stage_idx(int *sidx, const int64_t *__restrict__ idxb, int N, int K, int tid) {
    for (int i = tid; i < N * K; i += MR_THREADS) {
        const int n = i / K, k = i - n * K;
        sidx[k * N + n] = clampi(idxb[i], N);
    }
}

__global__ __launch_bounds__(MR_THREADS) void mrconv_fwd_kernel(const float *__restrict__ x,
                                                                const int64_t *__restrict__ idx,
                                                                float *__restrict__ out, int C, int N, int K, int CC) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y, c0 = blockIdx.x * CC, tid = threadIdx.x;
    const int cc = min(CC, C - c0);
    float *rows = reinterpret_cast<float *>(smem);
    int *sidx = reinterpret_cast<int *>(rows + (size_t)CC * N);
    stage_rows(rows, x + ((size_t)b * C + c0) * N, cc * N, tid);
    stage_idx(sidx, idx + (size_t)b * N * K, N, K, tid);
    __syncthreads();

    float *ob = out + ((size_t)b * 2 * C + 2 * c0) * N;
    int c = 0, n = tid;
    while (n >= N) { n -= N; ++c; }
    while (c < cc) {
        const float *row = rows + c * N;
        const float xi = row[n];
        float m = -INFINITY;
        for (int k = 0; k < K; ++k) m = fmaxf(m, row[sidx[k * N + n]] - xi);
        ob[(size_t)(2 * c) * N + n] = xi;
        ob[(size_t)(2 * c + 1) * N + n] = m;
        n += MR_THREADS;
        while (n >= N) { n -= N; ++c; }
    }
}

__global__ __launch_bounds__(MR_THREADS) void mrconv_bwd_kernel(const float *__restrict__ x,
                                                                const int64_t *__restrict__ idx,
                                                                const float *__restrict__ gout,
                                                                float *__restrict__ dx, int C, int N, int K, int CC) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y, c0 = blockIdx.x * CC, tid = threadIdx.x;
    const int cc = min(CC, C - c0);
    float *rows = reinterpret_cast<float *>(smem);
    float *acc = rows + (size_t)CC * N;
    int *sidx = reinterpret_cast<int *>(acc + (size_t)CC * N);
    stage_rows(rows, x + ((size_t)b * C + c0) * N, cc * N, tid);
    stage_idx(sidx, idx + (size_t)b * N * K, N, K, tid);
    const float *gb = gout + ((size_t)b * 2 * C + 2 * c0) * N;
    {   // acc <- g_even - g_odd  (identity branch, minus the centre term of every relative difference)
        int c = 0, n = tid;
        while (n >= N) { n -= N; ++c; }
        while (c < cc) {
            acc[c * N + n] = gb[(size_t)(2 * c) * N + n] - gb[(size_t)(2 * c + 1) * N + n];
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
            atomicAdd(&acc[c * N + bj], gb[(size_t)(2 * c + 1) * N + n]);
            n += MR_THREADS;
            while (n >= N) { n -= N; ++c; }
        }
    }
    __syncthreads();
    float *db = dx + ((size_t)b * C + c0) * N;
    for (int i = tid; i < cc * N; i += MR_THREADS) db[i] = acc[i];
}

static int pick_cc(int C, int N, int target_elems) {
    int cc = target_elems / N;
    if (cc < 1) cc = 1;
    if (cc > C) cc = C;
    return cc;
}

}  // namespace grafp

extern "C" int grafp_mrconv_fwd_f32(const float *x, const int64_t *idx, int B, int C, int N, int K, float *out,
                                    grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(x && idx && out, "mrconv_fwd: null pointer");
    GRAFP_REQUIRE(B > 0 && C > 0 && N > 0 && K > 0, "mrconv_fwd: bad shape B=%d C=%d N=%d K=%d", B, C, N, K);
    const int CC = pick_cc(C, N, 8192);
    const size_t lds = ((size_t)CC * N + (size_t)K * N) * 4;
    GRAFP_REQUIRE(lds <= 160 * 1024, "mrconv_fwd: N=%d K=%d needs %zu B of LDS (> 160 KiB)", N, K, lds);
    (void)hipFuncSetAttribute((const void *)mrconv_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mrconv_fwd_kernel, dim3((C + CC - 1) / CC, B), dim3(MR_THREADS), lds, (hipStream_t)stream, x,
                       idx, out, C, N, K, CC);
    GRAFP_CHECK_LAUNCH("mrconv_fwd_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_mrconv_bwd_f32(const float *x, const int64_t *idx, const float *grad_out, int B, int C, int N,
                                    int K, float *dx, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(x && idx && grad_out && dx, "mrconv_bwd: null pointer");
    GRAFP_REQUIRE(B > 0 && C > 0 && N > 0 && K > 0, "mrconv_bwd: bad shape B=%d C=%d N=%d K=%d", B, C, N, K);
    const int CC = pick_cc(C, N, 4096);
    const size_t lds = ((size_t)2 * CC * N + (size_t)K * N) * 4;
    GRAFP_REQUIRE(lds <= 160 * 1024, "mrconv_bwd: N=%d K=%d needs %zu B of LDS (> 160 KiB)", N, K, lds);
    (void)hipFuncSetAttribute((const void *)mrconv_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mrconv_bwd_kernel, dim3((C + CC - 1) / CC, B), dim3(MR_THREADS), lds, (hipStream_t)stream, x,
                       idx, grad_out, dx, C, N, K, CC);
    GRAFP_CHECK_LAUNCH("mrconv_bwd_kernel");
    return GRAFP_OK;
}
