// taps.hip -- the three taps of the stride-2 node convolution of Downsample (K10), gfx950.
//
// /root/reference/encoder/graph_encoder.py:16-28: Conv2d(C, 2C, 3, stride 2, padding 1) on the (N, 1) node grid.  Only
// kernel column 1 ever overlaps data, so the op is a 3-tap stride-2 convolution along N = one GEMM with K = 3C on the
// gathered taps  out[t*C + c][b][j] = x[c][b][2j + t - 1]  (zero outside [0, N)).  torch builds that operand with
// pad + three strided slices + cat (and their backward with zero-filled slice gradients): ~0.4 GB of traffic per
// Downsample at B = 512.  These two kernels read x once / write the gradient once.
#include "common.h"

namespace grafp {

__device__ __forceinline__ float tp_ld(const float *p) { return *p; }
__device__ __forceinline__ float tp_ld(const unsigned short *p) { return __uint_as_float(((unsigned)*p) << 16); }
__device__ __forceinline__ void tp_st(float *p, float v) { *p = v; }
__device__ __forceinline__ void tp_st(unsigned short *p, float v) {
    unsigned int u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    *p = (unsigned short)(u >> 16);
}

// rows = C * B contiguous rows of N (x) / n_out (each tap plane); plane stride of out = rows * n_out
template <typename T>
__global__ __launch_bounds__(256) void taps_fwd_kernel(const T *__restrict__ x, int64_t rows, int N, int n_out,
                                                       T *__restrict__ out) {
    const int64_t total = rows * n_out;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / n_out;
        const int j = (int)(i - r * n_out);
        const T *xr = x + r * N;
        const T zero = 0;
        out[i] = 2 * j - 1 >= 0 ? xr[2 * j - 1] : zero;
        out[total + i] = xr[2 * j];
        out[2 * total + i] = 2 * j + 1 < N ? xr[2 * j + 1] : zero;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void taps_bwd_kernel(const T *__restrict__ g, int64_t rows, int N, int n_out,
                                                       T *__restrict__ dx) {
    const int64_t total = rows * n_out, tx = rows * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tx; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / N;
        const int n = (int)(i - r * N);
        const T *g0 = g + r * n_out;
        float v;
        if ((n & 1) == 0) {
            v = tp_ld(g0 + total + (n >> 1));                       // tap 1 of output n/2
        } else {
            v = tp_ld(g0 + 2 * total + ((n - 1) >> 1));             // tap 2 of output (n-1)/2
            if (((n + 1) >> 1) < n_out) v += tp_ld(g0 + ((n + 1) >> 1));   // tap 0 of output (n+1)/2
        }
        tp_st(dx + i, v);
    }
}

}  // namespace grafp

static int taps_launch(const void *a, void *b, int dtype, int64_t rows, int N, bool fwd, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(a && b, "stride2_taps: null pointer");
    GRAFP_REQUIRE(rows > 0 && N > 0, "stride2_taps: bad shape rows=%lld N=%d", (long long)rows, N);
    GRAFP_REQUIRE(dtype == GRAFP_F32 || dtype == GRAFP_BF16, "stride2_taps: dtype %d not in {f32, bf16}", dtype);
    const int n_out = (N - 1) / 2 + 1;
    const int64_t work = rows * (fwd ? n_out : N);
    const int64_t nb = (work + 255) / 256;
    const dim3 grid((unsigned)(nb < 65536 ? nb : 65536));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GRAFP_F32) {
        if (fwd) hipLaunchKernelGGL(taps_fwd_kernel<float>, grid, dim3(256), 0, s, (const float *)a, rows, N, n_out, (float *)b);
        else hipLaunchKernelGGL(taps_bwd_kernel<float>, grid, dim3(256), 0, s, (const float *)a, rows, N, n_out, (float *)b);
    } else {
        if (fwd) hipLaunchKernelGGL(taps_fwd_kernel<unsigned short>, grid, dim3(256), 0, s, (const unsigned short *)a, rows, N, n_out, (unsigned short *)b);
        else hipLaunchKernelGGL(taps_bwd_kernel<unsigned short>, grid, dim3(256), 0, s, (const unsigned short *)a, rows, N, n_out, (unsigned short *)b);
    }
    GRAFP_CHECK_LAUNCH("taps kernel");
    return GRAFP_OK;
}

extern "C" int grafp_stride2_taps_fwd(const void *x, int dtype, int64_t rows, int N, void *out, grafp_stream_t stream) {
    return taps_launch(x, out, dtype, rows, N, true, stream);
}

extern "C" int grafp_stride2_taps_bwd(const void *grad_out, int dtype, int64_t rows, int N, void *dx,
                                      grafp_stream_t stream) {
    return taps_launch(grad_out, dx, dtype, rows, N, false, stream);
}
