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

// 8-wide variants (N a multiple of 8, 16-byte aligned bases -- every Downsample of the encoder): a thread owns 8
// consecutive inputs / gradient elements of one row (= 4 outputs per tap plane), moves them with 8/16-byte accesses and
// needs no division per element.  The scalar kernels above issued one 2-byte access and two 64-bit divisions per element
// (bwd: 118 us where the bytes take ~35).
template <typename T> struct TpV;
template <> struct TpV<float> {
    __device__ static void ld4(const float *p, float (&v)[4]) { const float4 t = *reinterpret_cast<const float4 *>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    __device__ static void st4(float *p, const float (&v)[4]) { *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct TpV<unsigned short> {
    __device__ static void ld4(const unsigned short *p, float (&v)[4]) {
        const uint2 t = *reinterpret_cast<const uint2 *>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    }
    __device__ static void st4(unsigned short *p, const float (&v)[4]) {
        unsigned short h[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) tp_st(&h[i], v[i]);
        *reinterpret_cast<uint2 *>(p) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    }
};

template <typename T>
__global__ __launch_bounds__(256) void taps_fwd8_kernel(const T *__restrict__ x, int64_t rows, int N, int n_out,
                                                        T *__restrict__ out) {
    const int per_row = N >> 3;                       // 8-input pieces per row
    const int64_t total = rows * n_out, pieces = rows * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pieces; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / per_row;
        const int p = (int)(i - r * per_row), n0 = p << 3, j0 = p << 2;
        const T *xr = x + r * N + n0;
        float a[4], b[4];
        TpV<T>::ld4(xr, a);
        TpV<T>::ld4(xr + 4, b);
        const float prev = n0 > 0 ? tp_ld(xr - 1) : 0.0f;
        const float t0[4] = {prev, a[1], a[3], b[1]}, t1[4] = {a[0], a[2], b[0], b[2]}, t2[4] = {a[1], a[3], b[1], b[3]};
        T *o = out + r * n_out + j0;
        TpV<T>::st4(o, t0);
        TpV<T>::st4(o + total, t1);
        TpV<T>::st4(o + 2 * total, t2);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void taps_bwd8_kernel(const T *__restrict__ g, int64_t rows, int N, int n_out,
                                                        T *__restrict__ dx) {
    const int per_row = N >> 3;
    const int64_t total = rows * n_out, pieces = rows * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pieces; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / per_row;
        const int p = (int)(i - r * per_row), n0 = p << 3, j0 = p << 2;
        const T *g0 = g + r * n_out + j0;
        float a0[4], a1[4], a2[4];
        TpV<T>::ld4(g0, a0);                          // tap 0 of outputs j0 .. j0+3
        TpV<T>::ld4(g0 + total, a1);                  // tap 1
        TpV<T>::ld4(g0 + 2 * total, a2);              // tap 2
        const float next0 = j0 + 4 < n_out ? tp_ld(g0 + 4) : 0.0f;      // tap 0 of output j0+4
        // even n = 2j: tap 1 of output j;  odd n = 2j+1: tap 2 of output j + tap 0 of output j+1
        const float lo[4] = {a1[0], a2[0] + a0[1], a1[1], a2[1] + a0[2]};
        const float hi[4] = {a1[2], a2[2] + a0[3], a1[3], a2[3] + next0};
        T *d = dx + r * N + n0;
        TpV<T>::st4(d, lo);
        TpV<T>::st4(d + 4, hi);
    }
}

}  // namespace grafp

static int taps_launch(const void *a, void *b, int dtype, int64_t rows, int N, bool fwd, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(a && b, "stride2_taps: null pointer");
    GRAFP_REQUIRE(rows > 0 && N > 0, "stride2_taps: bad shape rows=%lld N=%d", (long long)rows, N);
    GRAFP_REQUIRE(dtype == GRAFP_F32 || dtype == GRAFP_BF16, "stride2_taps: dtype %d not in {f32, bf16}", dtype);
    const int n_out = (N - 1) / 2 + 1;
    hipStream_t s8 = (hipStream_t)stream;
    const size_t es = dtype == GRAFP_F32 ? 4 : 2;
    if (N % 8 == 0 && ((uintptr_t)a % (4 * es)) == 0 && ((uintptr_t)b % (4 * es)) == 0) {
        const int64_t pieces = rows * (N / 8), nb8 = (pieces + 255) / 256;
        const dim3 grid8((unsigned)(nb8 < 65536 ? nb8 : 65536));
        if (dtype == GRAFP_F32) {
            if (fwd) hipLaunchKernelGGL(taps_fwd8_kernel<float>, grid8, dim3(256), 0, s8, (const float *)a, rows, N, n_out, (float *)b);
            else hipLaunchKernelGGL(taps_bwd8_kernel<float>, grid8, dim3(256), 0, s8, (const float *)a, rows, N, n_out, (float *)b);
        } else {
            if (fwd) hipLaunchKernelGGL(taps_fwd8_kernel<unsigned short>, grid8, dim3(256), 0, s8, (const unsigned short *)a, rows, N, n_out, (unsigned short *)b);
            else hipLaunchKernelGGL(taps_bwd8_kernel<unsigned short>, grid8, dim3(256), 0, s8, (const unsigned short *)a, rows, N, n_out, (unsigned short *)b);
        }
        GRAFP_CHECK_LAUNCH("taps8 kernel");
        return GRAFP_OK;
    }
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
