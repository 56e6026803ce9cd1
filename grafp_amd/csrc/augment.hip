// augment.hip -- device-side audio augmentation, gfx950.
//
// Replaces the two torch_audiomentations transforms the reference composes at
// /root/reference/modules/transformations.py:25-48 and applies per clip on DataLoader workers (:67-75, CPU) or per
// track in validation (:98-107):
//   ApplyImpulseResponse   y[t] = sum_{l <= t} ir[l] * x[t - l], truncated to the input length (its default,
//                          compensate_for_propagation_delay=False; the library convolves by FFT)
//   AddBackgroundNoise     y = x + rms(x) / 10^(snr_db / 20) * n / (rms(n) + 1e-8)
// Here the whole batch stays in HBM: the impulse responses and noise recordings are resident RAGGED banks (one flat
// buffer + per-recording start and length: no padding to the longest file), every clip picks a recording (or none) by index, and the two transforms are three launches for the batch.
//   * ir_convolve_kernel: direct convolution on the f32 FMA pipe.  A workgroup owns 4096 consecutive outputs of one
//     clip, a thread 8 consecutive outputs in each half of the tile; the impulse response is walked in 512-tap chunks
//     staged in LDS together with the input samples they touch, stored as (lower half, upper half) pairs.  Per 8 taps a
//     thread loads 8 new sample pairs (the other 8 of its 16-pair window are the previous step's) and 8 taps (broadcast)
//     and issues 64 packed FMAs: 5.3 FMAs per LDS dword (the LDS port allows 4 at the packed-FMA peak), so the
//     kernel is FMA-bound (2*T*L flops per clip; 1 s clips with 1 s responses = 0.51 GFLOP per clip).  Every output
//     is ONE fmaf chain in increasing tap order -- the order oracle/csrc/augment.c fixes -- so results are bit-equal
//     to the oracle.  (An FFT convolution needs ~30x fewer flops at L = 16000 but two 32768-point transforms per clip
//     do not fit the 160 KB LDS in one piece; at the batch sizes of the training step the direct form costs ~1 ms.)
//   * mix_partial_kernel + mix_apply_kernel: HBM-bound (read x and noise twice, write once); the noise row is read
//     circularly from a per-clip offset; partial sums of squares are reduced in a fixed order (deterministic).
#include <math.h>

#include "common.h"

namespace grafp {

constexpr int AC_THREADS = 256;
constexpr int AC_R = 8;                         // consecutive output PAIRS per thread
constexpr int AC_H = AC_THREADS * AC_R;         // 2048: half a tile
constexpr int AC_TT = 2 * AC_H;                 // 4096 outputs per workgroup
constexpr int AC_LC = 512;                      // taps per LDS chunk
constexpr int AC_ROWS = (AC_H + AC_LC) / 8;     // LDS rows of 8 sample pairs (+ one 16-byte pad: 20 dwords per row)

typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(AC_THREADS) void ir_convolve_kernel(const float *__restrict__ x, int64_t x_stride, int T,
                                                                 const float *__restrict__ ir_bank, const int64_t *__restrict__ ir_start,
                                                                 const int *__restrict__ ir_index,
                                                                 const int *__restrict__ ir_len,
                                                                 float *__restrict__ out, int64_t out_stride) {
    // xs: sample PAIRS (x[p], x[p + AC_H]) -- output o of the lower half-tile and output o of the upper half-tile
    // meet the same taps, so one packed FMA (v_pk_fma_f32, tap broadcast) advances both chains and its operands are
    // always an aligned register pair.  Rows of 8 pairs are padded to 80 bytes: a thread's window pieces (16 bytes,
    // lanes 64 bytes of samples apart) then fall on distinct banks.
    __shared__ __attribute__((aligned(16))) float xs[AC_ROWS * 20];
    __shared__ __attribute__((aligned(16))) float hs[AC_LC];
    // clips run fastest over the grid and the late (long) tiles come first: a tile's cost grows with its position
    // (causal sum), and with tiles fastest every XCD (block id mod 8) would own ONE tile position -- 8x imbalance
    const int b = blockIdx.x, tid = threadIdx.x;
    const int t0 = (int)(gridDim.y - 1 - blockIdx.y) * AC_TT;
    const float *xb = x + (size_t)b * x_stride;
    float *ob = out + (size_t)b * out_stride;
    const int ii = ir_index ? ir_index[b] : 0;
    if (ii < 0) {                                // this clip is not convolved: copy
        for (int i = tid; i < AC_TT && t0 + i < T; i += AC_THREADS) ob[t0 + i] = xb[t0 + i];
        return;
    }
    const int L = ir_len[ii];
    const float *h = ir_bank + ir_start[ii];
    f32x2 acc[AC_R];
#pragma unroll
    for (int j = 0; j < AC_R; ++j) acc[j] = f32x2{0.0f, 0.0f};
    // taps beyond the last output index of this tile only meet samples before the start of the clip
    const int l_end = L < t0 + AC_TT ? L : t0 + AC_TT;
    for (int l0 = 0; l0 < l_end; l0 += AC_LC) {
        __syncthreads();
        for (int i = tid; i < AC_LC; i += AC_THREADS) hs[i] = (l0 + i < L) ? h[l0 + i] : 0.0f;
        const int xbase = t0 - l0 - AC_LC;       // pair i = (x[xbase + i], x[xbase + i + AC_H])
        for (int i = tid; i < AC_H + AC_LC; i += AC_THREADS) {
            const int p = xbase + i, q = p + AC_H;
            const float a = (p >= 0 && p < T) ? xb[p] : 0.0f, c = (q >= 0 && q < T) ? xb[q] : 0.0f;
            *reinterpret_cast<float2 *>(xs + 2 * i + 4 * (i >> 3)) = make_float2(a, c);
        }
        __syncthreads();
        // Tap l = l0 + u0 + u meets, for output pair o = 8*tid + j, the sample pair number o - u0 - u + AC_LC
        //   = w[j - u + 8] of the 16-pair window w[k] = pair (8*tid + AC_LC - u0 - 8 + k): one LDS row per step;
        // stepping u0 by 8 keeps w[0..7] as w[8..15].  Every output is ONE fmaf chain, taps ascending.
        const float *wp = xs + 20 * (tid + AC_LC / 8);
        f32x2 hi[8], lo[8];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float4 v = *reinterpret_cast<const float4 *>(wp + 4 * m);
            hi[2 * m] = f32x2{v.x, v.y};
            hi[2 * m + 1] = f32x2{v.z, v.w};
        }
#pragma unroll 2
        for (int u0 = 0; u0 < AC_LC; u0 += 8) {
            const float *row = wp - 20 * (u0 / 8 + 1);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float4 v = *reinterpret_cast<const float4 *>(row + 4 * m);
                lo[2 * m] = f32x2{v.x, v.y};
                lo[2 * m + 1] = f32x2{v.z, v.w};
            }
            const float4 h0 = *reinterpret_cast<const float4 *>(hs + u0), h1 = *reinterpret_cast<const float4 *>(hs + u0 + 4);
            const float hv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < AC_R; ++j) {
                    const int k = j - u + 8;      // 1..15
                    acc[j] = __builtin_elementwise_fma(f32x2{hv[u], hv[u]}, k < 8 ? lo[k] : hi[k - 8], acc[j]);
                }
#pragma unroll
            for (int k = 0; k < 8; ++k) hi[k] = lo[k];
        }
    }
    const int t = t0 + tid * AC_R;
#pragma unroll
    for (int j = 0; j < AC_R; ++j) {
        if (t + j < T) ob[t + j] = acc[j].x;
        if (t + j + AC_H < T) ob[t + j + AC_H] = acc[j].y;
    }
}

constexpr int MX_THREADS = 256;
constexpr int MX_CHUNK = MX_THREADS * 16;       // samples per workgroup

__device__ __forceinline__ float2 mx_block_sum2(float a, float b, float2 *scratch, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
    }
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = make_float2(a, b);
    __syncthreads();
    float2 r = scratch[0];
    for (int w = 1; w < MX_THREADS / 64; ++w) {
        r.x += scratch[w].x;
        r.y += scratch[w].y;
    }
    return r;
}

// part[b][s] = (sum x^2, sum n^2) over chunk s of clip b; n[t] = bank[index[b]][(offset[b] + t) mod noise_len]
__global__ __launch_bounds__(MX_THREADS) void mix_partial_kernel(const float *__restrict__ x, int64_t x_stride, int T,
                                                                 const float *__restrict__ noise_bank,
                                                                 const int64_t *__restrict__ noise_start,
                                                                 const int *__restrict__ noise_len,
                                                                 const int *__restrict__ noise_index,
                                                                 const int *__restrict__ noise_offset,
                                                                 float2 *__restrict__ part) {
    __shared__ float2 scratch[MX_THREADS / 64];
    const int b = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int ni = noise_index[b];
    if (ni < 0) return;
    const int nl = noise_len[ni];
    const float *xb = x + (size_t)b * x_stride, *nb = noise_bank + noise_start[ni];
    const int lo = s * MX_CHUNK, hi = lo + MX_CHUNK < T ? lo + MX_CHUNK : T;
    int p = (int)(((int64_t)noise_offset[b] + lo + tid) % nl);
    const int step = MX_THREADS % nl;
    float sx = 0.0f, sn = 0.0f;
    for (int t = lo + tid; t < hi; t += MX_THREADS) {
        const float xv = xb[t], nv = nb[p];
        sx = __builtin_fmaf(xv, xv, sx);
        sn = __builtin_fmaf(nv, nv, sn);
        p += step;
        if (p >= nl) p -= nl;
    }
    const float2 r = mx_block_sum2(sx, sn, scratch, tid);
    if (tid == 0) part[(size_t)b * gridDim.x + s] = r;
}

__global__ __launch_bounds__(MX_THREADS) void mix_apply_kernel(const float *__restrict__ x, int64_t x_stride, int T,
                                                               const float *__restrict__ noise_bank,
                                                               const int64_t *__restrict__ noise_start,
                                                               const int *__restrict__ noise_len,
                                                               const int *__restrict__ noise_index,
                                                               const int *__restrict__ noise_offset,
                                                               const float *__restrict__ snr_db,
                                                               const float2 *__restrict__ part,
                                                               float *__restrict__ out, int64_t out_stride) {
    const int b = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int ni = noise_index[b];
    const float *xb = x + (size_t)b * x_stride;
    float *ob = out + (size_t)b * out_stride;
    const int lo = s * MX_CHUNK, hi = lo + MX_CHUNK < T ? lo + MX_CHUNK : T;
    if (ni < 0) {
        for (int t = lo + tid; t < hi; t += MX_THREADS) ob[t] = xb[t];
        return;
    }
    float sx = 0.0f, sn = 0.0f;
    for (int i = 0; i < (int)gridDim.x; ++i) {          // same order in every workgroup of the clip
        const float2 v = part[(size_t)b * gridDim.x + i];
        sx += v.x;
        sn += v.y;
    }
    const float rms_x = sqrtf(sx / (float)T), rms_n = sqrtf(sn / (float)T);
    // AddBackgroundNoise: the background is RMS-normalised (+1e-8), then scaled to rms(x) / 10^(snr/20)
    const float scale = (rms_x / powf(10.0f, snr_db[b] / 20.0f)) / (rms_n + 1e-8f);
    const int nl = noise_len[ni];
    const float *nb = noise_bank + noise_start[ni];
    int p = (int)(((int64_t)noise_offset[b] + lo + tid) % nl);
    const int step = MX_THREADS % nl;
    for (int t = lo + tid; t < hi; t += MX_THREADS) {
        ob[t] = __builtin_fmaf(scale, nb[p], xb[t]);
        p += step;
        if (p >= nl) p -= nl;
    }
}

}  // namespace grafp

extern "C" int grafp_ir_convolve_f32(const float *x, int64_t x_stride, int B, int T, const float *ir_bank,
                                     const int64_t *ir_start, int n_ir, const int32_t *ir_len, const int32_t *ir_index,
                                     float *out, int64_t out_stride, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(x && ir_bank && ir_start && ir_len && out, "ir_convolve: null pointer");
    GRAFP_REQUIRE(B > 0 && T > 0 && n_ir > 0 && (T + AC_TT - 1) / AC_TT <= 65535, "ir_convolve: bad shape B=%d T=%d n_ir=%d", B, T, n_ir);
    GRAFP_REQUIRE(x_stride >= T && out_stride >= T, "ir_convolve: strides shorter than the rows");
    GRAFP_REQUIRE(x != out, "ir_convolve: in-place operation is not supported");
    const dim3 grid(B, (T + AC_TT - 1) / AC_TT);
    hipLaunchKernelGGL(ir_convolve_kernel, grid, dim3(AC_THREADS), 0, (hipStream_t)stream, x, x_stride, T, ir_bank,
                       ir_start, (const int *)ir_index, (const int *)ir_len, out, out_stride);
    GRAFP_CHECK_LAUNCH("ir_convolve_kernel");
    return GRAFP_OK;
}

extern "C" size_t grafp_mix_snr_workspace(int B, int T) {
    if (B <= 0 || T <= 0) return 0;
    return (size_t)B * ((size_t)(T + grafp::MX_CHUNK - 1) / grafp::MX_CHUNK) * sizeof(float2);
}

extern "C" int grafp_mix_snr_f32(const float *x, int64_t x_stride, int B, int T, const float *noise_bank,
                                 const int64_t *noise_start, int n_noise, const int32_t *noise_len,
                                 const int32_t *noise_index, const int32_t *noise_offset, const float *snr_db,
                                 float *out, int64_t out_stride, void *ws, size_t ws_bytes, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(x && noise_bank && noise_start && noise_len && noise_index && noise_offset && snr_db && out, "mix_snr: null pointer");
    GRAFP_REQUIRE(B > 0 && T > 0 && n_noise > 0 && B <= 65535, "mix_snr: bad shape B=%d T=%d n_noise=%d", B, T, n_noise);
    GRAFP_REQUIRE(x_stride >= T && out_stride >= T, "mix_snr: strides shorter than the rows");
    const size_t need = grafp_mix_snr_workspace(B, T);
    if (!ws || ws_bytes < need) {
        set_error("mix_snr: workspace %zu bytes < required %zu", ws_bytes, need);
        return GRAFP_ERR_WORKSPACE;
    }
    const dim3 grid((T + MX_CHUNK - 1) / MX_CHUNK, B);
    hipLaunchKernelGGL(mix_partial_kernel, grid, dim3(MX_THREADS), 0, (hipStream_t)stream, x, x_stride, T, noise_bank,
                       noise_start, (const int *)noise_len, (const int *)noise_index, (const int *)noise_offset,
                       (float2 *)ws);
    hipLaunchKernelGGL(mix_apply_kernel, grid, dim3(MX_THREADS), 0, (hipStream_t)stream, x, x_stride, T, noise_bank,
                       noise_start, (const int *)noise_len, (const int *)noise_index, (const int *)noise_offset, snr_db,
                       (const float2 *)ws, out, out_stride);
    GRAFP_CHECK_LAUNCH("mix_partial_kernel / mix_apply_kernel");
    return GRAFP_OK;
}
