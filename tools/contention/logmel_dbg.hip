// logmel_dbg.hip -- grafp::logmel1024_kernel (csrc/logmel.hip) with a trace: the same statements in the same order, plus,
// per half-wave (= frame pair) and stage, the XOR over its 32 lanes of the XOR of the bits of what the stage produced:
//   0 windowed samples   1 first FFT x twiddle (as written to LDS)   2 transposed read from LDS   3 second FFT
//   4 power spectrum (as written to LDS)   5 the two mel outputs per lane
// A launch whose output differs from the reference launch differs first in the stage where the wrong value entered
// (tools/contention/two_stream.py, victim "logmel_dbg").
#include "../../grafp_amd/csrc/logmel.hip"

namespace grafp {

void set_error(const char *, ...) {}               // (capi.hip's error sink: the launchers of logmel.hip refer to it)

__device__ __forceinline__ unsigned hw_xor(unsigned v) {              // over the 32 lanes of a half-wave
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v ^= (unsigned)__shfl_xor((int)v, o);
    return v;
}
__device__ __forceinline__ unsigned bits2(float2 v) { return __float_as_uint(v.x) ^ (__float_as_uint(v.y) * 0x9e3779b1u); }

__global__ __launch_bounds__(256, 2) void logmel1024_dbg_kernel(const float *__restrict__ wav, int64_t wav_stride, int T, int hop,
                                                                int n_mels, int n_frames, const float *__restrict__ window,
                                                                const float2 *__restrict__ twiddle,
                                                                const float *__restrict__ fb, const int *__restrict__ band_lo,
                                                                const int *__restrict__ band_hi, float *__restrict__ out,
                                                                unsigned *__restrict__ dbg, int variant) {
    // variant 0: the whole kernel; 1: stop after the first FFT (its values written to LDS); 2: no filter staging, no
    // workgroup barrier, no LDS at all -- loads, window, 32-point FFT and twiddle recurrence in registers, trace, return;
    // 3: as 2 with the samples made up from indices instead of loaded; 4: as 3 with the window made up as well
    constexpr int N = 1024;
    __shared__ float2 lds[LM1K_PAIRS][32 * LM1K_ROW];
    __shared__ float wband[64][LM1K_BW];
    const int tid = threadIdx.x, l = tid & 31, g = tid >> 5, b = blockIdx.y;
    if (variant < 2) {
        for (int i = tid; i < n_mels * LM1K_BW; i += 256) {
            const int m = i / LM1K_BW, j = i - m * LM1K_BW, k = band_lo[m] + j;
            wband[m][j] = k <= band_hi[m] ? fb[(size_t)k * n_mels + m] : 0.0f;
        }
        __syncthreads();
    }
    const float *x = wav + (size_t)b * wav_stride;
    float2 *const my = lds[g];
    const float2 st = twiddle[l];
    const int f0 = (blockIdx.x * LM1K_PAIRS + g) * 2, f1 = f0 + 1;
    if (f0 >= n_frames) return;
    const bool has1 = f1 < n_frames;
    unsigned *d = dbg + ((size_t)b * ((n_frames + 1) / 2) + f0 / 2) * 6;
    float2 a[32];
#pragma unroll
    for (int n1 = 0; n1 < 32; ++n1) {
        const int t = 32 * n1 + l;
        int p0 = f0 * hop + t - N / 2, p1 = p0 + hop;
        p0 = p0 < 0 ? -p0 : (p0 >= T ? 2 * (T - 1) - p0 : p0);
        p1 = p1 < 0 ? -p1 : (p1 >= T ? 2 * (T - 1) - p1 : p1);
        p1 = p1 < 0 ? 0 : p1;
        a[n1] = make_float2(x[(unsigned)p0], x[(unsigned)p1]);
        if (variant >= 3)                                      // 3, 4: no sample loads -- values made up from the indices
            a[n1] = make_float2(__sinf(0.37f * (float)(p0 & 1023)) * 0.5f, __cosf(0.11f * (float)(p1 & 1023)) * 0.5f);
    }
    unsigned h = 0;
#pragma unroll
    for (int n1 = 0; n1 < 32; ++n1) {
        const float w = variant >= 4 ? 0.5f + 0.25f * (float)((32 * n1 + l) & 15) : window[32 * n1 + l];     // 4: no window loads
        a[n1] = make_float2(a[n1].x * w, has1 ? a[n1].y * w : 0.0f);
        h ^= bits2(a[n1]) * (2u * n1 + 1u);
    }
    h = hw_xor(h);
    if (l == 0) d[0] = h;
    lm_fft32(a);
    h = 0;
    {
        float2 w = make_float2(1.0f, 0.0f);
#pragma unroll
        for (int k1 = 0; k1 < 32; ++k1) {
            const float2 v = a[lm_rev5(k1)];
            const float2 r = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
            if (variant < 2) my[k1 * LM1K_ROW + l] = r;
            h ^= bits2(r) * (2u * k1 + 1u);
            w = make_float2(w.x * st.x - w.y * st.y, w.x * st.y + w.y * st.x);
        }
    }
    h = hw_xor(h);
    if (l == 0) d[1] = h;
    if (variant) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    h = 0;
#pragma unroll
    for (int n2 = 0; n2 < 32; ++n2) {
        a[n2] = my[l * LM1K_ROW + n2];
        h ^= bits2(a[n2]) * (2u * n2 + 1u);
    }
    h = hw_xor(h);
    if (l == 0) d[2] = h;
    lm_fft32(a);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    h = 0;
#pragma unroll
    for (int k2 = 0; k2 < 32; ++k2) {
        my[l + 32 * k2] = a[lm_rev5(k2)];
        h ^= bits2(a[lm_rev5(k2)]) * (2u * k2 + 1u);
    }
    h = hw_xor(h);
    if (l == 0) d[3] = h;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    h = 0;
#pragma unroll
    for (int i = 0; i < 17; ++i) {
        const int k = l + 32 * i, kc = k < 513 ? k : N - 1;
        const float2 z = my[kc], y = my[(N - kc) & (N - 1)];
        const float ar = 0.5f * (z.x + y.x), ai = 0.5f * (z.y - y.y);
        const float br = 0.5f * (z.y + y.y), bi = 0.5f * (y.x - z.x);
        const float2 p = make_float2(ar * ar + ai * ai, br * br + bi * bi);
        if (k < 513) {
            my[k] = p;
            h ^= bits2(p) * (2u * i + 1u);
        }
    }
    h = hw_xor(h);
    if (l == 0) d[4] = h;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    h = 0;
    for (int half = 0; half < 2; ++half) {
        const int m = half == 0 ? l : n_mels - 1 - l;
        if (half == 0 ? m >= n_mels : m < 32) continue;
        float acc0 = 0.0f, acc1 = 0.0f;
        const int lo = band_lo[m], hi = band_hi[m];
        const float *wr = wband[m];
        const int nj = hi - lo + 1 < LM1K_BW ? hi - lo + 1 : LM1K_BW;
        for (int j0 = 0; j0 < nj; j0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float2 pw = my[(lo + j0 + u) & (N - 1)];
                const float wgt = wr[j0 + u];
                acc0 = __builtin_fmaf(pw.x, wgt, acc0);
                acc1 = __builtin_fmaf(pw.y, wgt, acc1);
            }
        }
        float *o = out + ((size_t)b * n_mels + m) * n_frames + f0;
        o[0] = 10.0f * log10f(fmaxf(acc0, 1e-10f));
        if (has1) o[1] = 10.0f * log10f(fmaxf(acc1, 1e-10f));
        h ^= bits2(make_float2(acc0, acc1)) * (2u * half + 1u);
    }
    h = hw_xor(h);
    if (l == 0) d[5] = h;
}

}  // namespace grafp

extern "C" int logmel1024_dbg_launch(const float *wav, int64_t wav_stride, int B, int T, int hop, int n_mels, const float *window,
                                     const float *twiddle, const float *fb, const int32_t *band_lo, const int32_t *band_hi,
                                     float *out, unsigned *dbg, int variant, void *stream) {
    using namespace grafp;
    const int n_frames = 1 + T / hop;
    const dim3 g1k((n_frames + 2 * LM1K_PAIRS - 1) / (2 * LM1K_PAIRS), B);
    hipLaunchKernelGGL(logmel1024_dbg_kernel, g1k, dim3(256), 0, (hipStream_t)stream, wav, wav_stride, T, hop, n_mels, n_frames,
                       window, reinterpret_cast<const float2 *>(twiddle), fb, band_lo, band_hi, out, dbg, variant);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
