// logmel.hip -- log-mel spectrogram (K1/K1b of SURVEY.md section 2a), gfx950.
//
// Replaces torchaudio MelSpectrogram -> AmplitudeToDB as configured at
// /root/reference/modules/transformations.py:50-57 (applied :78,:83,:89,:111): ~6 library launches and a
// complex (B,513,32) intermediate per view.  Here one workgroup owns 8 frames of one clip: samples are
// read once (coalesced, reflect-padded and windowed on the fly), two real frames ride one complex
// radix-2 Stockham FFT in LDS, power, the SPARSE triangular mel filterbank (3-42 non-zero bins per band)
// and 10*log10 are applied before the only store.  Algorithmic bytes per clip: 64 000 in + 8 192 out.
#include <math.h>

#include "common.h"

namespace grafp {

constexpr int LM_THREADS = 256;
constexpr int LM_FPB = 8;  // frames per block

template <int NFFT>
__global__ __launch_bounds__(LM_THREADS) void logmel_kernel(const float *__restrict__ wav, int64_t wav_stride, int T,
                                                            int hop, int n_mels, int n_frames,
                                                            const float *__restrict__ window,
                                                            const float2 *__restrict__ twiddle,
                                                            const float *__restrict__ fb,
                                                            const int *__restrict__ band_lo,
                                                            const int *__restrict__ band_hi, float *__restrict__ out) {
    constexpr int NB = NFFT / 2 + 1;
    __shared__ float2 bufA[NFFT];
    __shared__ float2 bufB[NFFT];
    __shared__ float2 tw[NFFT / 2];
    __shared__ float win[NFFT];
    __shared__ float pw[2][NB];

    const int tid = threadIdx.x, b = blockIdx.y;
    const int frame0 = blockIdx.x * LM_FPB;
    const float *x = wav + (size_t)b * wav_stride;
    for (int i = tid; i < NFFT / 2; i += LM_THREADS) tw[i] = twiddle[i];
    for (int i = tid; i < NFFT; i += LM_THREADS) win[i] = window[i];
    __syncthreads();

    for (int pr = 0; pr < LM_FPB / 2; ++pr) {
        const int f0 = frame0 + 2 * pr, f1 = f0 + 1;
        if (f0 >= n_frames) break;  // block-uniform
        // frame f0 -> real part, frame f1 -> imaginary part (torch.stft center=True, pad_mode='reflect')
        for (int t = tid; t < NFFT; t += LM_THREADS) {
            int p0 = f0 * hop + t - NFFT / 2, p1 = p0 + hop;
            p0 = p0 < 0 ? -p0 : (p0 >= T ? 2 * (T - 1) - p0 : p0);
            p1 = p1 < 0 ? -p1 : (p1 >= T ? 2 * (T - 1) - p1 : p1);
            const float w = win[t];
            bufA[t] = make_float2(x[p0] * w, f1 < n_frames ? x[p1] * w : 0.0f);
        }
        __syncthreads();
        // radix-2 Stockham autosort, forward transform (twiddle = exp(-2 pi i j / NFFT))
        float2 *src = bufA, *dst = bufB;
#pragma unroll 1
        for (int p = 1; p < NFFT; p <<= 1) {
            for (int i = tid; i < NFFT / 2; i += LM_THREADS) {
                const int k = i & (p - 1);
                const int j = ((i - k) << 1) + k;
                const float2 w = tw[k * (NFFT / 2 / p)];
                const float2 u0 = src[i], v = src[i + NFFT / 2];
                const float2 u1 = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
                dst[j] = make_float2(u0.x + u1.x, u0.y + u1.y);
                dst[j + p] = make_float2(u0.x - u1.x, u0.y - u1.y);
            }
            __syncthreads();
            float2 *tmp = src; src = dst; dst = tmp;
        }
        // split the two real spectra: Xa = (Z[k] + conj Z[N-k]) / 2, Xb = (Z[k] - conj Z[N-k]) / (2i)
        for (int k = tid; k < NB; k += LM_THREADS) {
            const float2 z = src[k], y = src[(NFFT - k) & (NFFT - 1)];
            const float ar = 0.5f * (z.x + y.x), ai = 0.5f * (z.y - y.y);
            const float br = 0.5f * (z.y + y.y), bi = 0.5f * (y.x - z.x);
            pw[0][k] = ar * ar + ai * ai;
            pw[1][k] = br * br + bi * bi;
        }
        __syncthreads();
        // mel bands: lanes run over bands so fb rows are read coalesced; only non-zero bins are visited
        for (int it = tid; it < 2 * n_mels; it += LM_THREADS) {
            const int fs = it / n_mels, m = it - fs * n_mels;
            const int f = f0 + fs;
            if (f < n_frames) {
                float acc = 0.0f;
                const int lo = band_lo[m], hi = band_hi[m];
                for (int k = lo; k <= hi; ++k) acc = __builtin_fmaf(pw[fs][k], fb[(size_t)k * n_mels + m], acc);
                out[((size_t)b * n_mels + m) * n_frames + f] = 10.0f * log10f(fmaxf(acc, 1e-10f));
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void unfold_kernel(const float *__restrict__ spec, int n_mels, int n_frames,
                                                     int size, int step, int n_seg, float *__restrict__ seg) {
    const size_t total = (size_t)n_seg * n_mels * size;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int j = (int)(i % size);
        const int m = (int)((i / size) % n_mels);
        const int s = (int)(i / ((size_t)size * n_mels));
        seg[i] = spec[(size_t)m * n_frames + (size_t)s * step + j];
    }
}

}  // namespace grafp

extern "C" int grafp_logmel_f32(const float *wav, int64_t wav_stride, int B, int T, int n_fft, int hop, int n_mels,
                                const float *window, const float *twiddle, const float *fb, const int32_t *band_lo,
                                const int32_t *band_hi, float *out, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(wav && window && twiddle && fb && band_lo && band_hi && out, "logmel: null pointer");
    GRAFP_REQUIRE(B > 0 && hop > 0 && n_mels > 0 && n_mels <= 256, "logmel: bad B=%d hop=%d n_mels=%d", B, hop, n_mels);
    GRAFP_REQUIRE(T > n_fft / 2, "logmel: reflect padding needs T=%d > n_fft/2=%d", T, n_fft / 2);
    GRAFP_REQUIRE(wav_stride >= T, "logmel: wav_stride %lld < T %d", (long long)wav_stride, T);
    const int n_frames = 1 + T / hop;
    const dim3 grid((n_frames + LM_FPB - 1) / LM_FPB, B);
    hipStream_t s = (hipStream_t)stream;
    const float2 *tw = reinterpret_cast<const float2 *>(twiddle);
#define LM_LAUNCH(N)                                                                                                 \
    hipLaunchKernelGGL(logmel_kernel<N>, grid, dim3(LM_THREADS), 0, s, wav, wav_stride, T, hop, n_mels, n_frames,    \
                       window, tw, fb, band_lo, band_hi, out)
    switch (n_fft) {
        case 256: LM_LAUNCH(256); break;
        case 512: LM_LAUNCH(512); break;
        case 1024: LM_LAUNCH(1024); break;
        case 2048: LM_LAUNCH(2048); break;
        default: set_error("logmel: n_fft=%d not in {256,512,1024,2048}", n_fft); return GRAFP_ERR_ARG;
    }
#undef LM_LAUNCH
    GRAFP_CHECK_LAUNCH("logmel_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_unfold_segments_f32(const float *spec, int n_mels, int n_frames, int size, int step, float *seg,
                                         grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(spec && seg, "unfold_segments: null pointer");
    GRAFP_REQUIRE(n_mels > 0 && size > 0 && step > 0 && n_frames >= size, "unfold_segments: bad n_mels=%d n_frames=%d size=%d step=%d",
                  n_mels, n_frames, size, step);
    const int n_seg = (n_frames - size) / step + 1;
    const size_t total = (size_t)n_seg * n_mels * size;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(unfold_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, spec, n_mels, n_frames, size,
                       step, n_seg, seg);
    GRAFP_CHECK_LAUNCH("unfold_kernel");
    return GRAFP_OK;
}
