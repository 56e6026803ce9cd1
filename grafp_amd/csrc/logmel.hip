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

// ---- NFFT = 1024 (the model's configuration), round 4: 1024 = 32 x 32, each factor a 32-point FFT in REGISTERS ----------
// The radix-2 kernel above makes ten passes over an LDS buffer with a workgroup barrier behind each (and runs its four
// frame pairs one after the other): 231 us per 2048 clips = 0.04 of the HBM roofline (BENCH_r03).  Here a HALF-WAVE owns a
// frame pair (frame f0 -> real part, f1 -> imaginary part of one complex transform):
//   pass 1  lane n2 loads x[32 n1 + n2] (coalesced, reflect padding and window on the fly), transforms over n1 in its
//           registers (5 radix-2 stages, constants folded), multiplies by W_1024^(n2 k1) -- a recurrence on W_1024^n2,
//           31 complex products instead of 31 table gathers;
//   one transposition through the half-wave's own 8.25 KB of LDS (rows padded to 33 float2: conflict-free both ways);
//   pass 2  lane k1 transforms over n2 and writes Z[k1 + 32 k2];
//   mel     lane l sums bands l and 63 - l (3 ... 42 bins each: the pair is balanced) straight from Z[k], Z[1024 - k] --
//           both frames' power spectra are formed on the fly, never stored -- then 10 log10 and the only global stores.
// No workgroup barrier anywhere: a half-wave only ever reads LDS it wrote itself, and a wave's LDS accesses complete in
// program order.  8 pairs per 256-thread workgroup, 66 KB of LDS: two workgroups per CU.
constexpr float LM_C32[16] = {1.0f, 0.98078528f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f, 0.382683432f, 0.195090322f, 0.0f, -0.195090322f, -0.382683432f, -0.555570233f, -0.707106781f, -0.831469612f, -0.923879533f, -0.98078528f};
constexpr float LM_S32[16] = {0.0f, -0.195090322f, -0.382683432f, -0.555570233f, -0.707106781f, -0.831469612f, -0.923879533f, -0.98078528f, -1.0f, -0.98078528f, -0.923879533f, -0.831469612f, -0.707106781f, -0.555570233f, -0.382683432f, -0.195090322f};     // -sin: W_32^j = exp(-2 pi i j / 32)
__device__ __forceinline__ constexpr int lm_rev5(int v) {
    return ((v & 1) << 4) | ((v & 2) << 2) | (v & 4) | ((v & 8) >> 2) | ((v & 16) >> 4);
}
// decimation in frequency, in place: on return a[p] = A[rev5(p)]
__device__ __forceinline__ void lm_fft32(float2 (&a)[32]) {
#pragma unroll
    for (int len = 32; len >= 2; len >>= 1) {
        const int half = len >> 1, tstep = 32 / len;
#pragma unroll
        for (int i = 0; i < 32; i += len)
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const float2 u = a[i + j], v = a[i + j + half];
                a[i + j] = make_float2(u.x + v.x, u.y + v.y);
                const float dx = u.x - v.x, dy = u.y - v.y;
                const float c = LM_C32[j * tstep], sn = LM_S32[j * tstep];
                a[i + j + half] = make_float2(dx * c - dy * sn, dx * sn + dy * c);
            }
    }
}

constexpr int LM1K_PAIRS = 8;                    // frame pairs (half-waves) per workgroup
constexpr int LM1K_ROW = 33;                     // float2 per transposition row
constexpr int LM1K_ITERS = 1;                    // frame pairs a half-wave does one after the other (filters staged once)
constexpr int LM1K_BW = 48;                      // widest triangular filter the kernel takes (HTK, 64 bands of 513 bins: 42)
__global__ __launch_bounds__(256, 2) void logmel1024_kernel(const float *__restrict__ wav, int64_t wav_stride, int T,
                                                            int hop, int n_mels, int n_frames,
                                                            const float *__restrict__ window,
                                                            const float2 *__restrict__ twiddle,
                                                            const float *__restrict__ fb,
                                                            const int *__restrict__ band_lo,
                                                            const int *__restrict__ band_hi, float *__restrict__ out) {
    constexpr int N = 1024;
    __shared__ float2 lds[LM1K_PAIRS][32 * LM1K_ROW];
    __shared__ float wband[64][LM1K_BW];          // band m's non-zero filter weights, bins band_lo[m] ...: 12 KB
    const int tid = threadIdx.x, l = tid & 31, g = tid >> 5, b = blockIdx.y;
    // the triangular filters once per workgroup (gathered: fb is (bins, bands)); read from global memory inside the band
    // loop below, every iteration was a dependent ~1 us gather: 215 us per 2048 clips, no faster than the radix-2 kernel
    for (int i = tid; i < n_mels * LM1K_BW; i += 256) {
        const int m = i / LM1K_BW, j = i - m * LM1K_BW, k = band_lo[m] + j;
        wband[m][j] = k <= band_hi[m] ? fb[(size_t)k * n_mels + m] : 0.0f;
    }
    __syncthreads();                              // (the only workgroup barrier; before any half-wave leaves)
    const float *x = wav + (size_t)b * wav_stride;
    float2 *const my = lds[g];
    const float2 st = twiddle[l];
    for (int it = 0; it < LM1K_ITERS; ++it) {
        const int f0 = ((blockIdx.x * LM1K_ITERS + it) * LM1K_PAIRS + g) * 2, f1 = f0 + 1;
        if (f0 >= n_frames) return;               // (a whole half-wave; nobody waits for it)
        const bool has1 = f1 < n_frames;

        float2 a[32];
#pragma unroll
        for (int n1 = 0; n1 < 32; ++n1) {
            const int t = 32 * n1 + l;
            int p0 = f0 * hop + t - N / 2, p1 = p0 + hop;
            p0 = p0 < 0 ? -p0 : (p0 >= T ? 2 * (T - 1) - p0 : p0);
            p1 = p1 < 0 ? -p1 : (p1 >= T ? 2 * (T - 1) - p1 : p1);
            p1 = p1 < 0 ? 0 : p1;                  // (frame f1 beyond the signal: any valid address; zeroed below)
            // UNCONDITIONAL loads, all 64 in flight: behind `has1 ? x[p1] : 0` hipcc put every second load into a branch of
            // its own and waited for it there -- 32 dependent trips to memory per pair, 215 us per 2048 clips
            a[n1] = make_float2(x[(unsigned)p0], x[(unsigned)p1]);      // (unsigned: a 32-bit lane offset on the uniform base)
        }
#pragma unroll
        for (int n1 = 0; n1 < 32; ++n1) {
            const float w = window[32 * n1 + l];
            a[n1] = make_float2(a[n1].x * w, has1 ? a[n1].y * w : 0.0f);
        }
        lm_fft32(a);
        {   // a[rev5(k1)] *= W_1024^(l k1), k1 ascending: w <- w * W_1024^l
            float2 w = make_float2(1.0f, 0.0f);
#pragma unroll
            for (int k1 = 0; k1 < 32; ++k1) {
                const float2 v = a[lm_rev5(k1)];
                my[k1 * LM1K_ROW + l] = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
                w = make_float2(w.x * st.x - w.y * st.y, w.x * st.y + w.y * st.x);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int n2 = 0; n2 < 32; ++n2) a[n2] = my[l * LM1K_ROW + n2];
        lm_fft32(a);                              // a[rev5(k2)] = Z[l + 32 k2]
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();          // every lane has its inputs: the buffer may be overwritten
#pragma unroll
        for (int k2 = 0; k2 < 32; ++k2) my[l + 32 * k2] = a[lm_rev5(k2)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // both frames' power spectra, bins 0 .. 512, IN PLACE: bin k is formed from Z[k] and Z[1024 - k] by ONE lane, which
        // then overwrites Z[k] with (|Xa[k]|^2, |Xb[k]|^2) -- the upper half of Z is never written, nobody else reads Z[k]
#pragma unroll
        for (int i = 0; i < 17; ++i) {
            const int k = l + 32 * i, kc = k < 513 ? k : N - 1;       // (beyond bin 512: an index nobody writes)
            const float2 z = my[kc], y = my[(N - kc) & (N - 1)];
            const float ar = 0.5f * (z.x + y.x), ai = 0.5f * (z.y - y.y);
            const float br = 0.5f * (z.y + y.y), bi = 0.5f * (y.x - z.x);
            if (k < 513) my[k] = make_float2(ar * ar + ai * ai, br * br + bi * bi);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // mel bands l and n_mels - 1 - l (n_mels <= 64: at most two bands per lane), both frames at once
        for (int half = 0; half < 2; ++half) {
            const int m = half == 0 ? l : n_mels - 1 - l;          // bands 0 .. 31 by lane, bands >= 32 from the top down
            if (half == 0 ? m >= n_mels : m < 32) continue;
            float acc0 = 0.0f, acc1 = 0.0f;
            const int lo = band_lo[m], hi = band_hi[m];
            const float *wr = wband[m];
            // four bins per trip, their LDS reads independent of one another (one bin per trip waits for its reads every
            // time: ~90 exposed LDS latencies per lane); the rows of wband are zero beyond the band
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
            for (int k = lo + LM1K_BW; k <= hi; ++k) {             // (wider filters: other settings)
                const float2 pw = my[k];
                const float wgt = fb[(size_t)k * n_mels + m];
                acc0 = __builtin_fmaf(pw.x, wgt, acc0);
                acc1 = __builtin_fmaf(pw.y, wgt, acc1);
            }
            float *o = out + ((size_t)b * n_mels + m) * n_frames + f0;
            o[0] = 10.0f * log10f(fmaxf(acc0, 1e-10f));
            if (has1) o[1] = 10.0f * log10f(fmaxf(acc1, 1e-10f));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();          // the band sums are done with the buffer before the next pair overwrites it
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
    if (n_fft == 1024 && n_mels <= 64) {          // the model's configuration: the register-FFT kernel
        const dim3 g1k((n_frames + 2 * LM1K_PAIRS * LM1K_ITERS - 1) / (2 * LM1K_PAIRS * LM1K_ITERS), B);
        hipLaunchKernelGGL(logmel1024_kernel, g1k, dim3(256), 0, s, wav, wav_stride, T, hop, n_mels, n_frames, window, tw,
                           fb, band_lo, band_hi, out);
        GRAFP_CHECK_LAUNCH("logmel1024_kernel");
        return GRAFP_OK;
    }
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
