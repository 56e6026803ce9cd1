// peak_extract.hip -- spectrogram -> point-cloud features (K2 of SURVEY.md section 2a), gfx950.
//
// Replaces GPUPeakExtractorv2.forward (/root/reference/peak_extractor.py:56-82): amin/amax/sub/div,
// two cached positional ramps, cat, Conv2d(3->F, 7x7, stride (2,1), pad 3), ReLU, reshape = ~9 launches and a
// (B,3,64,32) intermediate.  One workgroup per clip: the clip (8 KB) is staged once in LDS, min/max are
// block-reduced, the zero-padded 3-channel image [T-ramp, F-ramp, (x-min)/(max-min)] is built in LDS and the
// convolution + ReLU writes the (F, Ho*W) node features directly.  HBM: 8 192 B in, 32 768 B out per clip.
// A constant clip gives 0/0 = NaN exactly as the reference does.
#include <math.h>

#include "common.h"

namespace grafp {

constexpr int PK_THREADS = 256;

__device__ __forceinline__ float block_reduce(float v, bool is_max, float *scratch, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float u = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, u) : fminf(v, u);
    }
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    float r = scratch[0];
    for (int w = 1; w < PK_THREADS / 64; ++w) r = is_max ? fmaxf(r, scratch[w]) : fminf(r, scratch[w]);
    return r;
}

// Build img[3][HP][WP] (zero padded) in LDS from one clip.  Returns nothing; ends with a barrier.
// WP = row pitch of the padded image in floats (>= W + 2 * pw).
__device__ __forceinline__ void build_image(float *img, float *scratch, const float *__restrict__ spec, int H, int W,
                                            int ph, int pw, int WP, const float *__restrict__ t_ramp,
                                            const float *__restrict__ f_ramp, int tid) {
    const int HP = H + 2 * ph;
    float lo = INFINITY, hi = -INFINITY;
    for (int i = tid; i < H * W; i += PK_THREADS) {
        const float v = spec[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    lo = block_reduce(lo, false, scratch, tid);
    hi = block_reduce(hi, true, scratch, tid);
    const float range = hi - lo;
    for (int i = tid; i < 3 * HP * WP; i += PK_THREADS) img[i] = 0.0f;
    __syncthreads();
    for (int i = tid; i < H * W; i += PK_THREADS) {
        const int y = i / W, x = i - y * W;
        const int o = (y + ph) * WP + x + pw;
        img[o] = t_ramp[x];
        img[HP * WP + o] = f_ramp[y];
        img[2 * HP * WP + o] = __fdiv_rn(spec[i] - lo, range);
    }
    __syncthreads();
}

__global__ __launch_bounds__(PK_THREADS) void peak_fwd_kernel(const float *__restrict__ spec, int H, int W,
                                                              const float *__restrict__ weight,
                                                              const float *__restrict__ bias, int F, int KH, int KW,
                                                              int sh, int Ho, const float *__restrict__ t_ramp,
                                                              const float *__restrict__ f_ramp,
                                                              float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ph = KH / 2, pw = KW / 2, HP = H + 2 * ph, WP = W + 2 * pw;
    float *img = reinterpret_cast<float *>(smem);
    float *sw = img + 3 * HP * WP;          // weights [F][3][KH][KW]
    float *scratch = sw + F * 3 * KH * KW;  // [4]
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int i = tid; i < F * 3 * KH * KW; i += PK_THREADS) sw[i] = weight[i];
    build_image(img, scratch, spec + (size_t)b * H * W, H, W, ph, pw, WP, t_ramp, f_ramp, tid);

    float *ob = out + (size_t)b * F * Ho * W;
    for (int i = tid; i < F * Ho * W; i += PK_THREADS) {
        const int f = i / (Ho * W), r = i - f * (Ho * W);
        const int y = r / W, x = r - y * W;
        float acc = bias[f];
        const float *wf = sw + f * 3 * KH * KW;
        for (int ci = 0; ci < 3; ++ci)
            for (int ky = 0; ky < KH; ++ky) {
                const float *row = img + ci * HP * WP + (y * sh + ky) * WP + x;
                const float *wr = wf + (ci * KH + ky) * KW;
                for (int kx = 0; kx < KW; ++kx) acc = __builtin_fmaf(wr[kx], row[kx], acc);
            }
        ob[i] = fmaxf(acc, 0.0f);
    }
}

// Per-workgroup partial sums over clips b = blockIdx.x, blockIdx.x + gridDim.x, ... -> partial[blockIdx.x][nW + F];
// peak_bwd_reduce_kernel adds the workgroups in index order, so the result does not depend on which workgroup finishes
// first (it did when this ended in float atomics: two runs of one training step gave different weights).
__global__ __launch_bounds__(PK_THREADS) void peak_bwd_kernel(const float *__restrict__ spec, int B, int H, int W,
                                                              int F, int KH, int KW, int sh, int Ho,
                                                              const float *__restrict__ t_ramp,
                                                              const float *__restrict__ f_ramp,
                                                              const float *__restrict__ out,
                                                              const float *__restrict__ gout,
                                                              float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ph = KH / 2, pw = KW / 2, HP = H + 2 * ph, WP = W + 2 * pw;
    const int nW = F * 3 * KH * KW, nPos = Ho * W;
    float *img = reinterpret_cast<float *>(smem);
    float *sg = img + 3 * HP * WP;   // masked grad [F][Ho*W]
    float *sacc = sg + F * nPos;     // [nW + F]
    float *scratch = sacc + nW + F;  // [4]
    const int tid = threadIdx.x;
    for (int i = tid; i < nW + F; i += PK_THREADS) sacc[i] = 0.0f;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        build_image(img, scratch, spec + (size_t)b * H * W, H, W, ph, pw, WP, t_ramp, f_ramp, tid);
        const float *ob = out + (size_t)b * F * nPos, *gb = gout + (size_t)b * F * nPos;
        for (int i = tid; i < F * nPos; i += PK_THREADS) sg[i] = ob[i] > 0.0f ? gb[i] : 0.0f;
        __syncthreads();
        for (int wi = tid; wi < nW + F; wi += PK_THREADS) {
            float acc = 0.0f;
            if (wi < nW) {
                const int f = wi / (3 * KH * KW), r = wi - f * (3 * KH * KW);
                const int ci = r / (KH * KW), r2 = r - ci * (KH * KW);
                const int ky = r2 / KW, kx = r2 - ky * KW;
                const float *g = sg + f * nPos;
                for (int y = 0; y < Ho; ++y) {
                    const float *row = img + ci * HP * WP + (y * sh + ky) * WP + kx;
                    for (int x = 0; x < W; ++x) acc = __builtin_fmaf(g[y * W + x], row[x], acc);
                }
            } else {
                const float *g = sg + (wi - nW) * nPos;
                for (int p = 0; p < nPos; ++p) acc += g[p];
            }
            sacc[wi] += acc;
        }
    }
    __syncthreads();
    for (int wi = tid; wi < nW + F; wi += PK_THREADS) partial[(size_t)blockIdx.x * (nW + F) + wi] = sacc[wi];
}

// dweight[i] (i < nW) / dbias[i - nW] = partial[0][i] + partial[1][i] + ... in workgroup order (a fixed f32 chain).
__global__ __launch_bounds__(PK_THREADS) void peak_bwd_reduce_kernel(const float *__restrict__ partial, int nblk, int nW,
                                                                     int F, float *__restrict__ dweight,
                                                                     float *__restrict__ dbias) {
    const int i = blockIdx.x * PK_THREADS + threadIdx.x, n = nW + F;
    if (i >= n) return;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;          // four interleaved chains (latency), fixed association
    int k = 0;
    for (; k + 4 <= nblk; k += 4) {
        s0 += partial[(size_t)k * n + i];
        s1 += partial[(size_t)(k + 1) * n + i];
        s2 += partial[(size_t)(k + 2) * n + i];
        s3 += partial[(size_t)(k + 3) * n + i];
    }
    for (; k < nblk; ++k) s0 += partial[(size_t)k * n + i];
    const float s = (s0 + s1) + (s2 + s3);
    if (i < nW) dweight[i] = s;
    else dbias[i - nW] = s;
}

// ---- fast path for the model's shape: F = 8 filters, 7x7 taps, W = 32 ----------------------------------------------
// The generic kernels above read two LDS operands per multiply-add (tap weight + pixel) and are LDS-bound.  Here a
// thread owns 4 horizontally adjacent output pixels x all 8 filters (32 accumulators): per (plane, tap row) it reads 10
// pixels once and the 8 weights of a tap with two broadcast ds_read_b128 -> 224 multiply-adds per 24 LDS reads.
// Same summation order per output as the generic kernel (bias, then taps in (plane, ky, kx) order).
constexpr int PK8_F = 8, PK8_K = 7, PK8_W = 32;

__global__ __launch_bounds__(PK_THREADS) void peak_fwd8_kernel(const float *__restrict__ spec, int H,
                                                               const float *__restrict__ weight,
                                                               const float *__restrict__ bias, int sh, int Ho,
                                                               const float *__restrict__ t_ramp,
                                                               const float *__restrict__ f_ramp,
                                                               float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int W = PK8_W, K = PK8_K, F = PK8_F, ph = K / 2, pw = K / 2, WP = W + 2 * pw, NT = 3 * K * K;
    const int HP = H + 2 * ph;
    float *swT = reinterpret_cast<float *>(smem);      // [tap][8], 16-byte aligned rows
    float *img = swT + NT * F;
    float *scratch = img + 3 * HP * WP;
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int i = tid; i < NT * F; i += PK_THREADS) {
        const int f = i / NT, tap = i - f * NT;        // weight[f][tap]
        swT[tap * F + f] = weight[i];
    }
    build_image(img, scratch, spec + (size_t)b * H * W, H, W, ph, pw, WP, t_ramp, f_ramp, tid);
    float *ob = out + (size_t)b * F * Ho * W;
    for (int g = tid; g < Ho * (W / 4); g += PK_THREADS) {
        const int y = g / (W / 4), x0 = (g - y * (W / 4)) * 4;
        float acc[F][4];
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int px = 0; px < 4; ++px) acc[f][px] = bias[f];
        for (int ci = 0; ci < 3; ++ci)
            for (int ky = 0; ky < K; ++ky) {
                const float *row = img + ci * HP * WP + (y * sh + ky) * WP + x0;
                float in[K + 3];
#pragma unroll
                for (int e = 0; e < K + 3; ++e) in[e] = row[e];
                const f32x4 *w4 = reinterpret_cast<const f32x4 *>(swT + (ci * K + ky) * K * F);
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const f32x4 wa = w4[2 * kx], wb = w4[2 * kx + 1];
#pragma unroll
                    for (int px = 0; px < 4; ++px) {
#pragma unroll
                        for (int f = 0; f < 4; ++f) {
                            acc[f][px] = __builtin_fmaf(wa[f], in[kx + px], acc[f][px]);
                            acc[4 + f][px] = __builtin_fmaf(wb[f], in[kx + px], acc[4 + f][px]);
                        }
                    }
                }
            }
#pragma unroll
        for (int f = 0; f < F; ++f) {
            f32x4 o;
#pragma unroll
            for (int px = 0; px < 4; ++px) o[px] = fmaxf(acc[f][px], 0.0f);
            *reinterpret_cast<f32x4 *>(ob + (size_t)f * Ho * W + y * W + x0) = o;
        }
    }
}

// Weight gradient: thread = one (plane, tap row) x one slice of the output positions, four horizontally adjacent
// positions at a time: 3 ds_read_b128 of pixels (rows padded to 40 floats = 16-byte aligned) + 8 ds_read_b128 of the
// masked gradient (stored position-major, the 8 filters of a position adjacent) feed 224 multiply-adds issued as 112
// v_pk_fma_f32 (filter pairs packed, pixel broadcast).  Accumulators persist over the clips of the workgroup; one LDS
// reduction over the slices in slice order, then the workgroup's 1 184 sums go to ITS row of `partial` -- no atomics
// (see peak_bwd_reduce_kernel).  LDS 66 KB: two workgroups per CU.
constexpr int PK8_COMBOS = 3 * PK8_K;                       // 21
constexpr int PK8_SLICES = PK_THREADS / PK8_COMBOS;         // 12
constexpr int PK8_WPB = 40;                                 // image row pitch of the backward kernel
constexpr int PK8_PER_T = PK8_F * PK8_K + PK8_F;            // 64 values per thread in the final reduction
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(PK_THREADS) void peak_bwd8_kernel(const float *__restrict__ spec, int B, int H, int sh,
                                                               int Ho, const float *__restrict__ t_ramp,
                                                               const float *__restrict__ f_ramp,
                                                               const float *__restrict__ out,
                                                               const float *__restrict__ gout,
                                                               float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int W = PK8_W, K = PK8_K, F = PK8_F, ph = K / 2, pw = K / 2, WP = PK8_WPB;
    const int HP = H + 2 * ph, nPos = Ho * W;
    float *img = reinterpret_cast<float *>(smem);
    float *sgT = img + 3 * HP * WP;                // masked grad [nPos][F]
    float *scratch = sgT + F * nPos;               // [4]
    float *part = img;                             // [SLICES][COMBOS][PER_T], after the clip loop (img / sgT are dead)
    const int tid = threadIdx.x;
    const int combo = tid % PK8_COMBOS, slice = tid / PK8_COMBOS;
    const bool active = slice < PK8_SLICES;
    const int ci = combo / K, ky = combo - ci * K;
    f32x2 acc[F / 2][K], bsum[F / 2];
#pragma unroll
    for (int f = 0; f < F / 2; ++f) {
        bsum[f] = f32x2{0.0f, 0.0f};
#pragma unroll
        for (int kx = 0; kx < K; ++kx) acc[f][kx] = f32x2{0.0f, 0.0f};
    }
    // the zero border of the three planes is written once: the per-clip fill only touches the interior
    for (int i = tid; i < 3 * HP * WP; i += PK_THREADS) img[i] = 0.0f;
    const int nv = H * W / 4;                      // float4 pieces of a clip (W = 32: 8 per row)
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        // every global read of the clip goes out FIRST, 16 bytes per lane: the spectrogram (<= 2 pieces per thread at
        // H = 64) and, per position quad, the 8 filters' outputs and gradients (16 loads in flight per thread)
        const f32x4 *sp4 = reinterpret_cast<const f32x4 *>(spec + (size_t)b * H * W);
        const float *ob = out + (size_t)b * F * nPos, *gb = gout + (size_t)b * F * nPos;
        f32x4 sv[2];
        float lo = INFINITY, hi = -INFINITY;
        if (nv <= 2 * PK_THREADS) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = tid + j * PK_THREADS;
                sv[j] = i < nv ? sp4[i] : f32x4{INFINITY, INFINITY, INFINITY, INFINITY};
            }
        }
        __syncthreads();                           // the previous clip's products are done with img / sgT
        for (int q = tid; q < nPos / 4; q += PK_THREADS) {
            f32x4 o4[F], g4[F];
#pragma unroll
            for (int f = 0; f < F; ++f) {
                o4[f] = *reinterpret_cast<const f32x4 *>(ob + (size_t)f * nPos + 4 * q);
                g4[f] = *reinterpret_cast<const f32x4 *>(gb + (size_t)f * nPos + 4 * q);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x4 a, c;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    a[f] = o4[f][e] > 0.0f ? g4[f][e] : 0.0f;
                    c[f] = o4[4 + f][e] > 0.0f ? g4[4 + f][e] : 0.0f;
                }
                f32x4 *dst = reinterpret_cast<f32x4 *>(sgT + (size_t)(4 * q + e) * F);
                dst[0] = a;
                dst[1] = c;
            }
        }
        if (nv <= 2 * PK_THREADS) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (tid + j * PK_THREADS < nv)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lo = fminf(lo, sv[j][e]);
                        hi = fmaxf(hi, sv[j][e]);
                    }
        } else {
            for (int i = tid; i < nv; i += PK_THREADS) {
                const f32x4 v = sp4[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo = fminf(lo, v[e]);
                    hi = fmaxf(hi, v[e]);
                }
            }
        }
        lo = block_reduce(lo, false, scratch, tid);
        hi = block_reduce(hi, true, scratch, tid);
        const float range = hi - lo;
        auto fill = [&](int i, const f32x4 &v) {
            const int y = i / (W / 4), x = (i - y * (W / 4)) * 4;
            const int o = (y + ph) * WP + x + pw;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                img[o + e] = t_ramp[x + e];
                img[HP * WP + o + e] = f_ramp[y];
                img[2 * HP * WP + o + e] = __fdiv_rn(v[e] - lo, range);
            }
        };
        if (nv <= 2 * PK_THREADS) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (tid + j * PK_THREADS < nv) fill(tid + j * PK_THREADS, sv[j]);
        } else {
            for (int i = tid; i < nv; i += PK_THREADS) fill(i, sp4[i]);
        }
        __syncthreads();
        if (active) {
            const float *plane = img + ci * HP * WP + ky * WP;
            for (int g = slice; g < Ho * (W / 4); g += PK8_SLICES) {
                const int y = g / (W / 4), x0 = (g - y * (W / 4)) * 4;
                const f32x4 *row = reinterpret_cast<const f32x4 *>(plane + (y * sh) * WP + x0);
                const f32x4 r0 = row[0], r1 = row[1], r2 = row[2];
                const float in[12] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3], r2[0], r2[1], r2[2], r2[3]};
                const f32x4 *gp = reinterpret_cast<const f32x4 *>(sgT + (size_t)(y * W + x0) * F);
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    const f32x4 ga = gp[2 * px], gc = gp[2 * px + 1];
                    const f32x2 g2[F / 2] = {f32x2{ga[0], ga[1]}, f32x2{ga[2], ga[3]}, f32x2{gc[0], gc[1]},
                                             f32x2{gc[2], gc[3]}};
#pragma unroll
                    for (int f = 0; f < F / 2; ++f) {
                        bsum[f] += g2[f];
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) {
                            const float pix = in[kx + px];
                            acc[f][kx] = __builtin_elementwise_fma(g2[f], f32x2{pix, pix}, acc[f][kx]);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    if (active) {
        float *mine = part + (size_t)(slice * PK8_COMBOS + combo) * PK8_PER_T;
#pragma unroll
        for (int f = 0; f < F; ++f) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) mine[f * K + kx] = acc[f / 2][kx][f & 1];
            mine[F * K + f] = bsum[f / 2][f & 1];
        }
    }
    __syncthreads();
    constexpr int NW = F * 3 * K * K;
    float *prow = partial + (size_t)blockIdx.x * (NW + F);
    for (int i = tid; i < PK8_COMBOS * F * K + F; i += PK_THREADS) {
        float s2 = 0.0f;
        if (i < PK8_COMBOS * F * K) {
            const int cb = i / (F * K), r = i - cb * (F * K);        // r = f * K + kx
            for (int sl = 0; sl < PK8_SLICES; ++sl) s2 += part[(size_t)(sl * PK8_COMBOS + cb) * PK8_PER_T + r];
            const int f = r / K, kx = r - f * K, c2 = cb / K, ky2 = cb - c2 * K;
            prow[((f * 3 + c2) * K + ky2) * K + kx] = s2;
        } else {
            const int f = i - PK8_COMBOS * F * K;                    // bias: the slices of combo 0 cover every position once
            for (int sl = 0; sl < PK8_SLICES; ++sl) s2 += part[(size_t)(sl * PK8_COMBOS) * PK8_PER_T + F * K + f];
            prow[NW + f] = s2;
        }
    }
}

}  // namespace grafp

extern "C" int grafp_peak_extract_fwd_f32(const float *spec, int B, int H, int W, const float *weight,
                                          const float *bias, int F, int KH, int KW, int stride_h,
                                          const float *t_ramp, const float *f_ramp, float *out,
                                          grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(spec && weight && bias && t_ramp && f_ramp && out, "peak_extract_fwd: null pointer");
    GRAFP_REQUIRE(B > 0 && H > 0 && W > 0 && F > 0 && KH > 0 && KW > 0 && stride_h > 0 && (KH & 1) && (KW & 1),
                  "peak_extract_fwd: bad shape B=%d H=%d W=%d F=%d K=%dx%d stride=%d", B, H, W, F, KH, KW, stride_h);
    const int ph = KH / 2, pw = KW / 2, Ho = (H + 2 * ph - KH) / stride_h + 1;
    const size_t lds = ((size_t)3 * (H + 2 * ph) * (W + 2 * pw) + (size_t)F * 3 * KH * KW + 4) * sizeof(float);
    GRAFP_REQUIRE(lds <= 160 * 1024, "peak_extract_fwd: needs %zu B of LDS (> 160 KiB)", lds);
    if (F == PK8_F && KH == PK8_K && KW == PK8_K && W == PK8_W) {
        (void)hipFuncSetAttribute((const void *)peak_fwd8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(peak_fwd8_kernel, dim3(B), dim3(PK_THREADS), lds, (hipStream_t)stream, spec, H, weight, bias,
                           stride_h, Ho, t_ramp, f_ramp, out);
        GRAFP_CHECK_LAUNCH("peak_fwd8_kernel");
        return GRAFP_OK;
    }
    (void)hipFuncSetAttribute((const void *)peak_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(peak_fwd_kernel, dim3(B), dim3(PK_THREADS), lds, (hipStream_t)stream, spec, H, W, weight, bias, F,
                       KH, KW, stride_h, Ho, t_ramp, f_ramp, out);
    GRAFP_CHECK_LAUNCH("peak_fwd_kernel");
    return GRAFP_OK;
}

static int peak_bwd_blocks(int B) { return B < 512 ? B : 512; }

extern "C" size_t grafp_peak_extract_bwd_workspace(int B, int F, int KH, int KW) {
    if (B <= 0 || F <= 0 || KH <= 0 || KW <= 0) return 0;
    return (size_t)peak_bwd_blocks(B) * ((size_t)F * 3 * KH * KW + F) * sizeof(float);
}

extern "C" int grafp_peak_extract_bwd_f32(const float *spec, int B, int H, int W, int F, int KH, int KW, int stride_h,
                                          const float *t_ramp, const float *f_ramp, const float *out,
                                          const float *grad_out, float *dweight, float *dbias, void *workspace,
                                          size_t workspace_bytes, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(spec && t_ramp && f_ramp && out && grad_out && dweight && dbias && workspace,
                  "peak_extract_bwd: null pointer");
    GRAFP_REQUIRE(B > 0 && H > 0 && W > 0 && F > 0 && KH > 0 && KW > 0 && stride_h > 0 && (KH & 1) && (KW & 1),
                  "peak_extract_bwd: bad shape B=%d H=%d W=%d F=%d K=%dx%d stride=%d", B, H, W, F, KH, KW, stride_h);
    GRAFP_REQUIRE(workspace_bytes >= grafp_peak_extract_bwd_workspace(B, F, KH, KW),
                  "peak_extract_bwd: workspace of %zu B, needs %zu", workspace_bytes,
                  grafp_peak_extract_bwd_workspace(B, F, KH, KW));
    const int ph = KH / 2, pw = KW / 2, Ho = (H + 2 * ph - KH) / stride_h + 1;
    const size_t lds = ((size_t)3 * (H + 2 * ph) * (W + 2 * pw) + (size_t)F * Ho * W + (size_t)F * 3 * KH * KW + F + 4) *
                       sizeof(float);
    const int blocks = peak_bwd_blocks(B), nW = F * 3 * KH * KW;
    float *partial = static_cast<float *>(workspace);
    bool launched = false;
    if (F == PK8_F && KH == PK8_K && KW == PK8_K && W == PK8_W) {
        const size_t work = (size_t)3 * (H + 2 * ph) * PK8_WPB + (size_t)F * Ho * W + 4;
        const size_t fin = (size_t)PK8_SLICES * PK8_COMBOS * PK8_PER_T;
        const size_t lds8 = (work > fin ? work : fin) * sizeof(float);
        if (lds8 <= 160 * 1024) {
            (void)hipFuncSetAttribute((const void *)peak_bwd8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds8);
            hipLaunchKernelGGL(peak_bwd8_kernel, dim3(blocks), dim3(PK_THREADS), lds8, (hipStream_t)stream, spec, B, H,
                               stride_h, Ho, t_ramp, f_ramp, out, grad_out, partial);
            GRAFP_CHECK_LAUNCH("peak_bwd8_kernel");
            launched = true;
        }
    }
    if (!launched) {
        GRAFP_REQUIRE(lds <= 160 * 1024, "peak_extract_bwd: needs %zu B of LDS (> 160 KiB)", lds);
        (void)hipFuncSetAttribute((const void *)peak_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(peak_bwd_kernel, dim3(blocks), dim3(PK_THREADS), lds, (hipStream_t)stream, spec, B, H, W, F,
                           KH, KW, stride_h, Ho, t_ramp, f_ramp, out, grad_out, partial);
        GRAFP_CHECK_LAUNCH("peak_bwd_kernel");
    }
    hipLaunchKernelGGL(peak_bwd_reduce_kernel, dim3((nW + F + PK_THREADS - 1) / PK_THREADS), dim3(PK_THREADS), 0,
                       (hipStream_t)stream, partial, blocks, nW, F, dweight, dbias);
    GRAFP_CHECK_LAUNCH("peak_bwd_reduce_kernel");
    return GRAFP_OK;
}
