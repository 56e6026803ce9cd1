// bn.hip -- fused [conv bias] + BatchNorm + activation + residual on the (C, M = B*N) activation layout, gfx950.
//
// Replaces, around every 1x1 convolution of the GraFPrint encoder, the chain the reference runs as separate
// library/elementwise launches: `+ bias` (Conv2d bias), BatchNorm2d (batch statistics in train mode,
// /root/reference/encoder/gcn_lib/torch_vertex.py:152-162, torch_nn.py:56-60, encoder/graph_encoder.py:52-55,131-133),
// ReLU / LeakyReLU(0.2), and the residual add (`torch_vertex.py:193`, `graph_encoder.py:65`).
// With channels as ROWS of a (C, M) matrix a channel's statistics are a reduction over one contiguous row:
//   forward  = stats pass (1 read) + apply pass (1 read [+1 residual read] + 1 write)
//   backward = reduce pass (2 reads) + dx pass (2 reads + 1 write); the activation mask is recomputed, not stored.
// HBM-bound: 12 (f32) / 6 (bf16) bytes per element forward.  Statistics use shifted sums (shift = first element
// of the row) in f32, which removes the E[x^2] - E[x]^2 cancellation for rows with |mean| >> std.
// A conv bias in front of a train-mode BatchNorm cancels in the output; it only shifts the running mean, and its
// gradient is exactly zero -- so it is folded in here instead of costing an elementwise launch + a reduction.
#include <math.h>

#include <type_traits>

#define GRAFP_STORE_FAMILY 1        // (common.h: GRAFP_ST_NT experiment builds)
#include "common.h"
#include "tuning.h"

namespace grafp {

constexpr int BN_THREADS = 256;

template <typename T> struct BnIO;
template <> struct BnIO<float> {
    static constexpr int W = 4;
    __device__ static void load(const float *p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    // outputs are streamed with the non-temporal hint: a plain-store copy of a 67-268 MB tensor runs at 3.7-4.9 TB/s
    // on MI355X, the same copy with `nt` stores at 6.2-6.7 TB/s (tools/microbench/copy_bench.hip)
    __device__ static void store(float *p, const float (&v)[4], bool plain = false) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 t = {v[0], v[1], v[2], v[3]};
        if (plain) store16_hint(p, __builtin_bit_cast(st_u32x4, t), true);      // (wave-uniform: see bn_plain_stores)
        else GRAFP_ST_NT(t, reinterpret_cast<f4 *>(p));
    }
    using Raw = float4;
    __device__ static void unpack(const float4 &t, float (&v)[4]) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    __device__ static float ld1(const float *p) { return *p; }
    __device__ static void st1(float *p, float v) { *p = v; }
};
__device__ __forceinline__ unsigned short f2bf(float v) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
template <> struct BnIO<unsigned short> {
    static constexpr int W = 8;
    __device__ static void load(const unsigned short *p, float (&v)[8]) {
        const uint4 t = *reinterpret_cast<const uint4 *>(p);
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ static void store(unsigned short *p, const float (&v)[8], bool plain = false) {
        // v_cvt_pk_bf16_f32 (round to nearest even, the conversion the GEMM epilogues use): one instruction per PAIR instead
        // of the six of the integer form per value -- a seventh of this file's backward kernel's vector instructions
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f2 pr = {v[2 * i], v[2 * i + 1]};
            w[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, b2));
        }
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 t = {w[0], w[1], w[2], w[3]};
        if (plain) store16_hint(p, __builtin_bit_cast(st_u32x4, t), true);
        else GRAFP_ST_NT(t, reinterpret_cast<u4 *>(p));
    }
    using Raw = uint4;
    __device__ static void unpack(const uint4 &t, float (&v)[8]) {
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ static float ld1(const unsigned short *p) { return __uint_as_float(((unsigned)*p) << 16); }
    __device__ static void st1(unsigned short *p, float v) { *p = f2bf(v); }
};

// dY of the BatchNorm backward is read twice by the two launches that follow it (data gradient and weight gradient of
// the convolution in front).  A tensor that fits the 256 MB Infinity Cache with room for the other operand is written
// with PLAIN stores (it stays cached for its two readers); larger ones keep the streaming hint, which wins there by
// sparing the producers' working set.  Same-box A/B of the whole step (tools/step_lib_ab.py, profiles/r06_c_*, r06_d_*):
// plain stores in this file -1.15 % at 128 pairs, -1.05 % at 256, +0.45 % at 512, +1.2 % at 1024; the threshold between
// them from tools/step_env_graph_ab.py (profiles/r06_e_bn_plain_threshold.txt: tensors up to 70 / 140 / 280 MB / all
// plain: 128 pairs -0.8 / -0.7 / -0.8 / -0.8 %, 256 pairs -0.8 / -1.2 / -0.6 / -0.7 %, 512 pairs +0.2 / -0.1 / +0.8 /
// +1.1 %): 140 MB.  A pure function of the tensor size.
static int bn_plain_stores(size_t bytes) {
    return bytes <= ((size_t)GRAFP_TUNE_INT("GRAFP_BN_BWD_PLAIN_MAX_MB", 140) << 20) ? 1 : 0;
}

template <int THREADS = 256>
__device__ __forceinline__ float2 block_sum2(float a, float b, float2 *scratch, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
    }
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = make_float2(a, b);
    __syncthreads();
    float2 r = scratch[0];
    for (int w = 1; w < THREADS / 64; ++w) {
        r.x += scratch[w].x;
        r.y += scratch[w].y;
    }
    return r;
}

__device__ __forceinline__ float act_fwd(float u, int act, float slope) {
    return act == 0 ? u : (u > 0.0f ? u : (act == 1 ? 0.0f : u * slope));
}
__device__ __forceinline__ float act_grad(float u, int act, float slope) {
    return act == 0 ? 1.0f : (u > 0.0f ? 1.0f : (act == 1 ? 0.0f : slope));
}

// ---- forward pass 1: partial shifted sums --------------------------------------------------------
template <typename T, bool VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(const T *__restrict__ x, int64_t M, int64_t Mg,
                                                              int64_t chunk, int Sg,
                                                              const float *__restrict__ pre_bias,
                                                              float *__restrict__ part) {
    __shared__ float2 scratch[BN_THREADS / 64];
    const int c = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int S = gridDim.x, g = s / Sg, sl = s - g * Sg;
    const T *row = x + (size_t)c * M;
    const float pb = pre_bias ? pre_bias[c] : 0.0f;
    const float shift = BnIO<T>::ld1(row + (int64_t)g * Mg) + pb;
    const int64_t gend = (int64_t)(g + 1) * Mg;
    const int64_t lo = (int64_t)g * Mg + (int64_t)sl * chunk, hi = (lo + chunk < gend) ? lo + chunk : gend;
    float a = 0.0f, q = 0.0f;
    constexpr int W = VEC ? BnIO<T>::W : 1;
    for (int64_t m = lo + (int64_t)tid * W; m < hi; m += (int64_t)BN_THREADS * W) {
        if (VEC && m + W <= hi) {
            float v[BnIO<T>::W];
            BnIO<T>::load(row + m, v);
#pragma unroll
            for (int i = 0; i < BnIO<T>::W; ++i) {
                const float d = (v[i] + pb) - shift;
                a += d;
                q = __builtin_fmaf(d, d, q);
            }
        } else {
            for (int i = 0; i < W && m + i < hi; ++i) {
                const float d = (BnIO<T>::ld1(row + m + i) + pb) - shift;
                a += d;
                q = __builtin_fmaf(d, d, q);
            }
        }
    }
    const float2 r = block_sum2(a, q, scratch, tid);
    if (tid == 0) {
        part[((size_t)c * S + s) * 2 + 0] = r.x;
        part[((size_t)c * S + s) * 2 + 1] = r.y;
    }
}

// ---- forward pass 2: finalise statistics (train) or take running ones (eval), then apply -----------
template <typename T, bool VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const T *__restrict__ x, int64_t M, int64_t Mg,
                                                              int64_t chunk, int Sg, int G,
                                                              const float *__restrict__ pre_bias,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta,
                                                              const T *__restrict__ residual, int act, float slope,
                                                              float eps, float momentum, int training,
                                                              float *__restrict__ running_mean,
                                                              float *__restrict__ running_var,
                                                              const float *__restrict__ part, T *__restrict__ out,
                                                              float *__restrict__ save_mean,
                                                              float *__restrict__ save_invstd) {
    const int c = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int S = gridDim.x, grp = s / Sg, sl = s - grp * Sg;
    const T *row = x + (size_t)c * M;
    const float pb = pre_bias ? pre_bias[c] : 0.0f;
    float mean, invstd;
    if (training) {
        // statistics of group `grp` (a view's columns); same summation order in every block of the group
        float a = 0.0f, q = 0.0f;
        for (int i = grp * Sg; i < (grp + 1) * Sg; ++i) {
            a += part[((size_t)c * S + i) * 2 + 0];
            q += part[((size_t)c * S + i) * 2 + 1];
        }
        const float shift = BnIO<T>::ld1(row + (int64_t)grp * Mg) + pb;
        const float dm = a / (float)Mg;
        const float var = fmaxf(q / (float)Mg - dm * dm, 0.0f);
        mean = shift + dm;
        invstd = 1.0f / sqrtf(var + eps);
        if (sl == 0 && tid == 0) {
            save_mean[c * G + grp] = mean;
            save_invstd[c * G + grp] = invstd;
        }
        if (s == 0 && tid == 0 && running_mean) {
            // running statistics advance once per group, in order -- exactly what G sequential forward calls do
            float rm = running_mean[c], rv = running_var[c];
            for (int g2 = 0; g2 < G; ++g2) {
                float a2 = 0.0f, q2 = 0.0f;
                for (int i = g2 * Sg; i < (g2 + 1) * Sg; ++i) {
                    a2 += part[((size_t)c * S + i) * 2 + 0];
                    q2 += part[((size_t)c * S + i) * 2 + 1];
                }
                const float sh2 = BnIO<T>::ld1(row + (int64_t)g2 * Mg) + pb;
                const float dm2 = a2 / (float)Mg;
                const float var2 = fmaxf(q2 / (float)Mg - dm2 * dm2, 0.0f);
                const float unbiased = Mg > 1 ? var2 * ((float)Mg / (float)(Mg - 1)) : var2;
                rm = (1.0f - momentum) * rm + momentum * (sh2 + dm2);
                rv = (1.0f - momentum) * rv + momentum * unbiased;
            }
            running_mean[c] = rm;
            running_var[c] = rv;
        }
    } else {
        mean = running_mean[c];
        invstd = 1.0f / sqrtf(running_var[c] + eps);
        if (sl == 0 && tid == 0) {
            save_mean[c * G + grp] = mean;
            save_invstd[c * G + grp] = invstd;
        }
    }
    const float g = gamma[c] * invstd;
    const float off = beta[c] + (pb - mean) * g;      // z = act(x*g + off) + r
    const T *rrow = residual ? residual + (size_t)c * M : nullptr;
    T *orow = out + (size_t)c * M;
    const int64_t gend = (int64_t)(grp + 1) * Mg;
    const int64_t lo = (int64_t)grp * Mg + (int64_t)sl * chunk, hi = (lo + chunk < gend) ? lo + chunk : gend;
    constexpr int W = VEC ? BnIO<T>::W : 1;
    int64_t m = lo + (int64_t)tid * W;
    if (VEC) {
        // four vectors in flight per thread (one load -> use -> store per iteration left the kernel latency-bound at
        // ~4.7 TB/s; this is the eval-mode / fingerprinting path)
        constexpr int U = 4;
        const int64_t step = (int64_t)BN_THREADS * W;
        for (; m + (U - 1) * step + W <= hi; m += U * step) {
            typename BnIO<T>::Raw rx[U], rr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                rx[u] = *reinterpret_cast<const typename BnIO<T>::Raw *>(row + m + u * step);
                if (rrow) rr[u] = *reinterpret_cast<const typename BnIO<T>::Raw *>(rrow + m + u * step);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float v[BnIO<T>::W], r[BnIO<T>::W];
                BnIO<T>::unpack(rx[u], v);
                if (rrow) BnIO<T>::unpack(rr[u], r);
#pragma unroll
                for (int i = 0; i < BnIO<T>::W; ++i) {
                    v[i] = act_fwd(__builtin_fmaf(v[i], g, off), act, slope);
                    if (rrow) v[i] += r[i];
                }
                BnIO<T>::store(orow + m + u * step, v);
            }
        }
    }
    for (; m < hi; m += (int64_t)BN_THREADS * W) {
        if (VEC && m + W <= hi) {
            float v[BnIO<T>::W], r[BnIO<T>::W];
            BnIO<T>::load(row + m, v);
            if (rrow) BnIO<T>::load(rrow + m, r);
#pragma unroll
            for (int i = 0; i < BnIO<T>::W; ++i) {
                v[i] = act_fwd(__builtin_fmaf(v[i], g, off), act, slope);
                if (rrow) v[i] += r[i];
            }
            BnIO<T>::store(orow + m, v);
        } else {
            for (int i = 0; i < W && m + i < hi; ++i) {
                float z = act_fwd(__builtin_fmaf(BnIO<T>::ld1(row + m + i), g, off), act, slope);
                if (rrow) z += BnIO<T>::ld1(rrow + m + i);
                BnIO<T>::st1(orow + m + i, z);
            }
        }
    }
}

// ---- backward pass 1: partial sums of dy and dy * xhat ----------------------------------------------
template <typename T, bool VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_reduce_kernel(const T *__restrict__ x, const T *__restrict__ dz,
                                                                   int64_t M, int64_t Mg, int64_t chunk, int Sg,
                                                                   int G, const float *__restrict__ pre_bias,
                                                                   const float *__restrict__ gamma,
                                                                   const float *__restrict__ beta,
                                                                   const float *__restrict__ save_mean,
                                                                   const float *__restrict__ save_invstd, int act,
                                                                   float slope, float *__restrict__ part) {
    __shared__ float2 scratch[BN_THREADS / 64];
    const int c = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int S = gridDim.x, grp = s / Sg, sl = s - grp * Sg;
    const T *row = x + (size_t)c * M, *grow = dz + (size_t)c * M;
    const float pb = pre_bias ? pre_bias[c] : 0.0f;
    const float mean = save_mean[c * G + grp], invstd = save_invstd[c * G + grp], ga = gamma[c], be = beta[c];
    const int64_t gend = (int64_t)(grp + 1) * Mg;
    const int64_t lo = (int64_t)grp * Mg + (int64_t)sl * chunk, hi = (lo + chunk < gend) ? lo + chunk : gend;
    float sd = 0.0f, sdx = 0.0f;
    constexpr int W = VEC ? BnIO<T>::W : 1;
    for (int64_t m = lo + (int64_t)tid * W; m < hi; m += (int64_t)BN_THREADS * W) {
        if (VEC && m + W <= hi) {
            float v[BnIO<T>::W], d[BnIO<T>::W];
            BnIO<T>::load(row + m, v);
            BnIO<T>::load(grow + m, d);
#pragma unroll
            for (int i = 0; i < BnIO<T>::W; ++i) {
                const float xh = ((v[i] + pb) - mean) * invstd;
                const float dy = d[i] * act_grad(__builtin_fmaf(xh, ga, be), act, slope);
                sd += dy;
                sdx = __builtin_fmaf(dy, xh, sdx);
            }
        } else {
            for (int i = 0; i < W && m + i < hi; ++i) {
                const float xh = ((BnIO<T>::ld1(row + m + i) + pb) - mean) * invstd;
                const float dy = BnIO<T>::ld1(grow + m + i) * act_grad(__builtin_fmaf(xh, ga, be), act, slope);
                sd += dy;
                sdx = __builtin_fmaf(dy, xh, sdx);
            }
        }
    }
    const float2 r = block_sum2(sd, sdx, scratch, tid);
    if (tid == 0) {
        part[((size_t)c * S + s) * 2 + 0] = r.x;
        part[((size_t)c * S + s) * 2 + 1] = r.y;
    }
}

// ---- backward pass 2: dgamma, dbeta, dx ---------------------------------------------------------------
template <typename T, bool VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_dx_kernel(const T *__restrict__ x, const T *__restrict__ dz,
                                                               int64_t M, int64_t Mg, int64_t chunk, int Sg, int G,
                                                               const float *__restrict__ pre_bias,
                                                               const float *__restrict__ gamma,
                                                               const float *__restrict__ beta,
                                                               const float *__restrict__ save_mean,
                                                               const float *__restrict__ save_invstd, int act,
                                                               float slope, int training,
                                                               const float *__restrict__ part, T *__restrict__ dx,
                                                               float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                               float *__restrict__ dpre_bias) {
    const int c = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int S = gridDim.x, grp = s / Sg, sl = s - grp * Sg;
    const T *row = x + (size_t)c * M, *grow = dz + (size_t)c * M;
    T *orow = dx + (size_t)c * M;
    const float pb = pre_bias ? pre_bias[c] : 0.0f;
    const float mean = save_mean[c * G + grp], invstd = save_invstd[c * G + grp], ga = gamma[c], be = beta[c];
    float sd = 0.0f, sdx = 0.0f;
    for (int i = grp * Sg; i < (grp + 1) * Sg; ++i) {
        sd += part[((size_t)c * S + i) * 2 + 0];
        sdx += part[((size_t)c * S + i) * 2 + 1];
    }
    if (s == 0 && tid == 0) {          // parameter gradients sum over all groups (fixed order)
        float td = 0.0f, tdx = 0.0f;
        for (int i = 0; i < S; ++i) {
            td += part[((size_t)c * S + i) * 2 + 0];
            tdx += part[((size_t)c * S + i) * 2 + 1];
        }
        dgamma[c] = tdx;
        dbeta[c] = td;
        // gradient of a bias added BEFORE the normalisation: cancels exactly under batch statistics; with running
        // statistics (eval) dx = ga*invstd*dy, so it is ga*invstd*sum(dy)
        if (dpre_bias) dpre_bias[c] = training ? 0.0f : ga * invstd * td;
    }
    // train: dx = ga*invstd * (dy - mean_g(dy) - xhat * mean_g(dy*xhat)) within the group; eval: dx = ga*invstd*dy
    const float k = ga * invstd;
    const float m1 = training ? sd / (float)Mg : 0.0f, m2 = training ? sdx / (float)Mg : 0.0f;
    const int64_t gend = (int64_t)(grp + 1) * Mg;
    const int64_t lo = (int64_t)grp * Mg + (int64_t)sl * chunk, hi = (lo + chunk < gend) ? lo + chunk : gend;
    constexpr int W = VEC ? BnIO<T>::W : 1;
    for (int64_t m = lo + (int64_t)tid * W; m < hi; m += (int64_t)BN_THREADS * W) {
        if (VEC && m + W <= hi) {
            float v[BnIO<T>::W], d[BnIO<T>::W];
            BnIO<T>::load(row + m, v);
            BnIO<T>::load(grow + m, d);
#pragma unroll
            for (int i = 0; i < BnIO<T>::W; ++i) {
                const float xh = ((v[i] + pb) - mean) * invstd;
                const float dy = d[i] * act_grad(__builtin_fmaf(xh, ga, be), act, slope);
                v[i] = k * ((dy - m1) - xh * m2);
            }
            BnIO<T>::store(orow + m, v);
        } else {
            for (int i = 0; i < W && m + i < hi; ++i) {
                const float xh = ((BnIO<T>::ld1(row + m + i) + pb) - mean) * invstd;
                const float dy = BnIO<T>::ld1(grow + m + i) * act_grad(__builtin_fmaf(xh, ga, be), act, slope);
                BnIO<T>::st1(orow + m + i, k * ((dy - m1) - xh * m2));
            }
        }
    }
}

// ---- single-pass variants: the chunk stays in registers across a row-wide rendezvous --------------------------
// The two-kernel forms above read X twice (forward) or X and dZ twice (backward).  Here a workgroup keeps its chunk
// (8 / 4 16-byte vectors per thread and operand, bf16 left packed) in VGPRs, publishes its partial sums, waits until
// all S workgroups of ITS ROW have published (the S workgroups of a row have consecutive block ids and are dispatched together -- the
// forward-progress assumption of a decoupled look-back scan), reduces the partials in the fixed order the two-kernel
// form uses, and finishes from registers: forward 3 -> 2 passes over HBM, backward 5 -> 3.
// `sync` = one counter line per row followed by S 8-byte slots per row, ALL ONES on entry and again on exit.
// Cross-XCD visibility: slots and counters move with agent-scope relaxed atomics, i.e. sc1 write-through stores and
// L2-bypassing loads (the per-XCD L2s are not coherent for plain accesses); no L2 writeback/invalidate.  The wait
// is bounded: a workgroup whose row-mates do not show up recomputes their partial sums itself (no trap, no hang).
constexpr int BN1_ITEMS_FWD = 8;                       // 16-byte vectors per thread, kept RAW (bf16 stays packed)
constexpr int BN1_ITEMS_BWD = 4;                       // per operand (x and dz); measured: fwd 8 / bwd 4 beat 4/4 and 8/8
constexpr int BN1_THREADS = 256;
constexpr int BN1_MIN_CHUNK = BN1_THREADS * BN1_ITEMS_BWD * 4;   // smallest chunk (f32 backward): bounds the slots per row
constexpr int BN1_MAX_S = 256;                         // workgroups per row
constexpr int BN1_SYNC_STRIDE = 64;                    // ints between row counters: one 256-byte line each, so the
                                                       // polls and arrivals of different rows never share a channel queue

constexpr unsigned BN1_EMPTY = 0xffffffffu;            // "not published yet" (a NaN pattern real sums are steered away from)

// Publishes this workgroup's partial pair into slot s of its row and waits until slots [w_lo, w_hi) of the row are filled:
// a workgroup needs the partials of ITS group (view) only -- the statistics of the views are independent -- and just
// chunk 0, which also writes the per-row results over all views, needs every slot.  With two views the wait is for the
// slowest of half as many workgroups (the rate of these kernels falls with the chunks per row: 5.2 TB/s at 16, 3.7 at 128).
// No read-modify-write sits on the critical path: a slot is ONE 8-byte write-through store, the wait is wave 0 polling
// the row's S slots with L2-bypassing loads (the successful poll already holds the data).  Returns with sp[0..S) set.
// The wait is BOUNDED and never traps: after `spin_limit` polls the slots that are still empty are left marked in sp
// (BN1_EMPTY in .x) and the caller recomputes exactly those partials from global memory itself (same thread mapping,
// same summation order => the same bits), so a workgroup never depends on row-mates that are not resident -- other
// kernels holding CUs (RCCL all-reduces overlapping backward, several ranks on one device, CU masks) cost time, not
// correctness.  Returns the number of slots the caller has to fill in (workgroup-uniform).
__device__ __forceinline__ int bn1_publish_and_wait(float a, float b, unsigned long long *slots_row, int s, int w_lo,
                                                    int w_hi, float2 *sp, int *n_missing, int spin_limit, int tid) {
    if (tid == 0) {
        unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
        if (ua == BN1_EMPTY) ua = 0xfffffffeu;          // still a NaN, but not the marker
        __hip_atomic_store(slots_row + s, (unsigned long long)ua | ((unsigned long long)ub << 32), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        *n_missing = 0;
    }
    if (tid < 64) {
        int spins = 0;
        for (;;) {
            int missing = 0;
            for (int i = w_lo + tid; i < w_hi; i += 64) {
                const unsigned long long v = __hip_atomic_load(slots_row + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)v == BN1_EMPTY) ++missing;
                sp[i] = make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
            }
            if (__all(missing == 0)) break;
            if (++spins > spin_limit) {
                if (missing) atomicAdd(n_missing, missing);
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    return *n_missing;
}

// After a workgroup has its copy of the partials it checks out of the row (fire and forget: the returned count is
// only looked at when the workgroup is done); the last one out empties the slots and re-arms the counter, so the
// whole `sync` buffer is all-ones again when the kernel ends.
__device__ __forceinline__ int bn1_checkout(int *counter) {
    return __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void bn1_rearm(int old, int S, int *counter, unsigned long long *slots_row) {
    if (old == S - 2) {                                  // counter starts at -1: the S-th checkout sees S - 2
        for (int i = 0; i < S; ++i)
            __hip_atomic_store(slots_row + i, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(counter, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The packed operands are kept across the rendezvous and unpacked AGAIN afterwards.  Without this fence the compiler
// keeps the unpacked / derived floats of the first phase alive instead (common subexpressions): 118 VGPRs for 4 + 4
// vectors, 188 for 8 + 8 -- two to four workgroups per CU, and every one of them that waits for its row-mates is a
// slot that moves no data.  The empty asm makes the registers opaque at no cost.
__device__ __forceinline__ void bn_opaque(float4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void bn_opaque(uint4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

template <typename T, bool RES>
__global__ __launch_bounds__(BN1_THREADS) void bn_fwd1_kernel(const T *__restrict__ x, int64_t M, int64_t Mg, int Sg,
                                                             int G, const float *__restrict__ pre_bias,
                                                             const float *__restrict__ gamma,
                                                             const float *__restrict__ beta,
                                                             const T *__restrict__ residual, int act, float slope,
                                                             float eps, float momentum,
                                                             float *__restrict__ running_mean,
                                                             float *__restrict__ running_var,
                                                             int *__restrict__ sync, T *__restrict__ out,
                                                             float *__restrict__ save_mean,
                                                             float *__restrict__ save_invstd, int spin_limit) {
    constexpr int W = BnIO<T>::W, ITEMS = BN1_ITEMS_FWD, CHUNK = BN1_THREADS * ITEMS * W;
    __shared__ float2 scratch[BN1_THREADS / 64];
    __shared__ float2 sp[BN1_MAX_S];
    __shared__ int n_missing;
    const int c = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int S = gridDim.x, grp = s / Sg, sl = s - grp * Sg;
    const T *row = x + (size_t)c * M;
    const float pb = pre_bias ? pre_bias[c] : 0.0f;
    const float shift = BnIO<T>::ld1(row + (int64_t)grp * Mg) + pb;
    const int64_t gend = (int64_t)(grp + 1) * Mg;
    const int64_t lo = (int64_t)grp * Mg + (int64_t)sl * CHUNK;
    const int64_t hi = (lo + CHUNK < gend) ? lo + CHUNK : gend;
    typename BnIO<T>::Raw raw[ITEMS];
    float a = 0.0f, q = 0.0f;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int64_t m = lo + ((int64_t)it * BN1_THREADS + tid) * W;
        if (m < hi) raw[it] = *reinterpret_cast<const typename BnIO<T>::Raw *>(row + m);
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int64_t m = lo + ((int64_t)it * BN1_THREADS + tid) * W;
        if (m < hi) {
            float v[W];
            BnIO<T>::unpack(raw[it], v);
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const float d = (v[i] + pb) - shift;
                a += d;
                q = __builtin_fmaf(d, d, q);
            }
        }
    }
    // the shortcut rows do not depend on the statistics: fetch them now, so their latency passes during the rendezvous
    // (RES is a template parameter: the 32 extra registers only exist in the instantiation that needs them)
    const T *rrow = RES ? residual + (size_t)c * M : nullptr;
    typename BnIO<T>::Raw rres[RES ? ITEMS : 1];
    if (RES) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int64_t m = lo + ((int64_t)it * BN1_THREADS + tid) * W;
            if (m < hi) rres[it] = *reinterpret_cast<const typename BnIO<T>::Raw *>(rrow + m);
        }
    }
    const float2 r = block_sum2<BN1_THREADS>(a, q, scratch, tid);
    unsigned long long *slots_row = reinterpret_cast<unsigned long long *>(sync + (size_t)gridDim.y * BN1_SYNC_STRIDE) + (size_t)c * S;
    int *counter = sync + (size_t)c * BN1_SYNC_STRIDE;
    const int w_lo = s == 0 ? 0 : grp * Sg, w_hi = s == 0 ? S : (grp + 1) * Sg;     // chunk 0 also writes the running stats
    if (bn1_publish_and_wait(r.x, r.y, slots_row, s, w_lo, w_hi, sp, &n_missing, spin_limit, tid) > 0) {
        // row-mates that did not show up in time: their partial sums straight from the row (identical order and bits)
        for (int i = w_lo; i < w_hi; ++i) {
            if (__float_as_uint(sp[i].x) != BN1_EMPTY) continue;            // LDS value: workgroup-uniform branch
            const int g2 = i / Sg;
            const float sh2 = BnIO<T>::ld1(row + (int64_t)g2 * Mg) + pb;
            const int64_t lo2 = (int64_t)g2 * Mg + (int64_t)(i - g2 * Sg) * CHUNK;
            const int64_t hi2 = (lo2 + CHUNK < (int64_t)(g2 + 1) * Mg) ? lo2 + CHUNK : (int64_t)(g2 + 1) * Mg;
            float a2 = 0.0f, q2 = 0.0f;
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int64_t m = lo2 + ((int64_t)it * BN1_THREADS + tid) * W;
                if (m < hi2) {
                    float v[W];
                    BnIO<T>::load(row + m, v);
#pragma unroll
                    for (int e = 0; e < W; ++e) {
                        const float d = (v[e] + pb) - sh2;
                        a2 += d;
                        q2 = __builtin_fmaf(d, d, q2);
                    }
                }
            }
            const float2 r2 = block_sum2<BN1_THREADS>(a2, q2, scratch, tid);
            if (tid == 0) sp[i] = r2;
        }
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) bn_opaque(raw[it]);
    int checkout = 0;
    if (tid == 0) checkout = bn1_checkout(counter);
    a = 0.0f, q = 0.0f;
    for (int i = grp * Sg; i < (grp + 1) * Sg; ++i) {
        a += sp[i].x;
        q += sp[i].y;
    }
    const float dm = a / (float)Mg;
    const float var = fmaxf(q / (float)Mg - dm * dm, 0.0f);
    const float mean = shift + dm;
    const float invstd = 1.0f / sqrtf(var + eps);
    if (sl == 0 && tid == 0) {
        save_mean[c * G + grp] = mean;
        save_invstd[c * G + grp] = invstd;
    }
    if (s == 0 && tid == 0 && running_mean) {
        float rm = running_mean[c], rv = running_var[c];
        for (int g2 = 0; g2 < G; ++g2) {
            float a2 = 0.0f, q2 = 0.0f;
            for (int i = g2 * Sg; i < (g2 + 1) * Sg; ++i) {
                a2 += sp[i].x;
                q2 += sp[i].y;
            }
            const float sh2 = BnIO<T>::ld1(row + (int64_t)g2 * Mg) + pb;
            const float dm2 = a2 / (float)Mg;
            const float var2 = fmaxf(q2 / (float)Mg - dm2 * dm2, 0.0f);
            const float unbiased = Mg > 1 ? var2 * ((float)Mg / (float)(Mg - 1)) : var2;
            rm = (1.0f - momentum) * rm + momentum * (sh2 + dm2);
            rv = (1.0f - momentum) * rv + momentum * unbiased;
        }
        running_mean[c] = rm;
        running_var[c] = rv;
    }
    const float g = gamma[c] * invstd;
    const float off = beta[c] + (pb - mean) * g;
    T *orow = out + (size_t)c * M;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int64_t m = lo + ((int64_t)it * BN1_THREADS + tid) * W;
        if (m < hi) {
            float v[W], rr[W];
            BnIO<T>::unpack(raw[it], v);
            if (RES) BnIO<T>::unpack(rres[it], rr);
#pragma unroll
            for (int i = 0; i < W; ++i) {
                v[i] = act_fwd(__builtin_fmaf(v[i], g, off), act, slope);
                if (RES) v[i] += rr[i];
            }
            BnIO<T>::store(orow + m, v);
        }
    }
    if (tid == 0) bn1_rearm(checkout, S, counter, slots_row);
}

template <typename T, int ITEMS, int THREADS = BN1_THREADS>
__global__ __launch_bounds__(THREADS) void bn_bwd1_kernel(const T *__restrict__ x, const T *__restrict__ dz,
                                                             int64_t M, int64_t Mg, int Sg, int G,
                                                             const float *__restrict__ pre_bias,
                                                             const float *__restrict__ gamma,
                                                             const float *__restrict__ beta,
                                                             const float *__restrict__ save_mean,
                                                             const float *__restrict__ save_invstd, int act,
                                                             float slope,
                                                             int *__restrict__ sync, T *__restrict__ dx,
                                                             float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                             float *__restrict__ dpre_bias, int spin_limit,
                                                             int plain_stores) {
    constexpr int W = BnIO<T>::W, CHUNK = THREADS * ITEMS * W;
    __shared__ float2 scratch[THREADS / 64];
    __shared__ float2 sp[BN1_MAX_S];
    __shared__ int n_missing;
    const int c = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const int S = gridDim.x, grp = s / Sg, sl = s - grp * Sg;
    const T *row = x + (size_t)c * M, *grow = dz + (size_t)c * M;
    T *orow = dx + (size_t)c * M;
    const float pb = pre_bias ? pre_bias[c] : 0.0f;
    const float mean = save_mean[c * G + grp], invstd = save_invstd[c * G + grp], ga = gamma[c], be = beta[c];
    // The kernel is partly bound by its vector instructions (27 per element against ~50 lane-operations per element that
    // 5 TB/s leave a CU), so the per-element arithmetic is folded into the fewest fused operations:
    //   xhat = fma(x, invstd, (pb - mean) invstd);  pre-activation = fma(x, gamma invstd, beta + (pb - mean) gamma invstd)
    //   -- the very expression the forward pass thresholds, so the ReLU decision is the forward one by construction;
    //   masked gradient = pre > 0 ? dz : dz * neg   (neg: 1 without activation, 0 for ReLU, the slope for LeakyReLU)
    const float xh0 = (pb - mean) * invstd, zg = ga * invstd, zoff = be + (pb - mean) * zg;
    const float neg = act == 0 ? 1.0f : (act == 1 ? 0.0f : slope);
    const int64_t gend = (int64_t)(grp + 1) * Mg;
    const int64_t lo = (int64_t)grp * Mg + (int64_t)sl * CHUNK;
    const int64_t hi = (lo + CHUNK < gend) ? lo + CHUNK : gend;
    typename BnIO<T>::Raw rx[ITEMS], rd[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int64_t m = lo + ((int64_t)it * THREADS + tid) * W;
        if (m < hi) {
            rx[it] = GRAFP_LD_ONCE(1, reinterpret_cast<const typename BnIO<T>::Raw *>(row + m));
            rd[it] = GRAFP_LD_ONCE(4, reinterpret_cast<const typename BnIO<T>::Raw *>(grow + m));
        }
    }
    float sd = 0.0f, sdx = 0.0f;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int64_t m = lo + ((int64_t)it * THREADS + tid) * W;
        if (m < hi) {
            float v[W], d[W];
            BnIO<T>::unpack(rx[it], v);
            BnIO<T>::unpack(rd[it], d);
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const float xh = __builtin_fmaf(v[i], invstd, xh0);
                const float dy = __builtin_fmaf(v[i], zg, zoff) > 0.0f ? d[i] : d[i] * neg;
                sd += dy;
                sdx = __builtin_fmaf(dy, xh, sdx);
            }
        }
    }
    const float2 r = block_sum2<THREADS>(sd, sdx, scratch, tid);
    unsigned long long *slots_row = reinterpret_cast<unsigned long long *>(sync + (size_t)gridDim.y * BN1_SYNC_STRIDE) + (size_t)c * S;
    int *counter = sync + (size_t)c * BN1_SYNC_STRIDE;
    const int w_lo = s == 0 ? 0 : grp * Sg, w_hi = s == 0 ? S : (grp + 1) * Sg;     // chunk 0 also writes dgamma / dbeta
    if (bn1_publish_and_wait(r.x, r.y, slots_row, s, w_lo, w_hi, sp, &n_missing, spin_limit, tid) > 0) {
        for (int i = w_lo; i < w_hi; ++i) {
            if (__float_as_uint(sp[i].x) != BN1_EMPTY) continue;
            const int g2 = i / Sg;
            const float mean2 = save_mean[c * G + g2], invstd2 = save_invstd[c * G + g2];
            const int64_t lo2 = (int64_t)g2 * Mg + (int64_t)(i - g2 * Sg) * CHUNK;
            const int64_t hi2 = (lo2 + CHUNK < (int64_t)(g2 + 1) * Mg) ? lo2 + CHUNK : (int64_t)(g2 + 1) * Mg;
            float sd2 = 0.0f, sdx2 = 0.0f;
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int64_t m = lo2 + ((int64_t)it * THREADS + tid) * W;
                if (m < hi2) {
                    float v[W], d[W];
                    BnIO<T>::load(row + m, v);
                    BnIO<T>::load(grow + m, d);
#pragma unroll
                    for (int e = 0; e < W; ++e) {
                        const float xh = __builtin_fmaf(v[e], invstd2, (pb - mean2) * invstd2);
                        const float dy = __builtin_fmaf(v[e], ga * invstd2, be + (pb - mean2) * (ga * invstd2)) > 0.0f ? d[e] : d[e] * neg;
                        sd2 += dy;
                        sdx2 = __builtin_fmaf(dy, xh, sdx2);
                    }
                }
            }
            const float2 r2 = block_sum2<THREADS>(sd2, sdx2, scratch, tid);
            if (tid == 0) sp[i] = r2;
        }
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        bn_opaque(rx[it]);
        bn_opaque(rd[it]);
    }
    int checkout = 0;
    if (tid == 0) checkout = bn1_checkout(counter);
    sd = 0.0f, sdx = 0.0f;
    for (int i = grp * Sg; i < (grp + 1) * Sg; ++i) {
        sd += sp[i].x;
        sdx += sp[i].y;
    }
    if (s == 0 && tid == 0) {
        float td = 0.0f, tdx = 0.0f;
        for (int i = 0; i < S; ++i) {
            td += sp[i].x;
            tdx += sp[i].y;
        }
        dgamma[c] = tdx;
        dbeta[c] = td;
        if (dpre_bias) dpre_bias[c] = 0.0f;          // training mode only: cancels in the normalisation
    }
    const float k = ga * invstd;
    const float m1 = sd / (float)Mg, m2 = sdx / (float)Mg;
    const float km1 = -(k * m1), km2 = -(k * m2);               // dx = k dy - k m1 - k m2 xh: two fmaf per element
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int64_t m = lo + ((int64_t)it * THREADS + tid) * W;
        if (m < hi) {
            float v[W], d[W];
            BnIO<T>::unpack(rx[it], v);
            BnIO<T>::unpack(rd[it], d);
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const float xh = __builtin_fmaf(v[i], invstd, xh0);
                const float dy = __builtin_fmaf(v[i], zg, zoff) > 0.0f ? d[i] : d[i] * neg;
                v[i] = __builtin_fmaf(xh, km2, __builtin_fmaf(dy, k, km1));
            }
            BnIO<T>::store(orow + m, v, plain_stores != 0);
        }
    }
    if (tid == 0) bn1_rearm(checkout, S, counter, slots_row);
}

// single-pass plan: Sg chunks per group, or 0 when the shape does not qualify
static int bn1_plan(int64_t Mg, int G, int W, int items, int threads = BN1_THREADS) {
    if (Mg % W) return 0;
    const int64_t chunk = (int64_t)threads * items * W;
    const int64_t Sg = (Mg + chunk - 1) / chunk;
    if (Sg * G > BN1_MAX_S) return 0;
    return (int)Sg;
}

struct BnPlan {
    int Sg;
    int64_t chunk;
};
static BnPlan bn_plan(int C, int64_t Mg, int G, int W) {
    const int64_t per_block = (int64_t)BN_THREADS * W * 4;      // >= 4 vector iterations per thread
    int64_t S = (Mg + per_block - 1) / per_block;
    const int64_t cap = 4096 / ((int64_t)C * G) > 1 ? 4096 / ((int64_t)C * G) : 1;   // ~4096 workgroups overall
    if (S > cap) S = cap;
    if (S < 1) S = 1;
    int64_t chunk = (Mg + S - 1) / S;
    chunk = (chunk + W - 1) / W * W;                             // chunk boundaries stay vector-aligned
    BnPlan p;
    p.chunk = chunk;
    p.Sg = (int)((Mg + chunk - 1) / chunk);
    return p;
}

template <typename T>
static bool bn_vec_ok(const void *a, const void *b, const void *c, const void *d, int64_t Mg) {
    const uintptr_t m = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d;
    return (Mg % BnIO<T>::W) == 0 && (m & 15) == 0;
}

}  // namespace grafp

// polls (~1 us each) before a workgroup stops waiting for its row-mates and recomputes their partial sums itself; a
// per-call argument of the *_1pass entry points (negative = this default; 0 = never wait)
static inline int bn_spin(int spin_limit) { return spin_limit < 0 ? (1 << 12) : spin_limit; }

extern "C" size_t grafp_bn_workspace(int C, int64_t M) {
    if (C <= 0 || M <= 0) return 0;
    const size_t two_pass = (size_t)4096 + (size_t)C * 8;          // >= C * G * Sg partial pairs for any G <= 8
    return two_pass * 2 * sizeof(float);
}

extern "C" size_t grafp_bn_sync_bytes(int C, int64_t M) {
    if (C <= 0 || M <= 0) return 0;
    // a counter line per row + at most (M / chunk + groups) <= M / chunk + 8 slots per row
    return (size_t)C * grafp::BN1_SYNC_STRIDE * sizeof(int) + (size_t)C * ((size_t)(M / grafp::BN1_MIN_CHUNK) + 8) * 8;
}

extern "C" int grafp_bn_fwd_1pass(const void *x, int dtype, int C, int64_t M, int groups, const float *pre_bias,
                                  const float *gamma, const float *beta, const void *residual, int act, float slope,
                                  float eps, float momentum, int training, float *running_mean, float *running_var,
                                  void *out, float *save_mean, float *save_invstd, void *ws, size_t ws_bytes,
                                  int32_t *sync, int spin_limit, grafp_stream_t stream) {
    using namespace grafp;
    const int spin = bn_spin(spin_limit);
    GRAFP_REQUIRE(x && gamma && beta && out && save_mean && save_invstd, "bn_fwd: null pointer");
    GRAFP_REQUIRE(C > 0 && M > 0 && C <= 65535, "bn_fwd: bad shape C=%d M=%lld", C, (long long)M);
    GRAFP_REQUIRE(groups >= 1 && groups <= 8 && M % groups == 0, "bn_fwd: groups=%d must be in [1,8] and divide M=%lld", groups, (long long)M);
    const int G = groups;
    const int64_t Mg = M / G;
    GRAFP_REQUIRE(dtype == GRAFP_F32 || dtype == GRAFP_BF16, "bn_fwd: dtype %d not in {f32, bf16}", dtype);
    GRAFP_REQUIRE(act >= 0 && act <= 2, "bn_fwd: act %d not in {0 none, 1 relu, 2 leaky}", act);
    GRAFP_REQUIRE(training || (running_mean && running_var), "bn_fwd: eval mode needs running statistics");
    if (!ws || ws_bytes < grafp_bn_workspace(C, M)) {
        set_error("bn_fwd: workspace %zu bytes < required %zu", ws_bytes, grafp_bn_workspace(C, M));
        return GRAFP_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    float *part = (float *)ws;
    if (sync && training) {
        const bool f32 = dtype == GRAFP_F32;
        const bool ok = f32 ? bn_vec_ok<float>(x, out, residual, nullptr, Mg) : bn_vec_ok<unsigned short>(x, out, residual, nullptr, Mg);
        const int Sg = ok ? bn1_plan(Mg, G, f32 ? 4 : 8, BN1_ITEMS_FWD) : 0;
        if (Sg > 0) {
            const dim3 grid(Sg * G, C);
#define BN_FWD1(T, RES)                                                                                                \
    hipLaunchKernelGGL((bn_fwd1_kernel<T, RES>), grid, dim3(BN1_THREADS), 0, s, (const T *)x, M, Mg, Sg, G, pre_bias,  \
                       gamma, beta, (const T *)residual, act, slope, eps, momentum, running_mean, running_var,         \
                       (int *)sync, (T *)out, save_mean, save_invstd, spin)
            if (f32) { if (residual) BN_FWD1(float, true); else BN_FWD1(float, false); }
            else { if (residual) BN_FWD1(unsigned short, true); else BN_FWD1(unsigned short, false); }
#undef BN_FWD1
            GRAFP_CHECK_LAUNCH("bn_fwd1_kernel");
            return GRAFP_OK;
        }
    }
#define BN_FWD(T, VEC)                                                                                                 \
    do {                                                                                                               \
        const BnPlan p = bn_plan(C, Mg, G, VEC ? BnIO<T>::W : 1);                                                      \
        const dim3 grid(p.Sg * G, C);                                                                                  \
        if (training)                                                                                                  \
            hipLaunchKernelGGL((bn_stats_kernel<T, VEC>), grid, dim3(BN_THREADS), 0, s, (const T *)x, M, Mg, p.chunk,   \
                               p.Sg, pre_bias, part);                                                                  \
        hipLaunchKernelGGL((bn_apply_kernel<T, VEC>), grid, dim3(BN_THREADS), 0, s, (const T *)x, M, Mg, p.chunk,      \
                           p.Sg, G, pre_bias, gamma, beta, (const T *)residual, act, slope, eps, momentum, training,            \
                           running_mean, running_var, part, (T *)out, save_mean, save_invstd);                         \
    } while (0)
    if (dtype == GRAFP_F32) {
        if (bn_vec_ok<float>(x, out, residual, nullptr, Mg)) BN_FWD(float, true); else BN_FWD(float, false);
    } else {
        if (bn_vec_ok<unsigned short>(x, out, residual, nullptr, Mg)) BN_FWD(unsigned short, true); else BN_FWD(unsigned short, false);
    }
#undef BN_FWD
    GRAFP_CHECK_LAUNCH("bn_stats_kernel / bn_apply_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_bn_fwd(const void *x, int dtype, int C, int64_t M, int groups, const float *pre_bias, const float *gamma,
                            const float *beta, const void *residual, int act, float slope, float eps, float momentum,
                            int training, float *running_mean, float *running_var, void *out, float *save_mean,
                            float *save_invstd, void *ws, size_t ws_bytes, grafp_stream_t stream) {
    return grafp_bn_fwd_1pass(x, dtype, C, M, groups, pre_bias, gamma, beta, residual, act, slope, eps, momentum, training,
                              running_mean, running_var, out, save_mean, save_invstd, ws, ws_bytes, nullptr, -1, stream);
}

extern "C" int grafp_bn_bwd_1pass(const void *x, const void *dz, int dtype, int C, int64_t M, int groups,
                                  const float *pre_bias, const float *gamma, const float *beta, const float *save_mean,
                                  const float *save_invstd, int act, float slope, int training, void *dx,
                                  float *dgamma, float *dbeta, float *dpre_bias, void *ws, size_t ws_bytes,
                                  int32_t *sync, int spin_limit, grafp_stream_t stream) {
    using namespace grafp;
    const int spin = bn_spin(spin_limit);
    GRAFP_REQUIRE(x && dz && gamma && beta && save_mean && save_invstd && dx && dgamma && dbeta, "bn_bwd: null pointer");
    GRAFP_REQUIRE(C > 0 && M > 0 && C <= 65535, "bn_bwd: bad shape C=%d M=%lld", C, (long long)M);
    GRAFP_REQUIRE(groups >= 1 && groups <= 8 && M % groups == 0, "bn_bwd: groups=%d must be in [1,8] and divide M=%lld", groups, (long long)M);
    const int G = groups;
    const int64_t Mg = M / G;
    GRAFP_REQUIRE(dtype == GRAFP_F32 || dtype == GRAFP_BF16, "bn_bwd: dtype %d not in {f32, bf16}", dtype);
    GRAFP_REQUIRE(act >= 0 && act <= 2, "bn_bwd: act %d not in {0 none, 1 relu, 2 leaky}", act);
    if (!ws || ws_bytes < grafp_bn_workspace(C, M)) {
        set_error("bn_bwd: workspace %zu bytes < required %zu", ws_bytes, grafp_bn_workspace(C, M));
        return GRAFP_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    float *part = (float *)ws;
    if (sync && training) {
        const bool f32 = dtype == GRAFP_F32;
        const bool ok = f32 ? bn_vec_ok<float>(x, dz, dx, nullptr, Mg) : bn_vec_ok<unsigned short>(x, dz, dx, nullptr, Mg);
        // from 16 chunks per row the rendezvous runs over fewer, larger chunks: 8 vectors per thread and operand instead
        // of 4 (tools/bn_bench.py, threshold swept 0 ... 128: 16 is best at 256, 512 and 2048 clip-views; with the
        // operands fenced across the wait -- bn_opaque -- this variant needs 122 VGPRs, 4 workgroups per CU)
        int items = BN1_ITEMS_BWD;
        int Sg = ok ? bn1_plan(Mg, G, f32 ? 4 : 8, items) : 0;
        const int plain = bn_plain_stores((size_t)C * (size_t)M * (f32 ? 4 : 2));
        const int items8_from = GRAFP_TUNE_INT("GRAFP_BN_BWD_ITEMS8_FROM", 16);
        if (ok && !f32 && (Sg == 0 || Sg * G > items8_from)) {
            items = 2 * BN1_ITEMS_BWD;
            Sg = bn1_plan(Mg, G, 8, items);
        }
        // ... and from 32 chunks per view on 512-thread workgroups: half the chunks again with the same registers per
        // thread and the same waves per CU (two workgroups instead of four).  tools/bn_bench.py at 2048 clip-views: 64
        // chunks per view (stage 0) 783 -> 703 us, 32 (stage 1) 681 -> 650; from 16 chunks (stage 2) it loses 1 %, and
        // 1024-thread workgroups -- ONE per CU, whose phases overlap nobody's -- lose 10-20 % everywhere.
        const int t512_from = GRAFP_TUNE_INT("GRAFP_BN_BWD_T512_FROM", 32);
        if (ok && !f32 && t512_from > 0 && items == 2 * BN1_ITEMS_BWD && Sg >= t512_from) {
            const int Sg2 = bn1_plan(Mg, G, 8, items, 512);
            if (Sg2 > 0) {
                const dim3 grid2(Sg2 * G, C);
                hipLaunchKernelGGL((bn_bwd1_kernel<unsigned short, 2 * BN1_ITEMS_BWD, 512>), grid2, dim3(512), 0, s,
                                   (const unsigned short *)x, (const unsigned short *)dz, M, Mg, Sg2, G, pre_bias, gamma,
                                   beta, save_mean, save_invstd, act, slope, (int *)sync, (unsigned short *)dx,
                                   dgamma, dbeta, dpre_bias, spin, plain);
                GRAFP_CHECK_LAUNCH("bn_bwd1_kernel");
                return GRAFP_OK;
            }
        }
        if (Sg > 0) {
            const dim3 grid(Sg * G, C);
            if (f32)
                hipLaunchKernelGGL((bn_bwd1_kernel<float, BN1_ITEMS_BWD>), grid, dim3(BN1_THREADS), 0, s, (const float *)x,
                                   (const float *)dz, M, Mg, Sg, G, pre_bias, gamma, beta, save_mean, save_invstd, act,
                                   slope, (int *)sync, (float *)dx, dgamma, dbeta, dpre_bias, spin, plain);
            else if (items == BN1_ITEMS_BWD)
                hipLaunchKernelGGL((bn_bwd1_kernel<unsigned short, BN1_ITEMS_BWD>), grid, dim3(BN1_THREADS), 0, s,
                                   (const unsigned short *)x, (const unsigned short *)dz, M, Mg, Sg, G, pre_bias, gamma,
                                   beta, save_mean, save_invstd, act, slope, (int *)sync, (unsigned short *)dx,
                                   dgamma, dbeta, dpre_bias, spin, plain);
            else
                hipLaunchKernelGGL((bn_bwd1_kernel<unsigned short, 2 * BN1_ITEMS_BWD>), grid, dim3(BN1_THREADS), 0, s,
                                   (const unsigned short *)x, (const unsigned short *)dz, M, Mg, Sg, G, pre_bias, gamma,
                                   beta, save_mean, save_invstd, act, slope, (int *)sync, (unsigned short *)dx,
                                   dgamma, dbeta, dpre_bias, spin, plain);
            GRAFP_CHECK_LAUNCH("bn_bwd1_kernel");
            return GRAFP_OK;
        }
    }
#define BN_BWD(T, VEC)                                                                                                 \
    do {                                                                                                               \
        const BnPlan p = bn_plan(C, Mg, G, VEC ? BnIO<T>::W : 1);                                                      \
        const dim3 grid(p.Sg * G, C);                                                                                  \
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, VEC>), grid, dim3(BN_THREADS), 0, s, (const T *)x, (const T *)dz,  \
                           M, Mg, p.chunk, p.Sg, G, pre_bias, gamma, beta, save_mean, save_invstd, act, slope, part);  \
        hipLaunchKernelGGL((bn_bwd_dx_kernel<T, VEC>), grid, dim3(BN_THREADS), 0, s, (const T *)x, (const T *)dz, M,   \
                           Mg, p.chunk, p.Sg, G, pre_bias, gamma, beta, save_mean, save_invstd, act, slope, training, part,    \
                           (T *)dx, dgamma, dbeta, dpre_bias);                                                         \
    } while (0)
    if (dtype == GRAFP_F32) {
        if (bn_vec_ok<float>(x, dz, dx, nullptr, Mg)) BN_BWD(float, true); else BN_BWD(float, false);
    } else {
        if (bn_vec_ok<unsigned short>(x, dz, dx, nullptr, Mg)) BN_BWD(unsigned short, true); else BN_BWD(unsigned short, false);
    }
#undef BN_BWD
    GRAFP_CHECK_LAUNCH("bn_bwd_reduce_kernel / bn_bwd_dx_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_bn_bwd(const void *x, const void *dz, int dtype, int C, int64_t M, int groups, const float *pre_bias,
                            const float *gamma, const float *beta, const float *save_mean, const float *save_invstd,
                            int act, float slope, int training, void *dx, float *dgamma, float *dbeta,
                            float *dpre_bias, void *ws, size_t ws_bytes, grafp_stream_t stream) {
    return grafp_bn_bwd_1pass(x, dz, dtype, C, M, groups, pre_bias, gamma, beta, save_mean, save_invstd, act, slope,
                              training, dx, dgamma, dbeta, dpre_bias, ws, ws_bytes, nullptr, -1, stream);
}
