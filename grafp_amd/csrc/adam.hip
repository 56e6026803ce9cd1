// adam.hip -- the optimizer update of a training step for ALL parameter tensors in a handful of launches, gfx950.
//
// Replaces optimizer.step() of /root/reference/train.py:79 (torch.optim.Adam with its defaults, train.py:174: betas
// (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) on the 271 parameter tensors of the model (18.4 M elements).
// torch's fused multi-tensor Adam needs 12 launches of ~26 us for them at ANY batch size (1.6 TB/s: 316 us, 2 % of a
// 128-pairs-per-GPU step); this is a plain streaming kernel: read p, g, m, v, write p, m, v = 28 bytes per element, four
// 16-byte vectors per array and thread in flight.
//
// Arithmetic (f32, one fixed order -- no contraction: the file is built with -ffp-contract=off; division and square
// root correctly rounded), with t = the step count AFTER this update (the caller bumps the counters first):
//   m  = m + (g - m) * (1 - beta1)                       (torch's lerp form)
//   v  = beta2 * v + (1 - beta2) * g * g
//   bc1 = 1 - beta1^t,  bc2s = sqrt(1 - beta2^t)         (double, as torch computes them)
//   p  = p - ((float)(lr / bc1) * m) / (sqrt(v) / (float)bc2s + eps)
// The table of tensors travels BY VALUE in the kernel arguments (64 entries per launch): no host-to-device copy, graph-
// capturable, no device-side state.
#include <math.h>

#include "common.h"

namespace grafp {

constexpr int AD_MULTI = 64;
constexpr int AD_VEC = 4;                         // 16-byte vectors per array and thread
constexpr int AD_BLOCK = 256 * AD_VEC * 4;        // elements per workgroup
struct AdamTable {
    float *p[AD_MULTI];
    const float *g[AD_MULTI];
    float *m[AD_MULTI];
    float *v[AD_MULTI];
    const float *step[AD_MULTI];
    int n[AD_MULTI], block_base[AD_MULTI + 1];
    int count;
    const float *lr_dev;                          // the learning rate as a 0-d device tensor (read at run time), or NULL
    double lr, beta1, beta2;
    float eps;
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float omb1, float b2, float omb2,
                                         float step_size, float bc2s, float eps) {
    m = m + (g - m) * omb1;
    v = b2 * v + omb2 * g * g;
    const float denom = __fsqrt_rn(v) / bc2s + eps;
    p = p - (step_size * m) / denom;
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamTable t) {
    __shared__ float s_hyper[2];
    int lo = 0, hi = t.count - 1;                                                  // uniform, <= 6 steps
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)blockIdx.x >= t.block_base[mid]) lo = mid;
        else hi = mid - 1;
    }
    const int e = lo, tid = threadIdx.x;
    const int64_t n = t.n[e];
    const int64_t base = (int64_t)(blockIdx.x - t.block_base[e]) * AD_BLOCK;
    float *__restrict__ P = t.p[e];
    const float *__restrict__ G = t.g[e];
    float *__restrict__ Mo = t.m[e];
    float *__restrict__ V = t.v[e];
    const bool vec = ((((uintptr_t)P | (uintptr_t)G | (uintptr_t)Mo | (uintptr_t)V) & 15) == 0) && base + AD_BLOCK <= n;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 rp[AD_VEC], rg[AD_VEC], rm[AD_VEC], rv[AD_VEC];
    if (vec) {                                    // the loads go out before the (double precision) bias corrections
#pragma unroll
        for (int u = 0; u < AD_VEC; ++u) {
            const int64_t i = base + ((int64_t)u * 256 + tid) * 4;
            rp[u] = *reinterpret_cast<const f4 *>(P + i);
            rg[u] = *reinterpret_cast<const f4 *>(G + i);
            rm[u] = *reinterpret_cast<const f4 *>(Mo + i);
            rv[u] = *reinterpret_cast<const f4 *>(V + i);
        }
    }
    if (tid == 0) {
        const double step = (double)*t.step[e];
        const double lr = t.lr_dev ? (double)*t.lr_dev : t.lr;
        const double bc1 = 1.0 - pow(t.beta1, step), bc2 = 1.0 - pow(t.beta2, step);
        s_hyper[0] = (float)(lr / bc1);
        s_hyper[1] = (float)sqrt(bc2);
    }
    __syncthreads();
    const float step_size = s_hyper[0], bc2s = s_hyper[1];
    const float omb1 = (float)(1.0 - t.beta1), b2 = (float)t.beta2, omb2 = (float)(1.0 - t.beta2), eps = t.eps;
    if (vec) {
#pragma unroll
        for (int u = 0; u < AD_VEC; ++u) {
            const int64_t i = base + ((int64_t)u * 256 + tid) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float p = rp[u][c], m = rm[u][c], v = rv[u][c];
                adam_one(p, rg[u][c], m, v, omb1, b2, omb2, step_size, bc2s, eps);
                rp[u][c] = p; rm[u][c] = m; rv[u][c] = v;
            }
            *reinterpret_cast<f4 *>(P + i) = rp[u];
            *reinterpret_cast<f4 *>(Mo + i) = rm[u];
            *reinterpret_cast<f4 *>(V + i) = rv[u];
        }
    } else {                                       // the last workgroup of an entry, or an unaligned view
        const int64_t end = base + AD_BLOCK < n ? base + AD_BLOCK : n;
        for (int64_t i = base + tid; i < end; i += 256) {
            float p = P[i], m = Mo[i], v = V[i];
            adam_one(p, G[i], m, v, omb1, b2, omb2, step_size, bc2s, eps);
            P[i] = p;
            Mo[i] = m;
            V[i] = v;
        }
    }
}

}  // namespace grafp

extern "C" int grafp_adam_multi_f32(float *const *params, const float *const *grads, float *const *exp_avgs,
                                    float *const *exp_avg_sqs, const float *const *steps, const int64_t *numels,
                                    int n_entries, const float *lr_dev, double lr, double beta1, double beta2, double eps,
                                    grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(n_entries >= 0 && (n_entries == 0 || (params && grads && exp_avgs && exp_avg_sqs && steps && numels)),
                  "adam_multi: null pointer");
    GRAFP_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, "adam_multi: bad hyper-parameters");
    for (int e0 = 0; e0 < n_entries; e0 += AD_MULTI) {
        AdamTable t;
        t.count = n_entries - e0 < AD_MULTI ? n_entries - e0 : AD_MULTI;
        int blocks = 0;
        for (int e = 0; e < t.count; ++e) {
            const int i = e0 + e;
            GRAFP_REQUIRE(params[i] && grads[i] && exp_avgs[i] && exp_avg_sqs[i] && steps[i] && numels[i] > 0 &&
                              numels[i] < (1ll << 31),
                          "adam_multi: bad entry %d", i);
            t.p[e] = params[i]; t.g[e] = grads[i]; t.m[e] = exp_avgs[i]; t.v[e] = exp_avg_sqs[i]; t.step[e] = steps[i];
            t.n[e] = (int)numels[i];
            t.block_base[e] = blocks;
            blocks += (int)((numels[i] + AD_BLOCK - 1) / AD_BLOCK);
        }
        t.block_base[t.count] = blocks;
        for (int e = t.count; e < AD_MULTI; ++e) {
            t.p[e] = nullptr; t.g[e] = nullptr; t.m[e] = nullptr; t.v[e] = nullptr; t.step[e] = nullptr; t.n[e] = 0;
            t.block_base[e + 1] = blocks;
        }
        t.lr_dev = lr_dev; t.lr = lr; t.beta1 = beta1; t.beta2 = beta2; t.eps = (float)eps;
        hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t);
        GRAFP_CHECK_LAUNCH("adam_multi_kernel");
    }
    return GRAFP_OK;
}
