// ntxent.hip -- fused NT-Xent forward + backward (K12 of SURVEY.md section 2a), gfx950.
//
// Replaces ntxent_loss (/root/reference/simclr/ntxent.py:4-29): the reference builds S = z z^T / tau
// (2B x 2B) and then runs a 2B-iteration Python loop of slice + cat + log_softmax (>= 1k tiny launches
// forward, as many backward).  Here:
//   pass 1 (ntxent_lse_kernel)   S tiles by exact-f32 MFMA (32x32x2), online log-sum-exp over c != r and
//                                the positive logit, per row                 -> lse[r], pos[r]
//                                (round 6: the candidate rows are split over `splits` workgroups per 32-row block --
//                                partial (max, sum, positive) per (split, row), combined in a fixed order by
//                                ntxent_lse_combine_kernel: 64 workgroups of 16 serial passes at 2 048 rows became
//                                256 of 4, and every rank of the data-parallel step runs this pass over ALL rows)
//   pass 2 (ntxent_grad_kernel)  S tiles recomputed; W = softmax_r + softmax_c - 2*onehot in registers;
//                                dZ_r += W^T Z_c by a second MFMA whose A operand IS the accumulator
//                                fragment (the k index of step `reg` is mfma_row(reg, half), no shuffle)
// S, softmax and one-hot never touch HBM.  Rows are ordered [view i | view j] (the loss is invariant to
// the reference's interleaved order), partner(r) = r +- B_all.
//
// Data-parallel form: every rank holds the all-gathered embeddings, runs pass 1 for ALL rows (2 B_all x
// 2 B_all x D flops, ~1 GFLOP at B_all = 1024: microseconds) and pass 2 only for its own rows, which
// yields d(mean loss)/d(z_local) exactly -- no collective in backward.
#include <math.h>

#include "common.h"

namespace grafp {

constexpr int NT_Q = 32;    // query rows per workgroup
constexpr int NT_C = 128;   // candidate rows per pass (32 per wave)
constexpr int NT_THREADS = 256;

__device__ __forceinline__ const float *nt_row(const float *zi, const float *zj, int Ball, int D, int r) {
    return r < Ball ? zi + (size_t)r * D : zj + (size_t)(r - Ball) * D;
}

// rows [r0, r0+nrows) -> dst[row * LS + c]; rows >= r_end are zero-filled.  LS = D + 1 (bank spread).
__device__ __forceinline__ void nt_stage(float *dst, const float *zi, const float *zj, int Ball, int D, int r0,
                                         int nrows, int r_end, int LS, int tid) {
    const int d4 = D >> 2;
    for (int i = tid; i < nrows * d4; i += NT_THREADS) {
        const int row = i / d4, c4 = i - row * d4;
        const int r = r0 + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < r_end) v = *reinterpret_cast<const float4 *>(nt_row(zi, zj, Ball, D, r) + c4 * 4);
        float *o = dst + row * LS + c4 * 4;
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
}

template <int D>
__device__ __forceinline__ f32x16 nt_s_tile(const float *sQ, const float *sCw, int LS, int l31, int half) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const float *qa = sQ + l31 * LS + half;
    const float *ca = sCw + l31 * LS + half;
#pragma unroll 8
    for (int kk = 0; kk < D; kk += 2) acc = mfma32x32x2(ca[kk], qa[kk], acc);
    return acc;
}

// ---- pass 1 -------------------------------------------------------------------------------------
template <int NDB>  // D / 32
__global__ __launch_bounds__(NT_THREADS) void ntxent_lse_kernel(const float *__restrict__ zi,
                                                                const float *__restrict__ zj, int Ball,
                                                                float inv_tau, float *__restrict__ lse,
                                                                float *__restrict__ pos, int cols_per_split,
                                                                float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = NDB * 32;
    constexpr int LS = D + 1;
    const int M = 2 * Ball;
    float *sQ = reinterpret_cast<float *>(smem);
    float *sC = sQ + NT_Q * LS;
    float *red = sC + NT_C * LS;  // [3][4][32]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int q0 = blockIdx.x * NT_Q;
    const int myq = q0 + l31;
    const int partner = myq < Ball ? myq + Ball : myq - Ball;
    nt_stage(sQ, zi, zj, Ball, D, q0, NT_Q, M, LS, tid);

    float m_run = -INFINITY, l_run = 0.0f, p = -INFINITY;
    // candidate rows [c_begin, c_end) of this workgroup (blockIdx.y = split; one split: all of them)
    const int c_begin = blockIdx.y * cols_per_split;
    const int c_end = c_begin + cols_per_split < M ? c_begin + cols_per_split : M;
    for (int c0 = c_begin; c0 < c_end; c0 += NT_C) {
        __syncthreads();
        nt_stage(sC, zi, zj, Ball, D, c0, NT_C, M, LS, tid);
        __syncthreads();
        const int cw = c0 + wave * 32;
        if (cw < c_end) {  // wave-uniform
            const f32x16 acc = nt_s_tile<D>(sQ, sC + wave * 32 * LS, LS, l31, half);
            float sv[16];
            float tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = cw + mfma_row(r, half);
                const bool valid = c < c_end && c != myq;
                const float s = valid ? acc[r] * inv_tau : -INFINITY;
                if (valid && c == partner) p = s;
                sv[r] = s;
                tmax = fmaxf(tmax, s);
            }
            const float nm = fmaxf(m_run, tmax);
            if (nm > -INFINITY) {
                float sum = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += expf(sv[r] - nm);   // exp(-inf) = 0 for masked entries
                l_run = l_run * expf(m_run - nm) + sum;
                m_run = nm;
            }
        }
    }
    {   // the other half-wave holds the other candidate rows of the same query
        const float om = __shfl_xor(m_run, 32), ol = __shfl_xor(l_run, 32), op = __shfl_xor(p, 32);
        const float nm = fmaxf(m_run, om);
        l_run = nm > -INFINITY ? l_run * expf(m_run - nm) + ol * expf(om - nm) : 0.0f;
        m_run = nm;
        p = fmaxf(p, op);
    }
    if (half == 0) {
        red[(0 * 4 + wave) * 32 + l31] = m_run;
        red[(1 * 4 + wave) * 32 + l31] = l_run;
        red[(2 * 4 + wave) * 32 + l31] = p;
    }
    __syncthreads();
    if (tid < 32 && myq < M) {
        float m = -INFINITY, pp = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            m = fmaxf(m, red[(0 * 4 + w) * 32 + tid]);
            pp = fmaxf(pp, red[(2 * 4 + w) * 32 + tid]);
        }
        float l = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = red[(0 * 4 + w) * 32 + tid];
            if (mw > -INFINITY) l += red[(1 * 4 + w) * 32 + tid] * expf(mw - m);
        }
        if (gridDim.y == 1) {
            lse[myq] = m + logf(l);
            pos[myq] = pp;
        } else {                                   // partial (max, sum, positive) of this split: combined below
            float *o = part + ((size_t)blockIdx.y * M + myq) * 3;
            o[0] = m; o[1] = l; o[2] = pp;
        }
    }
}

// lse[r], pos[r] from the splits' partial (max, sum, positive) triples, splits in ascending order (deterministic)
__global__ __launch_bounds__(256) void ntxent_lse_combine_kernel(const float *__restrict__ part, int M, int splits,
                                                                 float *__restrict__ lse, float *__restrict__ pos) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= M) return;
    float m = -INFINITY, pp = -INFINITY;
    for (int s2 = 0; s2 < splits; ++s2) {
        m = fmaxf(m, part[((size_t)s2 * M + r) * 3]);
        pp = fmaxf(pp, part[((size_t)s2 * M + r) * 3 + 2]);
    }
    float l = 0.0f;
    for (int s2 = 0; s2 < splits; ++s2) {
        const float ms = part[((size_t)s2 * M + r) * 3];
        if (ms > -INFINITY) l += part[((size_t)s2 * M + r) * 3 + 1] * expf(ms - m);
    }
    lse[r] = m + logf(l);
    pos[r] = pp;
}

// ---- pass 2 -------------------------------------------------------------------------------------
template <int NDB>  // D / 32
__global__ __launch_bounds__(NT_THREADS) void ntxent_grad_kernel(const float *__restrict__ zi,
                                                                 const float *__restrict__ zj, int Ball,
                                                                 int row_begin, int n_local, float inv_tau,
                                                                 const float *__restrict__ lse,
                                                                 const float *__restrict__ pos,
                                                                 float *__restrict__ loss_partial,
                                                                 float *__restrict__ dzi, float *__restrict__ dzj,
                                                                 int cols_per_split, float *__restrict__ dzpart) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = NDB * 32;
    constexpr int LS = D + 1;
    const int M = 2 * Ball;
    float *sQ = reinterpret_cast<float *>(smem);
    float *sC = sQ + NT_Q * LS;
    float *sLse = sC + NT_C * LS;  // [NT_C]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int view = blockIdx.y;
    const int q0l = blockIdx.x * NT_Q;                 // first local pair of this block
    const int r_first = view * Ball + row_begin + q0l;  // its global row
    const int r_end = view * Ball + row_begin + n_local;
    const int rq = r_first + l31;
    const bool valid_q = rq < r_end;
    const int partner = rq < Ball ? rq + Ball : rq - Ball;
    const float lse_q = valid_q ? lse[rq] : 0.0f;
    nt_stage(sQ, zi, zj, Ball, D, r_first, NT_Q, r_end, LS, tid);

    f32x16 dz[NDB];
#pragma unroll
    for (int b = 0; b < NDB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) dz[b][r] = 0.0f;

    // candidate rows [c_begin, c_end) of this workgroup (blockIdx.z = split; round 6: at 128 local rows of 2 048 -- one
    // rank of the 8-GPU step -- eight workgroups walked all sixteen 128-row chunks one after the other: 160 us)
    const int c_begin = blockIdx.z * cols_per_split;
    const int c_end = c_begin + cols_per_split < M ? c_begin + cols_per_split : M;
    for (int c0 = c_begin; c0 < c_end; c0 += NT_C) {
        __syncthreads();
        nt_stage(sC, zi, zj, Ball, D, c0, NT_C, M, LS, tid);
        if (tid < NT_C) sLse[tid] = (c0 + tid < M) ? lse[c0 + tid] : 0.0f;
        __syncthreads();
        const int cw = c0 + wave * 32;
        if (cw < c_end) {  // wave-uniform
            const float *sCw = sC + wave * 32 * LS;
            const f32x16 acc = nt_s_tile<D>(sQ, sCw, LS, l31, half);
            float w[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = wave * 32 + mfma_row(r, half);
                const int c = c0 + cl;
                const bool valid = valid_q && c < c_end && c != rq;
                const float s = acc[r] * inv_tau;
                const float v = expf(s - lse_q) + expf(s - sLse[cl]) - (c == partner ? 2.0f : 0.0f);
                w[r] = valid ? v : 0.0f;
            }
            // dZ[q][d] += sum_c W[c][q] * Z[c][d]; k-step r contracts candidate rows mfma_row(r, 0|1)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float *zrow = sCw + mfma_row(r, half) * LS + l31;
#pragma unroll
                for (int b = 0; b < NDB; ++b) dz[b] = mfma32x32x2(w[r], zrow[b * 32], dz[b]);
            }
        }
    }
    // deterministic cross-wave sum through LDS (reuses the candidate tile)
    float *buf = sC;
    for (int wv = 0; wv < 4; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int b = 0; b < NDB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int a = mfma_row(r, half) * D + b * 32 + l31;
                    buf[a] = (wv == 0 ? 0.0f : buf[a]) + dz[b][r];
                }
        }
    }
    __syncthreads();
    const float scale = inv_tau / (float)M;
    // one split: the gradient itself; several: this split's partial sum (ntxent_dz_combine_kernel adds them in order)
    float *out = gridDim.z == 1 ? (view == 0 ? dzi : dzj) + (size_t)q0l * D
                                : dzpart + ((size_t)(blockIdx.z * 2 + view) * n_local + q0l) * D;
    for (int i = tid; i < NT_Q * D; i += NT_THREADS)
        if (q0l + i / D < n_local) out[i] = buf[i] * scale;

    if (wave == 0 && blockIdx.z == 0) {
        float v = (half == 0 && valid_q) ? (lse_q - pos[rq]) : 0.0f;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) loss_partial[blockIdx.y * gridDim.x + blockIdx.x] = v;
    }
}

// dz[view][row][d] = sum over the splits, ascending (deterministic)
__global__ __launch_bounds__(256) void ntxent_dz_combine_kernel(const float *__restrict__ dzpart, int64_t n_per_view,
                                                                int splits, float *__restrict__ dzi,
                                                                float *__restrict__ dzj) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * n_per_view) return;
    float acc = 0.0f;
    for (int s2 = 0; s2 < splits; ++s2) acc += dzpart[(size_t)s2 * 2 * n_per_view + i];
    if (i < n_per_view) dzi[i] = acc;
    else dzj[i - n_per_view] = acc;
}

}  // namespace grafp

// candidate splits of pass 1: NT_C-row chunks dealt to up to 8 workgroups per row block, >= 2 chunks each
static int nt_splits(int M) {
    const int chunks = (M + grafp::NT_C - 1) / grafp::NT_C;
    int sp = chunks / 2;
    if (sp > 8) sp = 8;
    return sp < 1 ? 1 : sp;
}
extern "C" size_t grafp_ntxent_workspace(int B_all) {
    if (B_all <= 0) return 0;
    const size_t M = (size_t)2 * B_all;
    // lse[2B] + pos[2B] + partial triples of pass 1 + partial gradients of pass 2 (n_local <= B_all rows, D <= 128)
    return (2 * M + (size_t)nt_splits((int)M) * M * 3 + (size_t)nt_splits((int)M) * M * 128) * sizeof(float);
}

extern "C" int grafp_ntxent_num_partials(int n_local) {
    return n_local > 0 ? 2 * ((n_local + grafp::NT_Q - 1) / grafp::NT_Q) : 0;
}

extern "C" int grafp_ntxent_fwd_bwd_f32(const float *zi_all, const float *zj_all, int B_all, int D, int row_begin,
                                        int n_local, float tau, float *loss_partial, float *dzi, float *dzj, void *ws,
                                        size_t ws_bytes, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(zi_all && zj_all && loss_partial && dzi && dzj, "ntxent: null pointer");
    GRAFP_REQUIRE(B_all >= 1 && D >= 32 && D <= 128 && D % 32 == 0, "ntxent: need B_all >= 1 and D in {32,64,96,128} (B_all=%d D=%d)",
                  B_all, D);
    GRAFP_REQUIRE(row_begin >= 0 && n_local >= 1 && row_begin + n_local <= B_all,
                  "ntxent: local range [%d, %d) outside [0, %d)", row_begin, row_begin + n_local, B_all);
    GRAFP_REQUIRE(tau > 0.0f, "ntxent: tau must be positive");
    GRAFP_REQUIRE(((uintptr_t)zi_all & 15) == 0 && ((uintptr_t)zj_all & 15) == 0, "ntxent: embeddings must be 16-byte aligned");
    if (!ws || ws_bytes < grafp_ntxent_workspace(B_all)) {
        set_error("ntxent: workspace %zu bytes < required %zu", ws_bytes, grafp_ntxent_workspace(B_all));
        return GRAFP_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int M = 2 * B_all, LS = D + 1;
    float *lse = (float *)ws, *pos = lse + M, *part = pos + M;
    const float inv_tau = 1.0f / tau;
    const int splits = nt_splits(M);
    const int chunks = (M + NT_C - 1) / NT_C;
    const int cols_per_split = ((chunks + splits - 1) / splits) * NT_C;
    const size_t lds1 = ((size_t)(NT_Q + NT_C) * LS + 3 * 4 * 32) * sizeof(float);
    const size_t lds2 = ((size_t)(NT_Q + NT_C) * LS + NT_C) * sizeof(float);
    const dim3 grid1((M + NT_Q - 1) / NT_Q, splits);
    // pass 2: candidate splits so that ~256 workgroups run (never more than pass 1's, >= 2 chunks each)
    int splits2 = 256 / (2 * ((n_local + NT_Q - 1) / NT_Q));
    if (splits2 > splits) splits2 = splits;
    if (splits2 < 1) splits2 = 1;
    const int cols_per_split2 = ((chunks + splits2 - 1) / splits2) * NT_C;
    float *dzpart = part + (size_t)splits * M * 3;
    const dim3 grid((n_local + NT_Q - 1) / NT_Q, 2, splits2);
#define NT_LAUNCH(NDB)                                                                                              \
    (void)hipFuncSetAttribute((const void *)ntxent_lse_kernel<NDB>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                              (int)lds1);                                                                           \
    hipLaunchKernelGGL(ntxent_lse_kernel<NDB>, grid1, dim3(NT_THREADS), lds1, s, zi_all, zj_all, B_all, inv_tau,    \
                       lse, pos, cols_per_split, part);                                                             \
    if (splits > 1)                                                                                                 \
        hipLaunchKernelGGL(ntxent_lse_combine_kernel, dim3((M + 255) / 256), dim3(256), 0, s, (const float *)part,  \
                           M, splits, lse, pos);                                                                    \
    (void)hipFuncSetAttribute((const void *)ntxent_grad_kernel<NDB>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                              (int)lds2);                                                                           \
    hipLaunchKernelGGL(ntxent_grad_kernel<NDB>, grid, dim3(NT_THREADS), lds2, s, zi_all, zj_all, B_all, row_begin,  \
                       n_local, inv_tau, lse, pos, loss_partial, dzi, dzj, cols_per_split2, dzpart);                \
    if (splits2 > 1)                                                                                                \
        hipLaunchKernelGGL(ntxent_dz_combine_kernel, dim3((unsigned)(((int64_t)2 * n_local * D + 255) / 256)),      \
                           dim3(256), 0, s, (const float *)dzpart, (int64_t)n_local * D, splits2, dzi, dzj)
    switch (D / 32) {
        case 1: NT_LAUNCH(1); break;
        case 2: NT_LAUNCH(2); break;
        case 3: NT_LAUNCH(3); break;
        default: NT_LAUNCH(4); break;
    }
#undef NT_LAUNCH
    GRAFP_CHECK_LAUNCH("ntxent_lse_kernel / ntxent_grad_kernel");
    return GRAFP_OK;
}
