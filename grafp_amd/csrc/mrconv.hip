// mrconv.hip -- edge gather + max-relative aggregation (K6+K7 of SURVEY.md section 2a), gfx950.
//
// Replaces, for the live 'mr' graph conv (/root/reference/encoder/gcn_lib/torch_vertex.py:19-34):
//   x_i = batched_index_select(x, centre)   torch_nn.py:79-98   (a (B,C,N,k) copy of x itself)
//   x_j = batched_index_select(x, nn_idx)                       (a (B,C,N,k) gather)
//   rel = max_k(x_j - x_i) ; out = interleave(x, rel)           torch_vertex.py:29-32
// The reference moves ~4 full-tensor copies per call through HBM (201 MB twice at B=256 stage 0); here a
// workgroup stages a slab of channel rows of one clip in LDS, gathers neighbours from LDS and writes
// the interleaved (B,2C,N) result once.  HBM-bound: 4CN + 8kN read, 8CN written per clip.
//
// Backward recomputes the arg-max neighbour (first maximum, as torch.max on CPU) and scatter-adds into
// an LDS accumulator row, so dx is written once, coalesced, with no global atomics.
#include <math.h>

#define GRAFP_STORE_FAMILY 3        // (common.h: GRAFP_ST_NT experiment builds)
#include "common.h"
#include "tuning.h"

namespace grafp {

constexpr int MR_THREADS = 256;

__device__ __forceinline__ int clampi(int64_t v, int n) {
    return v < 0 ? 0 : (v >= n ? n - 1 : (int)v);
}

// f32 / bf16 element access (bf16 carried as unsigned short; round-to-nearest-even on store)
__device__ __forceinline__ float mr_ld(const float *p) { return *p; }
__device__ __forceinline__ float mr_ld(const unsigned short *p) { return __uint_as_float(((unsigned)*p) << 16); }
__device__ __forceinline__ void mr_st(float *p, float v) { *p = v; }
__device__ __forceinline__ void mr_st(unsigned short *p, float v) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) { *p = (unsigned short)((u >> 16) | 0x40); return; }   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    *p = (unsigned short)(u >> 16);
}

// 4-wide access (16 B of f32 / 8 B of bf16); callers guarantee 4-element alignment of the address
__device__ __forceinline__ void mr_ld4(const float *p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4 *>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void mr_ld4(const unsigned short *p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2 *>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
// a 4-element piece as it is loaded (what waits in registers while further slabs are in flight)
template <typename T> struct MrRaw;
template <> struct MrRaw<float> { typedef float4 type; static constexpr int NPAR = 2; };
template <> struct MrRaw<unsigned short> { typedef uint2 type; static constexpr int NPAR = 4; };
__device__ __forceinline__ void mr_unpack4(const float4 &t, float (&v)[4]) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
__device__ __forceinline__ void mr_unpack4(const uint2 &t, float (&v)[4]) {
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
__device__ __forceinline__ unsigned mr_pack(float lo, float hi) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const f2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));      // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
}
// (outputs are streamed with the non-temporal hint: +35-65 % on a plain copy of tensors this size, tools/microbench/copy_bench.hip)
// ... and with PLAIN stores when the result fits the Infinity Cache beside its reader's other operand (`plain`, wave-uniform:
// mr_plain_stores below; same-box A/B of the whole step, profiles/r06_c_*, r06_d_*: -0.5 % at 128 and 256 pairs, +0.5 % at
// 512 and 1024)
__device__ __forceinline__ void mr_st4(float *p, const float (&v)[4], bool plain = false) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 t = {v[0], v[1], v[2], v[3]};
    if (plain) store16_hint(p, __builtin_bit_cast(st_u32x4, t), true);
    else GRAFP_ST_NT(t, reinterpret_cast<f4 *>(p));
}
__device__ __forceinline__ void mr_st4(unsigned short *p, const float (&v)[4], bool plain = false) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 t = {mr_pack(v[0], v[1]), mr_pack(v[2], v[3])};
    if (plain) store8_hint(p, __builtin_bit_cast(st_u32x2, t), true);
    else GRAFP_ST_NT(t, reinterpret_cast<u2 *>(p));
}

// The backward scatter accumulates in 64-bit FIXED POINT with integer LDS atomics (ds_add_f32 runs at 0.33 lane-ops per
// clock per CU on gfx950, ds_add_u64 at 10: tools/microbench/lds_atomic_bench.hip; and the sum no longer depends on the
// order of the atomics: the gradient is deterministic).  Per slab the scale is 2^(39 - e), m = max |addend| < 2^e: an
// addend becomes the integer v = rint(g scale), |v| < 2^39, carried as TWO signed 32-bit fields of one 64-bit word,
// v = hi 2^20 + lo (|lo| < 2^20, |hi| < 2^19): a node receives at most N <= 2048 addends, so neither field's sum leaves
// its 31 bits, the 64-bit add keeps the borrows right, and building / reading the word costs 9 + 7 VALU instructions
// instead of the ~35 of the f32 <-> i64 conversions (both kernels were VALU-bound on them: 60 VALU instructions per
// element, 300 us per call where the bytes take 130).
__device__ __forceinline__ unsigned long long mr_fix_encode(float g, float scale) {
    const float v = __builtin_rintf(g * scale);
    const float hf = __builtin_truncf(v * 9.5367431640625e-7f);                 // 2^-20
    const float lf = __builtin_fmaf(hf, -1048576.0f, v);                        // exact
    const int hi = (int)hf, lo = (int)lf;
    return ((unsigned long long)(unsigned)(hi + (lo >> 31)) << 32) | (unsigned long long)(unsigned)lo;
}
__device__ __forceinline__ float mr_fix_decode(unsigned long long a) {
    const int lo = (int)(unsigned)a;
    const int hi = (int)(unsigned)(a >> 32) - (lo >> 31);
    return (float)__builtin_fma((double)hi, 1048576.0, (double)lo);             // the sum exactly, rounded once
}
// scale of a slab from the largest |addend| bits (integer maximum of the sign-stripped patterns: NaN / inf sort last)
__device__ __forceinline__ void mr_fix_scale(unsigned mbits, bool &poisoned, float &scale, float &inv_scale) {
    poisoned = mbits >= 0x7f800000u;
    const float m = __uint_as_float(mbits);
    int ex = 0;
    (void)frexpf(m, &ex);                                     // m < 2^ex
    int sh = mbits ? 39 - ex : 0;
    sh = sh > 100 ? 100 : sh;
    scale = ldexpf(1.0f, sh);
    inv_scale = ldexpf(1.0f, -sh);
}

// Activations are addressed as base + b*sb + c*sc + n (N contiguous): (B,C,N) has sb = C*N, sc = N; the
// GEMM-friendly (C,B,N) has sb = N, sc = B*N.  dynamic LDS: rows[CC*N] f32 | sidx[K*N] i32 (bwd: + acc[CC*N] i64, base f32).
// A workgroup owns CC channel rows of one clip; thread t walks the slab's elements t*V, t*V + 256*V, ... as a
// running (channel, node) pair -- no integer division in any loop.  V = 4 when N % 4 == 0 and the strides and
// base pointers are 4-element aligned (always true for the encoder's shapes), else 1.
#define MR_WALK(V, tid, N, c, n)                    \
    int c = ((tid) * (V)) / (N), n = ((tid) * (V)) - c * (N)
#define MR_NEXT(V, N, c, n)                         \
    do {                                            \
        n += MR_THREADS * (V);                      \
        while (n >= (N)) { n -= (N); ++c; }         \
    } while (0)

template <typename I>
__device__ __forceinline__ void stage_idx(int *sidx, const I *__restrict__ idxb, int N, int K, int tid) {
    for (int n = tid; n < N; n += MR_THREADS)
        for (int k = 0; k < K; ++k) sidx[k * N + n] = clampi((int64_t)idxb[(size_t)n * K + k], N);
}

template <typename T, int V, typename I>
__global__ __launch_bounds__(MR_THREADS) void mrconv_fwd_kernel(const T *__restrict__ x, int64_t x_sb, int64_t x_sc,
                                                                const I *__restrict__ idx, T *__restrict__ out,
                                                                int64_t o_sb, int64_t o_sc, int C, int N, int K,
                                                                int CC) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y, c0 = blockIdx.x * CC, tid = threadIdx.x;
    const int cc = min(CC, C - c0);
    float *rows = reinterpret_cast<float *>(smem);
    int *sidx = reinterpret_cast<int *>(rows + (size_t)CC * N);
    const T *xb = x + (size_t)b * x_sb + (size_t)c0 * x_sc;
    {
        MR_WALK(V, tid, N, c, n);
        while (c < cc) {
            float v[4];
            if (V == 4) mr_ld4(xb + (size_t)c * x_sc + n, v); else v[0] = mr_ld(xb + (size_t)c * x_sc + n);
#pragma unroll
            for (int e = 0; e < V; ++e) rows[c * N + n + e] = v[e];
            MR_NEXT(V, N, c, n);
        }
    }
    stage_idx<I>(sidx, idx + (size_t)b * N * K, N, K, tid);
    __syncthreads();

    T *ob = out + (size_t)b * o_sb + (size_t)(2 * c0) * o_sc;
    MR_WALK(V, tid, N, c, n);
    while (c < cc) {
        const float *row = rows + c * N;
        float xi[4], m[4];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            xi[e] = row[n + e];
            m[e] = -INFINITY;
        }
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int e = 0; e < V; ++e) m[e] = fmaxf(m[e], row[sidx[k * N + n + e]] - xi[e]);
        if (V == 4) {
            mr_st4(ob + (size_t)(2 * c) * o_sc + n, xi);
            mr_st4(ob + (size_t)(2 * c + 1) * o_sc + n, m);
        } else {
            mr_st(ob + (size_t)(2 * c) * o_sc + n, xi[0]);
            mr_st(ob + (size_t)(2 * c + 1) * o_sc + n, m[0]);
        }
        MR_NEXT(V, N, c, n);
    }
}

// Persistent 4-wide variant (N % 4 == 0, 4-element aligned strides/pointers -- every shape of the encoder): a
// workgroup stages the clip's edges ONCE and then walks several channel slabs of the clip, with the rows of the next
// slab already in flight (registers) while the current one is gathered out of LDS.  The one-slab-per-workgroup
// kernel above re-staged the 12 N bytes of edges for every 4 channel rows and exposed every load to latency.
constexpr int MRP_ITEMS = 4;                      // 4-element pieces per thread per slab: slab = 4096 elements
constexpr int MRP_SLAB = MRP_ITEMS * MR_THREADS * 4;

// WK: also write which neighbour won (first maximum, the rule of the backward pass) -- 2 bits per element, the four
// elements of a piece in one byte of arg (B, C, N / 4); K <= 4.  The backward pass then needs neither x nor the gather.
template <typename T, typename I, bool WK>
__global__ __launch_bounds__(MR_THREADS) void mrconv_fwd_p_kernel(const T *__restrict__ x, int64_t x_sb, int64_t x_sc,
                                                                  const I *__restrict__ idx, T *__restrict__ out,
                                                                  int64_t o_sb, int64_t o_sc, int C, int N, int K,
                                                                  int CC, unsigned char *__restrict__ arg, int plain) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y, tid = threadIdx.x;
    float *rows = reinterpret_cast<float *>(smem);                 // [CC*N] (<= MRP_SLAB floats)
    int *sidx = reinterpret_cast<int *>(rows + MRP_SLAB);          // [K][N]
    const int nslab = (C + CC - 1) / CC;
    const T *xb = x + (size_t)b * x_sb;
    T *ob = out + (size_t)b * o_sb;
    // this thread's pieces inside a slab: (channel offset, node) of piece it
    int pc[MRP_ITEMS], pn[MRP_ITEMS];
    {
        MR_WALK(4, tid, N, c, n);
#pragma unroll
        for (int it = 0; it < MRP_ITEMS; ++it) {
            pc[it] = c;
            pn[it] = n;
            MR_NEXT(4, N, c, n);
        }
    }
    // SEVERAL slabs in flight per workgroup: with one, 5 workgroups x 8 KB per CU = 10 MB on the whole chip, and bytes
    // in flight / HBM latency (~3.5 us under load) is what the kernel ran at (3.5 TB/s).  The pieces wait in registers
    // as they were loaded (bf16: 8 bytes per piece instead of four floats), so the same 32 registers hold FOUR slabs of
    // bf16 -- all a workgroup has with the usual 16 slabs per clip -- or two of f32.
    typedef typename MrRaw<T>::type Raw;
    constexpr int NPAR = MrRaw<T>::NPAR;
    Raw pv[NPAR][MRP_ITEMS];
    auto fetch = [&](int par, int slab) {
        const int c0 = slab * CC, cc = min(CC, C - c0);
#pragma unroll
        for (int it = 0; it < MRP_ITEMS; ++it)
            if (pc[it] < cc) pv[par][it] = GRAFP_LD_ONCE(32, reinterpret_cast<const Raw *>(xb + (size_t)(c0 + pc[it]) * x_sc + pn[it]));
    };
    int slab = blockIdx.x;
    const int G = gridDim.x;
#pragma unroll
    for (int par = 0; par < NPAR; ++par)
        if (slab + par * G < nslab) fetch(par, slab + par * G);
    stage_idx<I>(sidx, idx + (size_t)b * N * K, N, K, tid);
    while (slab < nslab) {
#pragma unroll
        for (int par = 0; par < NPAR; ++par) {
            if (slab < nslab) {                    // workgroup-uniform
                const int c0 = slab * CC, cc = min(CC, C - c0);
                __syncthreads();                   // previous slab fully consumed (and sidx staged, first time round)
                float xi[MRP_ITEMS][4];
#pragma unroll
                for (int it = 0; it < MRP_ITEMS; ++it) mr_unpack4(pv[par][it], xi[it]);
#pragma unroll
                for (int it = 0; it < MRP_ITEMS; ++it)
                    if (pc[it] < cc)                   // one 16-byte LDS write per piece (N % 4 == 0: aligned)
                        *reinterpret_cast<float4 *>(rows + pc[it] * N + pn[it]) =
                            make_float4(xi[it][0], xi[it][1], xi[it][2], xi[it][3]);
                __syncthreads();
                if (slab + NPAR * G < nslab) fetch(par, slab + NPAR * G);
#pragma unroll
                for (int it = 0; it < MRP_ITEMS; ++it) {
                    if (pc[it] < cc) {
                        const float *row = rows + pc[it] * N;
                        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                        float best[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                        unsigned bk = 0;
                        for (int k = 0; k < K; ++k) {
                            const int4 j4 = *reinterpret_cast<const int4 *>(sidx + k * N + pn[it]);
                            const float v[4] = {row[j4.x] - xi[it][0], row[j4.y] - xi[it][1], row[j4.z] - xi[it][2],
                                                row[j4.w] - xi[it][3]};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                m[e] = fmaxf(m[e], v[e]);
                                if (WK && (k == 0 || v[e] > best[e])) {  // mrconv_bwd's routing rule, comparison for comparison
                                    best[e] = v[e];
                                    bk = (bk & ~(3u << (2 * e))) | ((unsigned)k << (2 * e));
                                }
                            }
                        }
                        if (WK) arg[((size_t)b * C + c0 + pc[it]) * (N / 4) + pn[it] / 4] = (unsigned char)bk;
                        T *o = ob + (size_t)(2 * (c0 + pc[it])) * o_sc + pn[it];
                        mr_st4(o, xi[it], plain != 0);
                        mr_st4(o + o_sc, m, plain != 0);
                    }
                }
                slab += G;
            }
        }
    }
}

// BWD_ITEMS * 256 * V elements per workgroup: the odd-channel gradients stay in registers between the
// accumulator initialisation and the scatter phase.
// Measured: small slabs (ITEMS = 2: 2048 elements, ~36 KB of LDS, 4 workgroups per CU) beat 8192-element ones by
// 25 %; ITEMS = 8 remains for N too long for a small slab to hold a whole channel row.

template <typename T, int V, typename I, int BWD_ITEMS>
__global__ __launch_bounds__(MR_THREADS) void mrconv_bwd_kernel(const T *__restrict__ x, int64_t x_sb, int64_t x_sc,
                                                                const I *__restrict__ idx,
                                                                const T *__restrict__ gout, int64_t g_sb, int64_t g_sc,
                                                                T *__restrict__ dx, int C, int N, int K, int CC) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned s_max[MR_THREADS / 64];
    const int b = blockIdx.y, c0 = blockIdx.x * CC, tid = threadIdx.x;
    const int cc = min(CC, C - c0);
    // 64-bit fixed-point scatter accumulator + integer LDS atomics (see mrconv_bwd_p_kernel): deterministic, and
    // ds_add_u64 is 30x the rate of ds_add_f32 on gfx950
    const size_t slab_al = ((size_t)CC * N + 1) & ~(size_t)1;
    long long *acc = reinterpret_cast<long long *>(smem);
    float *rows = reinterpret_cast<float *>(acc + slab_al);
    float *base = rows + slab_al;                                 // g_even - g_odd (identity branch minus the centre terms)
    int *sidx = reinterpret_cast<int *>(base + slab_al);
    const T *xb = x + (size_t)b * x_sb + (size_t)c0 * x_sc;
    const T *gb = gout + (size_t)b * g_sb + (size_t)(2 * c0) * g_sc;
    float godd[BWD_ITEMS][4];
    unsigned m = 0;
    {   // stage x; base <- g_even - g_odd
        MR_WALK(V, tid, N, c, n);
#pragma unroll
        for (int it = 0; it < BWD_ITEMS; ++it) {
            if (c < cc) {
                float v[4], ge[4];
                if (V == 4) {
                    mr_ld4(xb + (size_t)c * x_sc + n, v);
                    mr_ld4(gb + (size_t)(2 * c) * g_sc + n, ge);
                    mr_ld4(gb + (size_t)(2 * c + 1) * g_sc + n, godd[it]);
                } else {
                    v[0] = mr_ld(xb + (size_t)c * x_sc + n);
                    ge[0] = mr_ld(gb + (size_t)(2 * c) * g_sc + n);
                    godd[it][0] = mr_ld(gb + (size_t)(2 * c + 1) * g_sc + n);
                }
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    rows[c * N + n + e] = v[e];
                    acc[c * N + n + e] = 0;
                    base[c * N + n + e] = ge[e] - godd[it][e];
                    m = max(m, __float_as_uint(godd[it][e]) & 0x7fffffffu);
                }
            }
            MR_NEXT(V, N, c, n);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((tid & 63) == 0) s_max[tid >> 6] = m;
    stage_idx<I>(sidx, idx + (size_t)b * N * K, N, K, tid);
    __syncthreads();
    m = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    bool poisoned;
    float scale, inv_scale;
    mr_fix_scale(m, poisoned, scale, inv_scale);
    if (!poisoned) {   // route g_odd[c][m] to the arg-max neighbour of m (first maximum)
        MR_WALK(V, tid, N, c, n);
#pragma unroll
        for (int it = 0; it < BWD_ITEMS; ++it) {
            if (c < cc) {
                const float *row = rows + c * N;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float xi = row[n + e];
                    float best = -INFINITY;
                    int bj = sidx[n + e];
                    for (int k = 0; k < K; ++k) {
                        const int j = sidx[k * N + n + e];
                        const float v = row[j] - xi;
                        if (v > best) { best = v; bj = j; }
                    }
                    atomicAdd(reinterpret_cast<unsigned long long *>(&acc[c * N + bj]), mr_fix_encode(godd[it][e], scale));
                }
            }
            MR_NEXT(V, N, c, n);
        }
    }
    __syncthreads();
    T *db = dx + (size_t)b * x_sb + (size_t)c0 * x_sc;
    MR_WALK(V, tid, N, c, n);
    while (c < cc) {
        if (V == 4) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[e] = poisoned ? NAN : base[c * N + n + e] + mr_fix_decode((unsigned long long)acc[c * N + n + e]) * inv_scale;
            mr_st4(db + (size_t)c * x_sc + n, v);
        } else {
            mr_st(db + (size_t)c * x_sc + n,
                  poisoned ? NAN : base[c * N + n] + mr_fix_decode((unsigned long long)acc[c * N + n]) * inv_scale);
        }
        MR_NEXT(V, N, c, n);
    }
}

// Persistent 4-wide backward (same conditions as mrconv_fwd_p_kernel): edges staged once per workgroup, slabs of
// 2048 elements (2 pieces per thread), the next slab's x / g_even / g_odd in flight while the current one scatters.
constexpr int MRB_ITEMS = 2;
constexpr int MRB_SLAB = MRB_ITEMS * MR_THREADS * 4;

template <typename T, typename I>
__global__ __launch_bounds__(MR_THREADS) void mrconv_bwd_p_kernel(const T *__restrict__ x, int64_t x_sb, int64_t x_sc,
                                                                  const I *__restrict__ idx,
                                                                  const T *__restrict__ gout, int64_t g_sb,
                                                                  int64_t g_sc, T *__restrict__ dx, int C, int N, int K,
                                                                  int CC) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned s_max[MR_THREADS / 64];
    const int b = blockIdx.y, tid = threadIdx.x;
    long long *acc = reinterpret_cast<long long *>(smem);          // [CC*N] fixed-point scatter accumulator
    float *rows = reinterpret_cast<float *>(acc + MRB_SLAB);       // [CC*N] x
    int *sidx = reinterpret_cast<int *>(rows + MRB_SLAB);          // [K][N]
    const int nslab = (C + CC - 1) / CC;
    const T *xb = x + (size_t)b * x_sb;
    const T *gb = gout + (size_t)b * g_sb;
    T *db = dx + (size_t)b * x_sb;
    int pc[MRB_ITEMS], pn[MRB_ITEMS];
    {
        MR_WALK(4, tid, N, c, n);
#pragma unroll
        for (int it = 0; it < MRB_ITEMS; ++it) {
            pc[it] = c;
            pn[it] = n;
            MR_NEXT(4, N, c, n);
        }
    }
    float pv[MRB_ITEMS][4], pe[MRB_ITEMS][4], po[MRB_ITEMS][4];     // x, g_even, g_odd of the slab in flight
    auto fetch = [&](int slab) {
        const int c0 = slab * CC, cc = min(CC, C - c0);
#pragma unroll
        for (int it = 0; it < MRB_ITEMS; ++it)
            if (pc[it] < cc) {
                const int c = c0 + pc[it];
                mr_ld4(xb + (size_t)c * x_sc + pn[it], pv[it]);
                mr_ld4(gb + (size_t)(2 * c) * g_sc + pn[it], pe[it]);
                mr_ld4(gb + (size_t)(2 * c + 1) * g_sc + pn[it], po[it]);
            }
    };
    int slab = blockIdx.x;
    if (slab < nslab) fetch(slab);
    stage_idx<I>(sidx, idx + (size_t)b * N * K, N, K, tid);
    for (; slab < nslab; slab += gridDim.x) {
        const int c0 = slab * CC, cc = min(CC, C - c0);
        __syncthreads();                       // previous slab written out (and sidx staged, first time round)
        float xi[MRB_ITEMS][4], godd[MRB_ITEMS][4], base[MRB_ITEMS][4];
        unsigned m = 0;
#pragma unroll
        for (int it = 0; it < MRB_ITEMS; ++it)
            if (pc[it] < cc) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rows[pc[it] * N + pn[it] + e] = pv[it][e];
                    acc[pc[it] * N + pn[it] + e] = 0;
                    base[it][e] = pe[it][e] - po[it][e];       // identity branch minus the centre terms
                    xi[it][e] = pv[it][e];
                    godd[it][e] = po[it][e];
                    m = max(m, __float_as_uint(po[it][e]) & 0x7fffffffu);
                }
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        if ((tid & 63) == 0) s_max[tid >> 6] = m;
        __syncthreads();
        if (slab + gridDim.x < nslab) fetch(slab + gridDim.x);
        m = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        bool poisoned;                                          // a non-finite gradient: the slab's output is NaN
        float scale, inv_scale;
        mr_fix_scale(m, poisoned, scale, inv_scale);
        // route g_odd[c][m] to the arg-max neighbour of m (first maximum)
#pragma unroll
        for (int it = 0; it < MRB_ITEMS; ++it) {
            if (pc[it] < cc) {
                const float *row = rows + pc[it] * N;
                float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                int bj[4];
                for (int k = 0; k < K; ++k) {
                    const int4 j4 = *reinterpret_cast<const int4 *>(sidx + k * N + pn[it]);
                    const int jj[4] = {j4.x, j4.y, j4.z, j4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = row[jj[e]] - xi[it][e];
                        if (k == 0 || v > best[e]) { best[e] = v; bj[e] = jj[e]; }
                    }
                }
                if (!poisoned) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        atomicAdd(reinterpret_cast<unsigned long long *>(&acc[pc[it] * N + bj[e]]),
                                  mr_fix_encode(godd[it][e], scale));
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < MRB_ITEMS; ++it)
            if (pc[it] < cc) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = poisoned ? NAN : base[it][e] + mr_fix_decode((unsigned long long)acc[pc[it] * N + pn[it] + e]) * inv_scale;
                mr_st4(db + (size_t)(c0 + pc[it]) * x_sc + pn[it], v);
            }
    }
}

// Backward from the recorded arg-max (mrconv_fwd_p_kernel<WK>): no x, no gather.  Per element: one byte-quarter, one
// fixed-point LDS atomic.  LDS: i64 accumulator slab | edges [K][N].  A thread owns 4 CONSECUTIVE elements, so every
// per-element LDS access of a wave strides 16 or 32 bytes per lane (PMC, first version: 70 % of the LDS-active cycles
// were bank conflicts, the LDS busy 60 % of the kernel): the accumulator is zeroed and read back as 16-byte vectors and
// the K edges of the thread's four nodes come as K 16-byte reads, selected in registers.
template <typename T, typename I>
__global__ __launch_bounds__(MR_THREADS) void mrconv_bwd_a_kernel(const unsigned char *__restrict__ arg,
                                                                  const I *__restrict__ idx,
                                                                  const T *__restrict__ gout, int64_t g_sb,
                                                                  int64_t g_sc, T *__restrict__ dx, int64_t d_sb,
                                                                  int64_t d_sc, int C, int N, int K, int CC, int plain) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned s_max[MR_THREADS / 64];
    const int b = blockIdx.y, tid = threadIdx.x;
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(smem);     // [CC*N] fixed-point scatter accumulator
    int *sidx = reinterpret_cast<int *>(acc + MRB_SLAB);                         // [K][N]
    const int nslab = (C + CC - 1) / CC;
    // this thread's pieces inside a slab: element offset pc * N + pn, node pn
    int off[MRB_ITEMS], pn[MRB_ITEMS], pc[MRB_ITEMS];
    {
        MR_WALK(4, tid, N, c, n);
#pragma unroll
        for (int it = 0; it < MRB_ITEMS; ++it) {
            pc[it] = c;
            pn[it] = n;
            off[it] = c * N + n;
            MR_NEXT(4, N, c, n);
        }
    }
    // running pointers of the pieces: slab s -> s + gridDim.x moves them by a constant
    const T *ge_p[MRB_ITEMS];
    T *dx_p[MRB_ITEMS];
    const unsigned char *ar_p[MRB_ITEMS];
    const int64_t cstep = (int64_t)gridDim.x * CC;
#pragma unroll
    for (int it = 0; it < MRB_ITEMS; ++it) {
        const int64_t c = (int64_t)blockIdx.x * CC + pc[it];
        ge_p[it] = gout + (size_t)b * g_sb + (size_t)(2 * c) * g_sc + pn[it];
        dx_p[it] = dx + (size_t)b * d_sb + (size_t)c * d_sc + pn[it];
        ar_p[it] = arg + ((size_t)b * C + c) * (N / 4) + pn[it] / 4;
    }
    const int64_t ge_step = 2 * cstep * g_sc, dx_step = cstep * d_sc, ar_step = cstep * (N / 4);
    // g_even, g_odd of the slabs in flight, as loaded (bf16: four slabs in the registers two took unpacked; see
    // mrconv_fwd_p_kernel -- the kernel runs at bytes in flight / latency)
    typedef typename MrRaw<T>::type Raw;
    constexpr int NPAR = MrRaw<T>::NPAR;
    Raw pe[NPAR][MRB_ITEMS], po[NPAR][MRB_ITEMS];
    unsigned pa[NPAR][MRB_ITEMS];
    auto fetch = [&](int par, int slab) {                          // slabs are fetched in order, each G after the last
        const int cc = min(CC, C - slab * CC);
#pragma unroll
        for (int it = 0; it < MRB_ITEMS; ++it) {
            if (pc[it] < cc) {
                pe[par][it] = GRAFP_LD_ONCE(16, reinterpret_cast<const Raw *>(ge_p[it]));
                po[par][it] = GRAFP_LD_ONCE(16, reinterpret_cast<const Raw *>(ge_p[it] + g_sc));
                pa[par][it] = *ar_p[it];
            }
            ge_p[it] += ge_step;
            ar_p[it] += ar_step;
        }
    };
    int slab = blockIdx.x;
    const int G = gridDim.x;
#pragma unroll
    for (int par = 0; par < NPAR; ++par)
        if (slab + par * G < nslab) fetch(par, slab + par * G);
    stage_idx<I>(sidx, idx + (size_t)b * N * K, N, K, tid);
    while (slab < nslab) {
#pragma unroll
        for (int par = 0; par < NPAR; ++par) {
            if (slab < nslab) {                    // workgroup-uniform
                const int cc = min(CC, C - slab * CC);
                __syncthreads();                   // previous slab written out (and the edges staged, first time round)
                float godd[MRB_ITEMS][4], base[MRB_ITEMS][4];
                unsigned ak[MRB_ITEMS];
                unsigned m = 0;
#pragma unroll
                for (int it = 0; it < MRB_ITEMS; ++it)
                    if (pc[it] < cc) {
                        ak[it] = pa[par][it];
                        *reinterpret_cast<uint4 *>(acc + off[it]) = make_uint4(0, 0, 0, 0);
                        *reinterpret_cast<uint4 *>(acc + off[it] + 2) = make_uint4(0, 0, 0, 0);
                        float ge[4];
                        mr_unpack4(pe[par][it], ge);
                        mr_unpack4(po[par][it], godd[it]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            base[it][e] = ge[e] - godd[it][e];               // identity branch minus the centre terms
                            m = max(m, __float_as_uint(godd[it][e]) & 0x7fffffffu);
                        }
                    }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
                if ((tid & 63) == 0) s_max[tid >> 6] = m;
                __syncthreads();
                if (slab + NPAR * G < nslab) fetch(par, slab + NPAR * G);
                m = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
                bool poisoned;
                float scale, inv_scale;
                mr_fix_scale(m, poisoned, scale, inv_scale);
                if (!poisoned) {
#pragma unroll
                    for (int it = 0; it < MRB_ITEMS; ++it)
                        if (pc[it] < cc) {
                            unsigned long long *arow = acc + (off[it] - pn[it]);
                            // the winner's edge of each of the 4 nodes: K vector reads, selected in registers
                            int bj[4] = {0, 0, 0, 0};
                            for (int k = 0; k < K; ++k) {
                                const int4 j4 = *reinterpret_cast<const int4 *>(sidx + k * N + pn[it]);
                                const int jj[4] = {j4.x, j4.y, j4.z, j4.w};
#pragma unroll
                                for (int e = 0; e < 4; ++e) bj[e] = ((ak[it] >> (2 * e)) & 3u) == (unsigned)k ? jj[e] : bj[e];
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) atomicAdd(arow + bj[e], mr_fix_encode(godd[it][e], scale));
                        }
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < MRB_ITEMS; ++it) {
                    if (pc[it] < cc) {
                        float v[4];
                        const uint4 a01 = *reinterpret_cast<const uint4 *>(acc + off[it]);
                        const uint4 a23 = *reinterpret_cast<const uint4 *>(acc + off[it] + 2);
                        const unsigned long long a[4] = {((unsigned long long)a01.y << 32) | a01.x,
                                                         ((unsigned long long)a01.w << 32) | a01.z,
                                                         ((unsigned long long)a23.y << 32) | a23.x,
                                                         ((unsigned long long)a23.w << 32) | a23.z};
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = poisoned ? NAN : base[it][e] + mr_fix_decode(a[e]) * inv_scale;
                        mr_st4(dx_p[it], v, plain != 0);
                    }
                    dx_p[it] += dx_step;
                }
                slab += G;
            }
        }
    }
}

// the store hint of the training-path kernels: a pure function of the bytes they write (see mr_st4)
static int mr_plain_stores(size_t bytes) {
    return bytes <= ((size_t)GRAFP_TUNE_INT("GRAFP_MR_PLAIN_MAX_MB", 140) << 20) ? 1 : 0;    // profiles/r06_e_mr_plain_threshold.txt
}

static int pick_cc(int C, int N, int target_elems) {
    int cc = target_elems / N;
    if (cc < 1) cc = 1;
    if (cc > C) cc = C;
    return cc;
}

}  // namespace grafp

// the shapes whose forward can record the arg-max and whose backward can run from it (pointer alignment aside)
static bool mrconv_arg_shape_ok(int dtype, int64_t x_sb, int64_t x_sc, int64_t o_sb, int64_t o_sc, int N, int K) {
    using namespace grafp;
    return (dtype == GRAFP_F32 || dtype == GRAFP_BF16) && N > 0 && K >= 1 && K <= 4 && N % 4 == 0 && x_sb % 4 == 0 &&
           x_sc % 4 == 0 && o_sb % 4 == 0 && o_sc % 4 == 0 && N <= MRB_SLAB && N <= MRP_SLAB &&
           ((size_t)MRP_SLAB + (size_t)K * N) * 4 <= 160 * 1024;
}

static int mrconv_fwd_impl(const void *x, int dtype, int64_t x_sb, int64_t x_sc, const void *idx, int idx32, int B,
                           int C, int N, int K, void *out, int64_t o_sb, int64_t o_sc, grafp_stream_t stream,
                           unsigned char *arg = nullptr) {
    using namespace grafp;
    GRAFP_REQUIRE(x && idx && out, "mrconv_fwd: null pointer");
    GRAFP_REQUIRE(B > 0 && C > 0 && N > 0 && K > 0, "mrconv_fwd: bad shape B=%d C=%d N=%d K=%d", B, C, N, K);
    GRAFP_REQUIRE(dtype == GRAFP_F32 || dtype == GRAFP_BF16, "mrconv_fwd: dtype %d not in {f32, bf16}", dtype);
#ifndef MR_FWD_ELEMS
#define MR_FWD_ELEMS 4096    // measured: 28 KB of LDS per workgroup (5 per CU) beats the 76 KB slab by 35 %
#endif
    const int CC = pick_cc(C, N, MR_FWD_ELEMS);
    const size_t lds = ((size_t)CC * N + (size_t)K * N) * 4;
    GRAFP_REQUIRE(lds <= 160 * 1024, "mrconv_fwd: N=%d K=%d needs %zu B of LDS (> 160 KiB)", N, K, lds);
    const dim3 grid((C + CC - 1) / CC, B);
    const size_t es = dtype == GRAFP_F32 ? 4 : 2;
    const bool v4 = (N % 4 == 0) && (x_sb % 4 == 0) && (x_sc % 4 == 0) && (o_sb % 4 == 0) && (o_sc % 4 == 0) &&
                    ((uintptr_t)x % (4 * es) == 0) && ((uintptr_t)out % (4 * es) == 0);
    if (v4 && N <= MRP_SLAB) {
        // persistent variant: whole channel rows per slab, a few slabs per workgroup
        const int ccp = MRP_SLAB / N < C ? MRP_SLAB / N : C;
        const int nslab = (C + ccp - 1) / ccp;
        int per_clip = nslab < 4 ? nslab : 4;                       // workgroups per clip
        while ((int64_t)per_clip * B < 1024 && per_clip < nslab) ++per_clip;
        const size_t ldsp = ((size_t)MRP_SLAB + (size_t)K * N) * 4;
        if (ldsp <= 160 * 1024) {
            const dim3 gridp(per_clip, B);
            const int plain = mr_plain_stores((size_t)2 * B * C * N * es);
#define MR_FWDP_W(T, I, WK)                                                                                            \
    (void)hipFuncSetAttribute((const void *)mrconv_fwd_p_kernel<T, I, WK>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)ldsp);                                                                              \
    hipLaunchKernelGGL((mrconv_fwd_p_kernel<T, I, WK>), gridp, dim3(MR_THREADS), ldsp, (hipStream_t)stream,            \
                       (const T *)x, x_sb, x_sc, (const I *)idx, (T *)out, o_sb, o_sc, C, N, K, ccp, arg, plain)
#define MR_FWDP(T, I)                                                                                                  \
    if (arg) { MR_FWDP_W(T, I, true); } else { MR_FWDP_W(T, I, false); }
            if (dtype == GRAFP_F32) { if (idx32) { MR_FWDP(float, int32_t); } else { MR_FWDP(float, int64_t); } }
            else { if (idx32) { MR_FWDP(unsigned short, int32_t); } else { MR_FWDP(unsigned short, int64_t); } }
#undef MR_FWDP
#undef MR_FWDP_W
            GRAFP_CHECK_LAUNCH("mrconv_fwd_p_kernel");
            return GRAFP_OK;
        }
    }
    GRAFP_REQUIRE(!arg, "mrconv_fwd_arg: shape / alignment outside grafp_mrconv_arg_supported");
#define MR_FWD_I(T, V, I)                                                                                              \
    (void)hipFuncSetAttribute((const void *)mrconv_fwd_kernel<T, V, I>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                              (int)lds);                                                                               \
    hipLaunchKernelGGL((mrconv_fwd_kernel<T, V, I>), grid, dim3(MR_THREADS), lds, (hipStream_t)stream, (const T *)x,   \
                       x_sb, x_sc, (const I *)idx, (T *)out, o_sb, o_sc, C, N, K, CC)
#define MR_FWD(T, V)                                                                                                   \
    if (idx32) { MR_FWD_I(T, V, int32_t); } else { MR_FWD_I(T, V, int64_t); }
    if (dtype == GRAFP_F32) { if (v4) { MR_FWD(float, 4); } else { MR_FWD(float, 1); } }
    else { if (v4) { MR_FWD(unsigned short, 4); } else { MR_FWD(unsigned short, 1); } }
#undef MR_FWD
#undef MR_FWD_I
    GRAFP_CHECK_LAUNCH("mrconv_fwd_kernel");
    return GRAFP_OK;
}

static int mrconv_bwd_impl(const void *x, int dtype, int64_t x_sb, int64_t x_sc, const void *idx, int idx32,
                           const void *grad_out, int64_t g_sb, int64_t g_sc, int B, int C, int N, int K, void *dx,
                           grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(x && idx && grad_out && dx, "mrconv_bwd: null pointer");
    GRAFP_REQUIRE(B > 0 && C > 0 && N > 0 && K > 0, "mrconv_bwd: bad shape B=%d C=%d N=%d K=%d", B, C, N, K);
    GRAFP_REQUIRE(dtype == GRAFP_F32 || dtype == GRAFP_BF16, "mrconv_bwd: dtype %d not in {f32, bf16}", dtype);
    const size_t es = dtype == GRAFP_F32 ? 4 : 2;
    const bool v4 = (N % 4 == 0) && (x_sb % 4 == 0) && (x_sc % 4 == 0) && (g_sb % 4 == 0) && (g_sc % 4 == 0) &&
                    ((uintptr_t)x % (4 * es) == 0) && ((uintptr_t)grad_out % (4 * es) == 0) && ((uintptr_t)dx % (4 * es) == 0);
    if (v4 && N <= MRB_SLAB) {
        const int ccp = MRB_SLAB / N < C ? MRB_SLAB / N : C;
        const int nslab = (C + ccp - 1) / ccp;
        int per_clip = nslab < 4 ? nslab : 4;
        while ((int64_t)per_clip * B < 1024 && per_clip < nslab) ++per_clip;
        const size_t ldsp = ((size_t)3 * MRB_SLAB + (size_t)K * N) * 4;      // i64 accumulator + f32 rows + edges
        if (ldsp <= 160 * 1024) {
            const dim3 gridp(per_clip, B);
#define MR_BWDP(T, I)                                                                                                  \
    (void)hipFuncSetAttribute((const void *)mrconv_bwd_p_kernel<T, I>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                              (int)ldsp);                                                                              \
    hipLaunchKernelGGL((mrconv_bwd_p_kernel<T, I>), gridp, dim3(MR_THREADS), ldsp, (hipStream_t)stream, (const T *)x,  \
                       x_sb, x_sc, (const I *)idx, (const T *)grad_out, g_sb, g_sc, (T *)dx, C, N, K, ccp)
            if (dtype == GRAFP_F32) { if (idx32) { MR_BWDP(float, int32_t); } else { MR_BWDP(float, int64_t); } }
            else { if (idx32) { MR_BWDP(unsigned short, int32_t); } else { MR_BWDP(unsigned short, int64_t); } }
#undef MR_BWDP
            GRAFP_CHECK_LAUNCH("mrconv_bwd_p_kernel");
            return GRAFP_OK;
        }
    }
    // a workgroup covers exactly ITEMS * 256 * V elements (whole channel rows)
    const int per_item = MR_THREADS * (v4 ? 4 : 1);
    const int items = N <= 2 * per_item ? 2 : 8;
    int CC = (items * per_item) / N;
    if (CC < 1) CC = 1;
    if (CC > C) CC = C;
    GRAFP_REQUIRE((size_t)CC * N <= (size_t)items * per_item,
                  "mrconv_bwd: N=%d exceeds the %d nodes a workgroup covers", N, items * per_item);
    // i64 accumulator + f32 rows + f32 identity term per slab element, + the clip's edges
    auto lds_of = [&](int cc_) { return ((size_t)4 * (((size_t)cc_ * N + 1) & ~(size_t)1) + (size_t)K * N) * 4; };
    while (CC > 1 && lds_of(CC) > 160 * 1024) --CC;
    const size_t lds = lds_of(CC);
    GRAFP_REQUIRE(lds <= 160 * 1024, "mrconv_bwd: N=%d K=%d needs %zu B of LDS (> 160 KiB)", N, K, lds);
    const dim3 grid((C + CC - 1) / CC, B);
#define MR_BWD_II(T, V, I, IT)                                                                                         \
    (void)hipFuncSetAttribute((const void *)mrconv_bwd_kernel<T, V, I, IT>,                                            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                   \
    hipLaunchKernelGGL((mrconv_bwd_kernel<T, V, I, IT>), grid, dim3(MR_THREADS), lds, (hipStream_t)stream,             \
                       (const T *)x, x_sb, x_sc, (const I *)idx, (const T *)grad_out, g_sb, g_sc, (T *)dx, C, N, K, CC)
#define MR_BWD_I(T, V, I)                                                                                              \
    if (items == 2) { MR_BWD_II(T, V, I, 2); } else { MR_BWD_II(T, V, I, 8); }
#define MR_BWD(T, V)                                                                                                   \
    if (idx32) { MR_BWD_I(T, V, int32_t); } else { MR_BWD_I(T, V, int64_t); }
    if (dtype == GRAFP_F32) { if (v4) { MR_BWD(float, 4); } else { MR_BWD(float, 1); } }
    else { if (v4) { MR_BWD(unsigned short, 4); } else { MR_BWD(unsigned short, 1); } }
#undef MR_BWD
#undef MR_BWD_I
#undef MR_BWD_II
    GRAFP_CHECK_LAUNCH("mrconv_bwd_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_mrconv_fwd_strided(const void *x, int dtype, int64_t x_sb, int64_t x_sc, const int64_t *idx, int B,
                                        int C, int N, int K, void *out, int64_t o_sb, int64_t o_sc,
                                        grafp_stream_t stream) {
    return mrconv_fwd_impl(x, dtype, x_sb, x_sc, idx, 0, B, C, N, K, out, o_sb, o_sc, stream);
}
extern "C" int grafp_mrconv_fwd_strided_i32(const void *x, int dtype, int64_t x_sb, int64_t x_sc, const int32_t *idx,
                                            int B, int C, int N, int K, void *out, int64_t o_sb, int64_t o_sc,
                                            grafp_stream_t stream) {
    return mrconv_fwd_impl(x, dtype, x_sb, x_sc, idx, 1, B, C, N, K, out, o_sb, o_sc, stream);
}
extern "C" int grafp_mrconv_bwd_strided(const void *x, int dtype, int64_t x_sb, int64_t x_sc, const int64_t *idx,
                                        const void *grad_out, int64_t g_sb, int64_t g_sc, int B, int C, int N, int K,
                                        void *dx, grafp_stream_t stream) {
    return mrconv_bwd_impl(x, dtype, x_sb, x_sc, idx, 0, grad_out, g_sb, g_sc, B, C, N, K, dx, stream);
}
extern "C" int grafp_mrconv_bwd_strided_i32(const void *x, int dtype, int64_t x_sb, int64_t x_sc, const int32_t *idx,
                                            const void *grad_out, int64_t g_sb, int64_t g_sc, int B, int C, int N,
                                            int K, void *dx, grafp_stream_t stream) {
    return mrconv_bwd_impl(x, dtype, x_sb, x_sc, idx, 1, grad_out, g_sb, g_sc, B, C, N, K, dx, stream);
}

extern "C" int grafp_mrconv_arg_supported(int dtype, int64_t x_sb, int64_t x_sc, int64_t o_sb, int64_t o_sc, int N, int K) {
    return mrconv_arg_shape_ok(dtype, x_sb, x_sc, o_sb, o_sc, N, K) ? 1 : 0;
}
extern "C" int grafp_mrconv_fwd_arg(const void *x, int dtype, int64_t x_sb, int64_t x_sc, const void *idx, int idx_is_i32,
                                    int B, int C, int N, int K, void *out, int64_t o_sb, int64_t o_sc, uint8_t *arg,
                                    grafp_stream_t stream) {
    GRAFP_REQUIRE(arg, "mrconv_fwd_arg: null pointer");
    GRAFP_REQUIRE(mrconv_arg_shape_ok(dtype, x_sb, x_sc, o_sb, o_sc, N, K),
                  "mrconv_fwd_arg: shape outside grafp_mrconv_arg_supported (N=%d K=%d)", N, K);
    return mrconv_fwd_impl(x, dtype, x_sb, x_sc, idx, idx_is_i32, B, C, N, K, out, o_sb, o_sc, stream, arg);
}
extern "C" int grafp_mrconv_bwd_arg(const uint8_t *arg, int dtype, const void *idx, int idx_is_i32, const void *grad_out,
                                    int64_t g_sb, int64_t g_sc, int B, int C, int N, int K, void *dx, int64_t d_sb,
                                    int64_t d_sc, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(arg && idx && grad_out && dx, "mrconv_bwd_arg: null pointer");
    GRAFP_REQUIRE(B > 0 && C > 0, "mrconv_bwd_arg: bad shape B=%d C=%d", B, C);
    const size_t es = dtype == GRAFP_F32 ? 4 : 2;
    GRAFP_REQUIRE(mrconv_arg_shape_ok(dtype, d_sb, d_sc, g_sb, g_sc, N, K) && (uintptr_t)grad_out % (4 * es) == 0 &&
                      (uintptr_t)dx % (4 * es) == 0,
                  "mrconv_bwd_arg: shape / alignment outside grafp_mrconv_arg_supported (N=%d K=%d)", N, K);
    const int ccp = MRB_SLAB / N < C ? MRB_SLAB / N : C;
    const int nslab = (C + ccp - 1) / ccp;
    int per_clip = nslab < 4 ? nslab : 4;
    while ((int64_t)per_clip * B < 1024 && per_clip < nslab) ++per_clip;
    const size_t ldsp = (size_t)8 * MRB_SLAB + (size_t)4 * K * N;             // i64 accumulator + edges [K][N]
    const dim3 gridp(per_clip, B);
    const int plain = mr_plain_stores((size_t)B * C * N * es);
#define MR_BWDA(T, I)                                                                                                  \
    (void)hipFuncSetAttribute((const void *)mrconv_bwd_a_kernel<T, I>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                              (int)ldsp);                                                                              \
    hipLaunchKernelGGL((mrconv_bwd_a_kernel<T, I>), gridp, dim3(MR_THREADS), ldsp, (hipStream_t)stream, arg,           \
                       (const I *)idx, (const T *)grad_out, g_sb, g_sc, (T *)dx, d_sb, d_sc, C, N, K, ccp, plain)
    if (dtype == GRAFP_F32) { if (idx_is_i32) { MR_BWDA(float, int32_t); } else { MR_BWDA(float, int64_t); } }
    else { if (idx_is_i32) { MR_BWDA(unsigned short, int32_t); } else { MR_BWDA(unsigned short, int64_t); } }
#undef MR_BWDA
    GRAFP_CHECK_LAUNCH("mrconv_bwd_a_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_mrconv_fwd_f32(const float *x, const int64_t *idx, int B, int C, int N, int K, float *out,
                                    grafp_stream_t stream) {
    return grafp_mrconv_fwd_strided(x, GRAFP_F32, (int64_t)C * N, N, idx, B, C, N, K, out, (int64_t)2 * C * N, N, stream);
}

extern "C" int grafp_mrconv_bwd_f32(const float *x, const int64_t *idx, const float *grad_out, int B, int C, int N,
                                    int K, float *dx, grafp_stream_t stream) {
    return grafp_mrconv_bwd_strided(x, GRAFP_F32, (int64_t)C * N, N, idx, grad_out, (int64_t)2 * C * N, N, B, C, N, K, dx,
                                    stream);
}
