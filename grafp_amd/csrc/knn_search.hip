// knn_search.hip -- brute-force fingerprint search (K13 of SURVEY.md section 2a), gfx950.
//
// Replaces faiss.IndexFlatL2 add/search as used at /root/reference/eval.py:54,212-213,269-270.
// The database stays resident in HBM as (n,128) f32 plus one squared norm per row (the `add` step).
// A workgroup (4 waves, 1 per SIMD) owns one contiguous slice of rows and one group of queries.  The 4
// waves are arranged as QW query-waves x RW row-waves (1x4 for <= 32 queries, 2x2 for <= 64, else 4x1):
//   - the query operand (32 queries x 128 dims) lives in 64 VGPRs per lane for the whole kernel;
//   - database rows stream HBM -> registers (prefetched one tile ahead) -> LDS -> exact-f32 MFMA
//     (v_mfma_f32_32x32x2_f32, a c-ordered fmaf chain = the order oracle/csrc/flat_search.c fixes);
//   - dis = (qq + dd) - 2*ip, clamped at 0; a lane keeps a candidate only if dis <= the query's current
//     k-th best; survivors go to a per-query LDS queue and are folded into the sorted top list by a
//     64-lane bitonic sort once 17+ have accumulated (amortised ~8 VALU per survivor).
// Partial lists (one per row slice) are merged by search_merge_kernel (also grafp_merge_topk for
// per-GPU shards).  Ordering everywhere is (distance, id) lexicographic => lowest id wins ties.
//
// Roofline: one pass streams n*(512+4) bytes; 2*128 flops per (row, query).  HBM-bound up to ~50
// queries per pass, f32-matrix-bound (157.3 TFLOP/s) beyond.  See DESIGN.md "search_partial_kernel".
#include <math.h>

#include "common.h"

namespace grafp {

constexpr int SR_D = 128;
constexpr int SR_LS = 129;       // LDS row stride of the database tile (bank spread for ds_read_b32)
constexpr int SR_QS = 65;        // slot stride per query in the selection arrays
constexpr int SR_EMPTY = 0x7fffffff;
constexpr int SR_TRIGGER = 16;   // fold the queue when more than this many survivors are pending

#define WAVE_SYNC()                                                   \
    do {                                                              \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");        \
        __builtin_amdgcn_wave_barrier();                              \
    } while (0)

template <typename I>
__device__ __forceinline__ bool lex_lt(float d1, I i1, float d2, I i2) {
    return d1 < d2 || (d1 == d2 && i1 < i2);
}

__device__ __forceinline__ long long shfl_xor_i(long long v, int m) {
    int lo = __shfl_xor((int)(v & 0xffffffffll), m), hi = __shfl_xor((int)(v >> 32), m);
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ int shfl_xor_i(int v, int m) { return __shfl_xor(v, m); }

// 64-lane bitonic sort, one (d, i) pair per lane, ascending by (d, i)
template <typename I>
__device__ __forceinline__ void wave_sort64(float &d, I &i, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const float od = __shfl_xor(d, j);
            const I oi = shfl_xor_i(i, j);
            const bool want_min = ((lane & j) == 0) == ((lane & k) == 0);
            const bool take = want_min ? lex_lt(od, oi, d, i) : lex_lt(d, i, od, oi);
            d = take ? od : d;
            i = take ? oi : i;
        }
    }
}

// Per-wave selection state in LDS: 32 queries x (32 sorted best | 32 pending), counts, thresholds.
struct Sel {
    float *sd;
    int *si;
    int *cnt;
    float *thr;
    __device__ __forceinline__ void bind(float *base) {
        sd = base;
        si = reinterpret_cast<int *>(base + 32 * SR_QS);
        cnt = si + 32 * SR_QS;
        thr = reinterpret_cast<float *>(cnt + 32);
    }
    static constexpr int kFloats = 2 * 32 * SR_QS + 64;
    __device__ __forceinline__ void init(int lane) {
        for (int j = lane; j < 32 * SR_QS; j += 64) {
            sd[j] = INFINITY;
            si[j] = SR_EMPTY;
        }
        if (lane < 32) {
            cnt[lane] = 0;
            thr[lane] = INFINITY;
        }
    }
    // fold query q's pending entries into its sorted list (whole wave cooperates)
    __device__ __forceinline__ void fold(int q, int k, int lane) {
        const int c = cnt[q];
        float d = INFINITY;
        int i = SR_EMPTY;
        if (lane < 32 || lane - 32 < c) {
            d = sd[q * SR_QS + lane];
            i = si[q * SR_QS + lane];
        }
        wave_sort64(d, i, lane);
        if (lane < 32) {
            sd[q * SR_QS + lane] = d;
            si[q * SR_QS + lane] = i;
        }
        const float t = __shfl(d, k - 1);
        if (lane == 0) {
            cnt[q] = 0;
            thr[q] = t;
        }
    }
    __device__ __forceinline__ void fold_where(bool need, int k, int lane) {
        unsigned long long mask = __ballot(need && lane < 32);
        while (mask) {
            const int q = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            fold(q, k, lane);
        }
    }
};

// ---- squared row norms (the `index.add` step); thread j chains over row j out of an LDS tile ------
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float *__restrict__ m, int64_t n, int d,
                                                         float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *tile = reinterpret_cast<float *>(smem);  // [256][d+1]
    const int ls = d + 1, tid = threadIdx.x;
    for (int64_t r0 = (int64_t)blockIdx.x * 256; r0 < n; r0 += (int64_t)gridDim.x * 256) {
        const int rows = (int)((n - r0) < 256 ? (n - r0) : 256);
        __syncthreads();
        const float *src = m + r0 * d;
        for (int i = tid; i < rows * d; i += 256) {
            const int r = i / d, c = i - r * d;
            tile[r * ls + c] = src[i];
        }
        __syncthreads();
        if (tid < rows) {
            float s = 0.0f;
            const float *row = tile + tid * ls;
            for (int c = 0; c < d; ++c) s = __builtin_fmaf(row[c], row[c], s);
            out[r0 + tid] = s;
        }
    }
}

// ---- main pass -----------------------------------------------------------------------------------
template <int QW>
__global__ __launch_bounds__(256, 1) void search_partial_kernel(const float *__restrict__ db,
                                                                const float *__restrict__ dd, int64_t n,
                                                                const float *__restrict__ q,
                                                                const float *__restrict__ qq, int nq, int k,
                                                                int64_t rows_per_split, int64_t id_base,
                                                                float *__restrict__ part_d,
                                                                int64_t *__restrict__ part_i) {
    constexpr int RW = 4 / QW;
    constexpr int TROWS = 32 * RW;
    constexpr int NV = TROWS * 32 / 256;  // float4 per thread per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *tile = reinterpret_cast<float *>(smem);  // [TROWS][SR_LS]
    float *sDD = tile + TROWS * SR_LS;              // [TROWS]
    float *selbase = sDD + TROWS;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int qw = wave % QW, rw = wave / QW;
    Sel sel;
    sel.bind(selbase + wave * Sel::kFloats);
    sel.init(lane);

    const int split = blockIdx.x;
    const int qbase = (blockIdx.y * QW + qw) * 32;
    const int64_t row_begin = (int64_t)split * rows_per_split;
    const int64_t row_end = (row_begin + rows_per_split < n) ? row_begin + rows_per_split : n;
    const int qi = qbase + l31;
    const bool qvalid = qi < nq;

    float bq[64];  // query operand: B[k = 2s + half][j = l31]
    {
        const float *qrow = q + (size_t)(qvalid ? qi : 0) * SR_D + half;
#pragma unroll
        for (int s = 0; s < 64; ++s) bq[s] = qvalid ? qrow[2 * s] : 0.0f;
    }
    const float myqq = qvalid ? qq[qi] : 0.0f;

    const int ntiles = (int)((row_end - row_begin + TROWS - 1) / TROWS);
    float4 pf[NV];
    const float4 *db4 = reinterpret_cast<const float4 *>(db);
    auto prefetch = [&](int t) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int item = tid + v * 256;
            const int64_t grow = row_begin + (int64_t)t * TROWS + (item >> 5);
            pf[v] = grow < row_end ? db4[grow * 32 + (item & 31)] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (ntiles > 0) prefetch(0);
    WAVE_SYNC();

    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();  // every wave is done reading the previous tile
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int item = tid + v * 256;
            float *o = tile + (item >> 5) * SR_LS + (item & 31) * 4;
            o[0] = pf[v].x; o[1] = pf[v].y; o[2] = pf[v].z; o[3] = pf[v].w;
        }
        if (tid < TROWS) {
            const int64_t grow = row_begin + (int64_t)t * TROWS + tid;
            sDD[tid] = grow < row_end ? dd[grow] : INFINITY;
        }
        __syncthreads();
        if (t + 1 < ntiles) prefetch(t + 1);  // in flight while this tile is consumed

        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        {
            const float *arow = tile + (rw * 32 + l31) * SR_LS + half;
#pragma unroll
            for (int s = 0; s < 64; ++s) acc = mfma32x32x2(arow[2 * s], bq[s], acc);
        }
        // selection: lane holds 16 distances of query l31 against rows mfma_row(r, half) of its 32-row slab
        float thr_q = sel.thr[l31];
        const int64_t slab0 = row_begin + (int64_t)t * TROWS + rw * 32;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int r = ph * 8 + r8;
                const int lrow = mfma_row(r, half);
                const int64_t grow = slab0 + lrow;
                float dis = (myqq + sDD[rw * 32 + lrow]) - 2.0f * acc[r];
                dis = dis < 0.0f ? 0.0f : dis;
                if (qvalid && grow < row_end && dis <= thr_q) {
                    const int pos = atomicAdd(&sel.cnt[l31], 1);
                    sel.sd[l31 * SR_QS + 32 + pos] = dis;
                    sel.si[l31 * SR_QS + 32 + pos] = (int)grow;
                }
            }
            WAVE_SYNC();
            sel.fold_where(sel.cnt[l31] > SR_TRIGGER, k, lane);
            WAVE_SYNC();
            thr_q = sel.thr[l31];
        }
    }
    WAVE_SYNC();
    sel.fold_where(sel.cnt[l31] > 0, k, lane);
    __syncthreads();
    // fold the other row-waves' lists into row-wave 0's
    if (rw == 0) {
        for (int orw = 1; orw < RW; ++orw) {
            Sel oth;
            oth.bind(selbase + (orw * QW + qw) * Sel::kFloats);
            for (int qs = 0; qs < 32; ++qs) {
                float d;
                int i;
                if (lane < 32) {
                    d = sel.sd[qs * SR_QS + lane];
                    i = sel.si[qs * SR_QS + lane];
                } else {
                    d = oth.sd[qs * SR_QS + lane - 32];
                    i = oth.si[qs * SR_QS + lane - 32];
                }
                wave_sort64(d, i, lane);
                if (lane < 32) {
                    sel.sd[qs * SR_QS + lane] = d;
                    sel.si[qs * SR_QS + lane] = i;
                }
            }
            WAVE_SYNC();
        }
        for (int qs = 0; qs < 32; ++qs) {
            const int qo = qbase + qs;
            if (qo < nq && lane < k) {
                const size_t o = ((size_t)split * nq + qo) * k + lane;
                const int i = sel.si[qs * SR_QS + lane];
                part_d[o] = sel.sd[qs * SR_QS + lane];
                part_i[o] = i == SR_EMPTY ? (int64_t)-1 : id_base + (int64_t)i;
            }
        }
    }
}

// ---- merge of P sorted partial lists per query (one wave per query) -----------------------------
__global__ __launch_bounds__(64) void search_merge_kernel(const float *__restrict__ part_d,
                                                          const int64_t *__restrict__ part_i, int P, int nq, int k,
                                                          float *__restrict__ out_d, int64_t *__restrict__ out_i) {
    __shared__ float sd[64];
    __shared__ long long si[64];
    const int q = blockIdx.x, lane = threadIdx.x;
    constexpr long long EMPTY = 0x7fffffffffffffffll;
    // every part's k-th best bounds the global k-th best from above
    float thr = INFINITY;
    for (int p = lane; p < P; p += 64) {
        const size_t o = ((size_t)p * nq + q) * k + (k - 1);
        if (part_i[o] >= 0) thr = fminf(thr, part_d[o]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) thr = fminf(thr, __shfl_xor(thr, o));

    sd[lane] = INFINITY;
    si[lane] = EMPTY;
    int cnt = 0;  // wave-uniform
    const int total = P * k;
    WAVE_SYNC();
    auto fold = [&]() {
        float d = sd[lane];
        long long i = si[lane];
        if (lane >= 32 + cnt) { d = INFINITY; i = EMPTY; }
        wave_sort64(d, i, lane);
        WAVE_SYNC();
        sd[lane] = lane < 32 ? d : INFINITY;
        si[lane] = lane < 32 ? i : EMPTY;
        const float t = __shfl(d, k - 1);
        thr = fminf(thr, t);
        cnt = 0;
        WAVE_SYNC();
    };
    for (int base = 0; base < total; base += 32) {
        const int e = base + lane;
        bool pass = false;
        float d = INFINITY;
        long long i = EMPTY;
        if (lane < 32 && e < total) {
            const int p = e / k, j = e - p * k;
            const size_t o = ((size_t)p * nq + q) * k + j;
            i = part_i[o];
            d = part_d[o];
            pass = i >= 0 && d <= thr;
        }
        const unsigned long long mask = __ballot(pass);
        const int add = __popcll(mask);
        if (add == 0) continue;
        if (cnt + add > 32) fold();
        if (pass) {
            const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
            sd[32 + pos] = d;
            si[32 + pos] = i;
        }
        cnt += add;
        WAVE_SYNC();
    }
    if (cnt > 0) fold();
    if (lane < k) {
        const long long i = si[lane];
        out_d[(size_t)q * k + lane] = sd[lane];
        out_i[(size_t)q * k + lane] = i == EMPTY ? (int64_t)-1 : (int64_t)i;
    }
}

struct SearchPlan {
    int qw, qgroups, splits, trows;
    int64_t rows_per_split;
};

static SearchPlan make_plan(int64_t n, int nq) {
    SearchPlan p;
    p.qw = nq <= 32 ? 1 : (nq <= 64 ? 2 : 4);
    p.trows = 32 * (4 / p.qw);
    p.qgroups = (nq + 32 * p.qw - 1) / (32 * p.qw);
    int64_t splits = 256 / p.qgroups;  // one workgroup per CU when the query groups allow it
    if (splits < 1) splits = 1;
    const int64_t max_splits = n / (4 * p.trows) > 1 ? n / (4 * p.trows) : 1;
    if (splits > max_splits) splits = max_splits;
    int64_t rps = (n + splits - 1) / splits;
    rps = (rps + p.trows - 1) / p.trows * p.trows;
    if (rps < p.trows) rps = p.trows;
    p.rows_per_split = rps;
    p.splits = (int)((n + rps - 1) / rps);
    if (p.splits < 1) p.splits = 1;
    return p;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace grafp

extern "C" int grafp_row_sqnorm_f32(const float *m, int64_t n, int d, float *out, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(m && out, "row_sqnorm: null pointer");
    GRAFP_REQUIRE(n >= 0 && d > 0 && d <= 152, "row_sqnorm: bad n=%lld d=%d (d <= 152)", (long long)n, d);
    if (n == 0) return GRAFP_OK;
    const size_t lds = (size_t)256 * (d + 1) * sizeof(float);
    (void)hipFuncSetAttribute((const void *)row_sqnorm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), lds, (hipStream_t)stream,
                       m, n, d, out);
    GRAFP_CHECK_LAUNCH("row_sqnorm_kernel");
    return GRAFP_OK;
}

extern "C" size_t grafp_knn_search_workspace(int64_t n, int nq, int d, int k) {
    using namespace grafp;
    if (n <= 0 || nq <= 0 || d != SR_D || k < 1) return 0;
    const SearchPlan p = make_plan(n, nq);
    return align256((size_t)nq * sizeof(float)) + align256((size_t)p.splits * nq * k * sizeof(float)) +
           align256((size_t)p.splits * nq * k * sizeof(int64_t));
}

extern "C" int grafp_knn_search_l2_f32(const float *db, const float *db_sqnorm, int64_t n, const float *q, int nq,
                                       int d, int k, int64_t id_base, float *out_dist, int64_t *out_ids, void *ws,
                                       size_t ws_bytes, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(db && db_sqnorm && q && out_dist && out_ids, "knn_search: null pointer");
    GRAFP_REQUIRE(d == SR_D, "knn_search: d=%d unsupported (fingerprints are 128-d)", d);
    GRAFP_REQUIRE(n >= 1 && n < 0x7fffffffll && nq >= 1, "knn_search: bad n=%lld nq=%d", (long long)n, nq);
    GRAFP_REQUIRE(k >= 1 && k <= GRAFP_SEARCH_MAX_K, "knn_search: k=%d not in [1, %d]", k, GRAFP_SEARCH_MAX_K);
    GRAFP_REQUIRE(((uintptr_t)db & 15) == 0 && ((uintptr_t)q & 3) == 0, "knn_search: db must be 16-byte aligned");
    const size_t need = grafp_knn_search_workspace(n, nq, d, k);
    if (!ws || ws_bytes < need) {
        set_error("knn_search: workspace %zu bytes < required %zu", ws_bytes, need);
        return GRAFP_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const SearchPlan p = make_plan(n, nq);
    float *qq = (float *)ws;
    float *part_d = (float *)((char *)ws + align256((size_t)nq * sizeof(float)));
    int64_t *part_i = (int64_t *)((char *)part_d + align256((size_t)p.splits * nq * k * sizeof(float)));
    int rc = grafp_row_sqnorm_f32(q, nq, d, qq, stream);
    if (rc != GRAFP_OK) return rc;
    const size_t lds = ((size_t)p.trows * SR_LS + p.trows + 4 * Sel::kFloats) * sizeof(float);
    const dim3 grid(p.splits, p.qgroups);
#define SR_LAUNCH(QW)                                                                                               \
    (void)hipFuncSetAttribute((const void *)search_partial_kernel<QW>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                              (int)lds);                                                                            \
    hipLaunchKernelGGL(search_partial_kernel<QW>, grid, dim3(256), lds, s, db, db_sqnorm, n, q, qq, nq, k,          \
                       p.rows_per_split, id_base, part_d, part_i)
    if (p.qw == 1) { SR_LAUNCH(1); }
    else if (p.qw == 2) { SR_LAUNCH(2); }
    else { SR_LAUNCH(4); }
#undef SR_LAUNCH
    GRAFP_CHECK_LAUNCH("search_partial_kernel");
    hipLaunchKernelGGL(search_merge_kernel, dim3(nq), dim3(64), 0, s, part_d, part_i, p.splits, nq, k, out_dist,
                       out_ids);
    GRAFP_CHECK_LAUNCH("search_merge_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_merge_topk(const float *part_dist, const int64_t *part_ids, int P, int nq, int k, float *out_dist,
                                int64_t *out_ids, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(part_dist && part_ids && out_dist && out_ids, "merge_topk: null pointer");
    GRAFP_REQUIRE(P >= 1 && nq >= 1 && k >= 1 && k <= GRAFP_SEARCH_MAX_K, "merge_topk: bad P=%d nq=%d k=%d", P, nq, k);
    hipLaunchKernelGGL(search_merge_kernel, dim3(nq), dim3(64), 0, (hipStream_t)stream, part_dist, part_ids, P, nq, k,
                       out_dist, out_ids);
    GRAFP_CHECK_LAUNCH("search_merge_kernel");
    return GRAFP_OK;
}
