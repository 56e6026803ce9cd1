// knn_search.hip -- brute-force fingerprint search (K13 of SURVEY.md section 2a), gfx950.
//
// Replaces faiss.IndexFlatL2 add/search as used at /root/reference/eval.py:54,212-213,269-270.
// The database stays resident in HBM as (n,128) f32 plus one squared norm per row (the `add` step).
// A search is four small-to-large launches, none of which keeps per-query sorted state while it streams:
//   1. search_bound_kernel  over a sample (first max(64k, n/16) rows): every lane keeps the minimum distance it
//      sees; lanes fall into 64 groups with DISJOINT row sets, one integer atomicMin per lane at the end.
//   2. search_thr_kernel    per query, the k-th smallest of the 64 group minima: at least k rows are <= it, so it
//      bounds the k-th best distance from above (typically ~25 x n/sample rows fall under it).
//   3. search_scan_kernel   the one pass over the whole database: HBM -> registers (one tile ahead) -> LDS ->
//      exact-f32 MFMA (v_mfma_f32_32x32x2_f32 = the k-ordered fmaf chain oracle/csrc/flat_search.c fixes),
//      dis = (qq + dd) - 2*ip clamped at 0, and a lane appends (dis, row) to its query's candidate list only when
//      dis <= the bound -- 16 compares and one ballot per 32x32 block in the common case.  No selection state in
//      LDS, so two workgroups share a CU and one's MFMA chain hides the other's loads, stores and barriers.
//   4. search_select_kernel one workgroup per query sorts its few hundred candidates by (distance, id) with
//      64-lane bitonic folds.  A query whose list overflowed (pathological data: thousands of exact ties) is
//      rescanned exactly by that workgroup with the same arithmetic -- slow, never wrong.
// A workgroup is 4 waves arranged as QW query-waves x RW row-waves (1x4 for <= 32 queries, 2x2 for <= 64,
// else 4x1); the query operand (32 queries x 128 dims) lives in 64 VGPRs per lane for the whole kernel.
// Ordering everywhere is (distance, id) lexicographic => lowest id wins ties.  search_merge_kernel
// (grafp_merge_topk) merges per-GPU shard results.
//
// Roofline: one pass streams n*(512+4) bytes; 2*128 flops per (row, query).  HBM-bound up to ~50
// queries per pass, f32-matrix-bound (157.3 TFLOP/s) beyond.  See DESIGN.md "search_scan_kernel".
// The default entry (grafp_knn_search_l2_pre, second half of this file) runs the same pipeline on a bf16 copy of the
// database with a rigorous rounding margin and rescoring in exact f32: same bits out, 2-8x faster.  Its launches
// (round 4): search_bound_bf16_kernel (pre-pass, raw group maxima, no atomics) -> search_thr_pre_kernel (query norms,
// empty lists, bound) -> search_scan_bf16_kernel (LDS-DMA ring, bf16 MFMA, one compare per 32 x 32 block; from 768
// queries on in two parts with search_select_exact_kernel<true> tightening the bound in between; from 1024 queries two
// query sets per wave) -> search_select_exact_kernel<false> (histogram bound, exact f32 distances of the ~100 rows
// under it, rank by counting).
#include <math.h>

#include <type_traits>

#include "common.h"
#include "dma_ring.h"
#include "topk.h"
#include "tuning.h"

namespace grafp {

constexpr int SR_D = 128;
constexpr int SR_LS = 129;       // LDS row stride of the database tile (ds_read_b32 banks are mod 32)
constexpr int SR_GROUPS = 64;    // disjoint sample groups per query in the pre-pass
constexpr int SR_CAP = 4096;     // candidate slots per query (typical fill: a few hundred)
constexpr int SR_NSUB = 16;      // pre-filter path: the slots are SR_NSUB sub-lists with their own counters, picked by the
                                 // row slice of the appending workgroup -- hundreds of increments on ONE counter serialise
                                 // in L2 (a small batch spent 40 us of a 90 us scan there)
constexpr int SR_SUBCAP = SR_CAP / SR_NSUB;

// (database slice, query group) of this workgroup.  The grid is (slices, query groups); workgroups go to the 8 XCDs
// round-robin in dispatch order (x fastest), so without a remap the query groups that stream the SAME slice sit on
// different XCDs and every one of them pulls the slice through its own L2: at nq = 4096 the 256 MB bf16 copy was read 32
// times from the memory side (5 TB/s, the bound of that launch).  With the remap an XCD owns whole slices: all query
// groups of a slice run side by side on one L2 and the database crosses the fabric about once.
__device__ __forceinline__ void search_block(int &split, int &qgroup) {
    const int total = (int)(gridDim.x * gridDim.y);
    const int v = xcd_remap((int)(blockIdx.x + gridDim.x * blockIdx.y), total);
    split = v / (int)gridDim.y;
    qgroup = v - split * (int)gridDim.y;
}

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

// ---- the streaming loop shared by the pre-pass and the main pass -------------------------------------------------
// Tiles of TROWS rows go HBM -> registers (tile t+1 is in flight while tile t is consumed) -> LDS; wave (qw, rw)
// multiplies rows [rw*32, rw*32+32) of the tile with its 32 queries and hands the 32x32 accumulator block to
// `on_block(t, acc)`: lane (l31, half) holds query l31 x rows mfma_row(r, half), r = 0..15.
template <int QW, typename F>
__device__ __forceinline__ void stream_tiles(const float *__restrict__ db, const float *__restrict__ dd,
                                             int64_t row_begin, int64_t row_end, float *tile, float *sDD,
                                             const float (&bq)[64], F &&on_block) {
    constexpr int RW = 4 / QW;
    constexpr int TROWS = 32 * RW;
    constexpr int NV = TROWS * 32 / 256;  // float4 per thread per tile
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int rw = wave / QW;
    const int ntiles = row_end > row_begin ? (int)((row_end - row_begin + TROWS - 1) / TROWS) : 0;
    const int nrows = (int)(row_end - row_begin);
    const float4 *base4 = reinterpret_cast<const float4 *>(db) + row_begin * 32;
    const float *ddb = dd + row_begin;
    float4 pf[NV];
    float pdd = 0.0f;
    auto prefetch = [&](int t) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int item = tid + v * 256;
            int lr = t * TROWS + (item >> 5);
            lr = lr < nrows ? lr : nrows - 1;          // rows past the end: any readable data, their norm is NaN
            pf[v] = base4[(size_t)lr * 32 + (item & 31)];
        }
        if (tid < TROWS) {
            const int lr = t * TROWS + tid;
            pdd = lr < nrows ? ddb[lr] : __builtin_nanf("");   // NaN fails every comparison downstream
        }
    };
    if (ntiles > 0) prefetch(0);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();  // every wave is done reading the previous tile
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int item = tid + v * 256;
            float *o = tile + (item >> 5) * SR_LS + (item & 31) * 4;
            o[0] = pf[v].x; o[1] = pf[v].y; o[2] = pf[v].z; o[3] = pf[v].w;
        }
        if (tid < TROWS) sDD[tid] = pdd;
        __syncthreads();
        if (t + 1 < ntiles) prefetch(t + 1);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        // A operand: AB LDS reads are issued a batch ahead of the AB dependent MFMAs that consume them (the
        // scheduler barriers keep the compiler from sinking each read next to its use and exposing LDS latency)
        const float *arow = tile + (rw * 32 + l31) * SR_LS + half;
        constexpr int AB = QW == 1 ? 8 : 16;   // batch depth (register budget: the 1x4 shape prefetches 64 VGPRs)
        float a[2][AB];
#pragma unroll
        for (int i = 0; i < AB; ++i) a[0][i] = arow[2 * i];
#pragma unroll
        for (int b = 0; b < 64 / AB; ++b) {
            if (b + 1 < 64 / AB) {
#pragma unroll
                for (int i = 0; i < AB; ++i) a[(b + 1) & 1][i] = arow[2 * (AB * (b + 1) + i)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < AB; ++i) acc = mfma32x32x2(a[b & 1][i], bq[AB * b + i], acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        on_block(t, acc);
    }
}

// query operand of the MFMA: B[k = 2s + half][j = l31]
__device__ __forceinline__ void load_queries(const float *__restrict__ q, int qi, bool qvalid, int half,
                                             float (&bq)[64]) {
    // row 0 stands in for a lane without a query: the loads are unconditional (64 in flight behind ONE wait -- written
    // as `qvalid ? load : 0` every element became its own branch, load and vmcnt(0): 64 serial round trips, a fifth of a
    // small-batch launch) and the value is dropped afterwards
    const float *qrow = q + (size_t)(qvalid ? qi : 0) * SR_D + half;
#pragma unroll
    for (int s = 0; s < 64; ++s) bq[s] = qrow[2 * s];
#pragma unroll
    for (int s = 0; s < 64; ++s) bq[s] = qvalid ? bq[s] : 0.0f;
}

// ---- launch 0: query norms, empty group minima, empty candidate lists ---------------------------------------------
__global__ __launch_bounds__(256) void search_init_kernel(const float *__restrict__ q, int nq, float *__restrict__ qq,
                                                          int *__restrict__ gmin, int *__restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (int64_t)nq * SR_GROUPS) gmin[i] = 0x7f800000;  // +inf
    if (i < (int64_t)nq * SR_NSUB) cnt[i] = 0;          // (the f32 path only uses the first nq)
    if (i < nq) {
        const float *row = q + i * SR_D;
        float s = 0.0f;
        for (int c = 0; c < SR_D; ++c) s = __builtin_fmaf(row[c], row[c], s);   // same chain as row_sqnorm_kernel
        qq[i] = s;
    }
}

// ---- launch 1: pre-pass over the sample --------------------------------------------------------------------------
// Distances are >= 0, so their f32 bit patterns order like ints and the group minimum is one integer atomicMin.
template <int QW>
__global__ __launch_bounds__(256, 2) void search_bound_kernel(const float *__restrict__ db,
                                                              const float *__restrict__ dd, int64_t n_sample,
                                                              const float *__restrict__ q,
                                                              const float *__restrict__ qq, int nq,
                                                              int64_t rows_per_split, int *__restrict__ gmin) {
    constexpr int RW = 4 / QW;
    constexpr int TROWS = 32 * RW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *tile = reinterpret_cast<float *>(smem);  // [TROWS][SR_LS]
    float *sDD = tile + TROWS * SR_LS;              // [TROWS]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int qw = wave % QW, rw = wave / QW;
    int split, qgroup;
    search_block(split, qgroup);
    const int qi = (qgroup * QW + qw) * 32 + l31;
    const bool qvalid = qi < nq;
    const int64_t row_begin = (int64_t)split * rows_per_split;
    const int64_t row_end = (row_begin + rows_per_split < n_sample) ? row_begin + rows_per_split : n_sample;
    float bq[64];
    load_queries(q, qi, qvalid, half, bq);
    const float myqq = qvalid ? qq[qi] : 0.0f;
    float best = INFINITY;
    stream_tiles<QW>(db, dd, row_begin, row_end, tile, sDD, bq, [&](int, const f32x16 &acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // (qq + dd) - 2*ip in one rounding (2*ip is exact); rows past the end carry NaN and are ignored by fminf
            const float x = __builtin_fmaf(-2.0f, acc[r], myqq + sDD[rw * 32 + mfma_row(r, half)]);
            best = fminf(best, x);
        }
    });
    best = best < 0.0f ? 0.0f : best;     // the clamp commutes with the minimum
    if (qvalid && best < INFINITY) {
        // the lanes serving one query differ in (split, row-wave, half): consecutive ids cover all 64 groups
        const int g = (((split * RW + rw) * 2) + half) & (SR_GROUPS - 1);
        atomicMin(&gmin[(size_t)qi * SR_GROUPS + g], __float_as_int(best));
    }
}

// ---- launch 2: bound = k-th smallest group minimum (one wave per query) -------------------------------------------
__global__ __launch_bounds__(256) void search_thr_kernel(const int *__restrict__ gmin, int nq, int k,
                                                         float *__restrict__ thr) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    float d = __int_as_float(gmin[(size_t)qi * SR_GROUPS + lane]);
    int i = lane;
    wave_sort64(d, i, lane);
    if (lane == k - 1) thr[qi] = d;      // +inf when fewer than k groups saw a row: every row is a candidate
}

// ---- launch 3: the pass over the database -------------------------------------------------------------------------
template <int QW>
__global__ __launch_bounds__(256, 2) void search_scan_kernel(const float *__restrict__ db,
                                                             const float *__restrict__ dd, int64_t n,
                                                             const float *__restrict__ q,
                                                             const float *__restrict__ qq, int nq,
                                                             int64_t rows_per_split, const float *__restrict__ thr,
                                                             int *__restrict__ cnt, float *__restrict__ cand_d,
                                                             int *__restrict__ cand_i) {
    constexpr int RW = 4 / QW;
    constexpr int TROWS = 32 * RW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *tile = reinterpret_cast<float *>(smem);
    float *sDD = tile + TROWS * SR_LS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int qw = wave % QW, rw = wave / QW;
    int split, qgroup;
    search_block(split, qgroup);
    const int qi = (qgroup * QW + qw) * 32 + l31;
    const bool qvalid = qi < nq;
    const int64_t row_begin = (int64_t)split * rows_per_split;
    const int64_t row_end = (row_begin + rows_per_split < n) ? row_begin + rows_per_split : n;
    float bq[64];
    load_queries(q, qi, qvalid, half, bq);
    const float myqq = qvalid ? qq[qi] : 0.0f;
    const float thr_q = qvalid ? thr[qi] : -1.0f;      // -1: nothing passes (distances are clamped at 0)
    stream_tiles<QW>(db, dd, row_begin, row_end, tile, sDD, bq, [&](int t, const f32x16 &acc) {
        // dis = max(0, x), x = (qq + dd) - 2*ip in one rounding (2*ip is exact).  thr >= 0, so dis <= thr <=> x <= thr:
        // the common case is one add, one fma and one compare per element, the lane masks OR-ed in scalar registers.
        // Rows past the end carry NaN norms and fail the compare.
        float x[16];
        unsigned long long any = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x[r] = __builtin_fmaf(-2.0f, acc[r], myqq + sDD[rw * 32 + mfma_row(r, half)]);
            any |= __ballot(x[r] <= thr_q);
        }
        if (any != 0) {
            const int64_t slab0 = row_begin + (int64_t)t * TROWS + rw * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (x[r] <= thr_q) {
                    const int pos = atomicAdd(&cnt[qi], 1);
                    if (pos < SR_CAP) {                 // beyond: cnt > SR_CAP tells the select kernel to rescan
                        cand_d[(size_t)qi * SR_CAP + pos] = x[r] < 0.0f ? 0.0f : x[r];
                        cand_i[(size_t)qi * SR_CAP + pos] = (int)(slab0 + mfma_row(r, half));
                    }
                }
            }
        }
    });
}

// ---- launch 4: per query, the k best of its candidates by (distance, id) ------------------------------------------
__global__ __launch_bounds__(256) void search_select_kernel(const float *__restrict__ db,
                                                            const float *__restrict__ dd, int64_t n,
                                                            const float *__restrict__ q,
                                                            const float *__restrict__ qq, int nq, int k,
                                                            int64_t id_base, const float *__restrict__ thr,
                                                            const int *__restrict__ cnt,
                                                            const float *__restrict__ cand_d,
                                                            const int *__restrict__ cand_i,
                                                            float *__restrict__ out_d, int64_t *__restrict__ out_i) {
    __shared__ float pend_d[4][WT_PEND];
    __shared__ int pend_i[4][WT_PEND];
    __shared__ float wtop_d[4][32];
    __shared__ int wtop_i[4][32];
    __shared__ float sq[SR_D];
    const int qi = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    WaveTop top;
    top.init(pend_d[wave], pend_i[wave], thr[qi]);
    const int c = cnt[qi];
    if (c <= SR_CAP) {
        for (int e0 = 0; e0 < c; e0 += 256) {
            const int e = e0 + tid;
            const bool valid = e < c;
            const float d = valid ? cand_d[(size_t)qi * SR_CAP + e] : INFINITY;
            const int i = valid ? cand_i[(size_t)qi * SR_CAP + e] : SR_EMPTY;
            top.push(valid, d, i, k, lane);
        }
    } else {
        // the list overflowed: exact rescan of the whole database for this query, one row per thread, same
        // c-ordered fmaf chain and distance expression as the MFMA path
        if (tid < SR_D) sq[tid] = q[(size_t)qi * SR_D + tid];
        __syncthreads();
        const float myqq = qq[qi];
        const float4 *db4 = reinterpret_cast<const float4 *>(db);
        for (int64_t r0 = 0; r0 < n; r0 += 256) {
            const int64_t row = r0 + tid;
            const bool valid = row < n;
            float d = INFINITY;
            if (valid) {
                float ip = 0.0f;
                for (int c4 = 0; c4 < SR_D / 4; ++c4) {
                    const float4 x = db4[row * 32 + c4];
                    ip = __builtin_fmaf(x.x, sq[4 * c4 + 0], ip);
                    ip = __builtin_fmaf(x.y, sq[4 * c4 + 1], ip);
                    ip = __builtin_fmaf(x.z, sq[4 * c4 + 2], ip);
                    ip = __builtin_fmaf(x.w, sq[4 * c4 + 3], ip);
                }
                d = (myqq + dd[row]) - 2.0f * ip;
                d = d < 0.0f ? 0.0f : d;
            }
            top.push(valid, d, (int)row, k, lane);
        }
    }
    if (top.pc > 0) top.fold(k, lane);
    // merge the four wave lists as a tree: (0,1) and (2,3) in parallel, then the two winners
    if ((wave & 1) && lane < 32) {
        wtop_d[wave][lane] = top.td;
        wtop_i[wave][lane] = top.ti;
    }
    __syncthreads();
    float td = top.td;
    int ti = top.ti;
    if (!(wave & 1)) {
        if (lane >= 32) {
            td = wtop_d[wave + 1][lane - 32];
            ti = wtop_i[wave + 1][lane - 32];
        }
        wave_sort64(td, ti, lane);
        if (wave == 2 && lane < 32) {
            wtop_d[2][lane] = td;
            wtop_i[2][lane] = ti;
        }
    }
    __syncthreads();
    if (wave == 0) {
        if (lane >= 32) {
            td = wtop_d[2][lane - 32];
            ti = wtop_i[2][lane - 32];
        }
        wave_sort64(td, ti, lane);
        if (lane < k) {
            out_d[(size_t)qi * k + lane] = td;
            out_i[(size_t)qi * k + lane] = ti == SR_EMPTY ? (int64_t)-1 : id_base + (int64_t)ti;
        }
    }
}

// =====================================================================================================================
// bf16 pre-filter path (grafp_knn_search_l2_pre): the SAME exact results, from a 16x cheaper scan.
//
// The f32 scan above is bound by the exact-f32 matrix rate for large batches and by the 512 B/row stream for small
// ones.  Here the scan runs on a bf16 copy of the database (256 B/row) with v_mfma_f32_32x32x16_bf16, and only
// decides which rows MAY be among the k best; the exact f32 distance (same fmaf chain as the oracle) is then
// computed for those few hundred rows per query by the select kernel.  The decision is conservative:
//   x^ = round_bf16(x) has |x^_c - x_c| <= u |x_c|, u = 2^-8, so |<q^,x^> - <q,x>| <= (2u + u^2) |q| |x|
//   and, with |q||x| <= (qq + dd)/2, the approximate distance d~ = (qq + dd) - 2<q^,x^> obeys
//   |d~ - d| <= (2u + u^2)(qq + dd) = 0.0078278 (qq + dd);  SB_SLACK = 0.008 leaves 1.7e-4 (qq + dd) for the f32
//   accumulation inside the MFMA (<= 128 * 2^-24 relative) and the rounding of the test itself.
//   * bound pre-pass: group minima of d~ + SLACK (qq + dd) >= group minima of d  -> a valid bound, as before;
//   * scan: a row is kept iff d~ - SLACK (qq + dd) <= bound, i.e. every row with d <= bound is kept.
// Rearranged, the scan test per element is <q^,x^> >= A_q + H_row: one add and one compare.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in registers (uint4 went to scratch)
constexpr int SB_TR = 64;            // rows per ring stage
constexpr int SB_NS = 3;             // ring stages: two tiles in flight behind the one being multiplied
constexpr int SB_STAGE = SB_TR * 256 + SB_TR * 4;   // rows (256 B, 16-byte pieces XOR-swizzled by row) + their norms
constexpr int SB_PER = 5;            // LDS-DMA instructions per tile and wave: 4 x 1 KB of rows + 16 norms
constexpr float SB_SLACK = 0.008f;

// The bf16 pre-filter path's form of launches 0 and 2 in one: query norm (the serial chain of search_init_kernel, lane 0),
// empty candidate lists, and the bound from the pre-pass's raw group maxima gmax[q][ngroups] (group g belongs to class
// g & 63; d~ + SLACK (qq + dd) = qq kplus - 2 max, and the fma is monotone, so the class minimum of the bounds is the
// bound of the class maximum: the same bits the per-lane atomicMin of rounds 1-3 produced).  One wave per query.
__global__ __launch_bounds__(256) void search_thr_pre_kernel(const float *__restrict__ q, int nq,
                                                             const float *__restrict__ gmax, int ngroups, int k,
                                                             float *__restrict__ qq, float *__restrict__ thr,
                                                             int *__restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    float m = -INFINITY;
    {                                                      // eight loads in flight per lane (ngroups is 64 ... 1024)
        const float *gq = gmax + (size_t)qi * ngroups;
        int g = lane;
        for (; g + 7 * 64 < ngroups; g += 8 * 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = gq[g + 64 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u) m = fmaxf(m, v[u]);
        }
        for (; g < ngroups; g += 64) m = fmaxf(m, gq[g]);
    }
    // the query's norm: the serial chain of row_sqnorm_kernel, every lane running it on broadcasts (v_readlane) of the two
    // coalesced loads that fetched the row -- one lane walking the row element by element was 5 of this kernel's 8 us
    const float v0 = q[(size_t)qi * SR_D + lane], v1 = q[(size_t)qi * SR_D + 64 + lane];
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 64; ++c) {
        const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v0), c));
        s = __builtin_fmaf(x, x, s);
    }
#pragma unroll
    for (int c = 0; c < 64; ++c) {
        const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v1), c));
        s = __builtin_fmaf(x, x, s);
    }
    if (lane == 0) qq[qi] = s;
    if (lane < SR_NSUB) cnt[qi * SR_NSUB + lane] = 0;
    float d = m > -INFINITY ? __builtin_fmaf(-2.0f, m, s * (1.0f + SB_SLACK)) : INFINITY;
    d = d < 0.0f ? 0.0f : d;
    int i = lane;
    wave_sort64(d, i, lane);
    if (lane == k - 1) thr[qi] = d;      // +inf when fewer than k groups saw a row: every row is a candidate
}

constexpr int SB_WGS_NQS2 = 2;      // workgroups per CU of the two-query-set form (three: 168 registers, 16-21 dwords spilled, same time)

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
    unsigned int u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float *__restrict__ src, int64_t n,
                                                          unsigned short *__restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dst[i] = f32_to_bf16_rne(src[i]);
}

// query operand of the bf16 MFMA: lane (l31, half) holds dims 16 s + 8 half + 0..7 of query l31, s = 0..7
// `scale`: the operand is round_bf16(scale * q) -- the scans fold their constant factor into the query side, see
// stream_tiles_bf16 (the error bound of the product is relative to |scale q||x|, i.e. unchanged after dividing back)
__device__ __forceinline__ void load_queries_bf16(const float *__restrict__ q, int qi, bool qvalid, int half,
                                                  bf16x8 (&bq)[8], float scale) {
    typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));   // q is only promised 4-byte alignment
    const f32x4_a4 *qrow = reinterpret_cast<const f32x4_a4 *>(q + (size_t)(qvalid ? qi : 0) * SR_D + 8 * half);
    f32x4 raw[16];                           // unconditional loads, all in flight at once (see load_queries)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        raw[2 * s] = qrow[4 * s];
        raw[2 * s + 1] = qrow[4 * s + 1];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e)
            bq[s][e] = qvalid ? (short)f32_to_bf16_rne(raw[2 * s + (e >> 2)][e & 3] * scale) : (short)0;
}

// Tiles of SB_TR bf16 rows through a ring of SB_NS LDS stages, filled by LDS-DMA (no registers, no LDS-write phase):
// every wave issues its quarter of a tile two tiles ahead -- 4 x global_load_lds_dwordx4 (64 lanes x 16 B = 4 rows each)
// and one dword DMA for 16 of the rows' norms -- and waits for it with a hand-counted vmcnt before the tile's barrier.
// (A DMA instruction costs its wave 60-185 issue cycles: with two producer waves the other two stood at the barrier.)
//   The register-prefetch form this replaces had ONE tile in flight and, through a conditional load the compiler
//   drained on the spot, waited for it right after issuing it; its per-block code recycled one fragment register quad
//   (read, wait, MFMA, eight times) and built 16 lane masks with 16 dependent scalar ORs: 12 000 cycles per 128-row
//   tile at nq = 4096 against 3 000 of MFMA work.
//   Measured and dropped: touching the lines of the tiles further ahead (one dword per 128-byte line) to pull them into
//   L2 early -- nq = 4096 unchanged, nq = 41 scan 82 us against 60 (96 streams per XCD overflow its L2: every line is
//   fetched twice); the DMA instructions issued from inside the MFMA chain instead of in front of the blocks (neutral).
//   Where a tile's ~4 000 cycles per wave go at nq = 4096 (s_memtime, three workgroups per CU): LDS fragment batches
//   ~650, MFMA chains ~510, DMA issue ~380, barrier + DMA wait ~430, block tests ~130 each and ~650 for the every second
//   block that holds a hit.
// LDS rows are 256 B unpadded; piece p of LDS row r holds piece p ^ (r & 15) of the database row (the DMA lane picks
// its global address accordingly), so the 16-byte fragment reads of a lane group hit 16 different bank groups.
// Wave (qw, rw) multiplies row blocks rw, rw + RW, ... of the tile with its NQS sets of 32 queries: on_block(t, rb, j, acc,
// nb) per set j; on_tile(t) runs once per tile on every thread at the quiescent point behind the tile barrier.
// Round 6: the accumulator chains START from the rows' squared norms (the C operand of a block's first MFMA is the
// lane's 16 norms as they come out of LDS -- NaN past the end of the slice), and the caller folds its constant into the
// query operand (q' = -2 q / kappa): acc = dd + <q'^, x^> = dd - (2 / kappa) <q^,x^>, the quantity both scans threshold
// (scan: keep iff acc <= thr / kminus - qq; pre-pass: the minimum of acc).  Before, every block paid eight v_pk_fma_f32
// per query set for hv * k + acc and sixteen register moves to line the norms up -- and every VALU slot of this loop is
// paid in full (tools/search_abl.py: the block callback was 300 us of a 1 130 us pass over 1M rows x 4096 queries, 100
// after: profiles/r06_search_abl_nqs2.txt, r06_search_abl_nqs2_after.txt).
// nb: the lane's 16 norms where they stand in LDS (row mfma_row(r, half) at nb[8 (r >> 2) + (r & 3)]).
// ABL (measurement builds, tools/search_abl.py): bit 0 drops the MFMAs, bit 1 the per-block callback, bit 2 the DMA issue
// after the prologue, bit 3 the fragment reads -- what the loop costs without each of its parts.
// prepare() runs once, right behind the DMA issue of the first two tiles: the caller loads and converts its query operand
// there, under the ring fill instead of in front of it (the two latencies added up in every workgroup's prologue).
template <int QW, int NQS, int ABL = 0, typename P, typename F, typename G>
__device__ __forceinline__ void stream_tiles_bf16(const unsigned short *__restrict__ dbh, const float *__restrict__ dd,
                                                  int64_t row_begin, int64_t row_end, unsigned char *ring,
                                                  bf16x8 (&bq)[NQS][8], P &&prepare, F &&on_block, G &&on_tile) {
    constexpr int RW = 4 / QW;
    static_assert(SB_TR % (32 * RW) == 0 && SB_TR == 64, "one dword DMA covers the 64 norms of a tile");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int rw = wave / QW;
    const int ntiles = row_end > row_begin ? (int)((row_end - row_begin + SB_TR - 1) / SB_TR) : 0;
    if (ntiles == 0) {
        prepare();
        return;
    }
    const int nrows = (int)(row_end - row_begin);
    const unsigned lds0 = (unsigned)(uintptr_t)(gm_lptr)ring;
    const unsigned char *rows = reinterpret_cast<const unsigned char *>(dbh) + row_begin * 256;
    const float *ddb = dd + row_begin;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // LDS-DMA bases travel in M0: scalar operands
    const int sub = lane >> 4, pos = lane & 15;
    // Every VALU instruction of this loop is paid in full: on one SIMD the waves' MFMAs and their other vector
    // instructions add up (tools/search_abl.py: 631 us of MFMA + 625 us of everything else = 1207 us, round 4).  So the
    // addresses are strength-reduced: the DMA source of piece j in tile t is a per-lane pointer of tile 0 plus t * 16 KB
    // (one 64-bit add; the clamp against the end of the slice only in its last, partial tile), and the ring stage is a
    // compile-time constant (the tile loop is unrolled by SB_NS), so a fragment read is ds_read_b128 with one of eight
    // per-lane registers and an immediate offset -- no address arithmetic per block at all.
    const bool partial = (nrows & (SB_TR - 1)) != 0;
    const unsigned char *src0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 4 * (wave_u * 4 + j) + sub;
        src0[j] = rows + (size_t)r * 256 + ((pos ^ (r & 15)) << 4);
    }
    const float *nsrc0 = ddb + wave_u * 16 + pos;
    auto issue = [&](int t, int stage) {              // this wave's quarter of tile t (clamped by the caller)
        const unsigned st = __builtin_amdgcn_readfirstlane(lds0 + stage * SB_STAGE);
        if (partial && t == ntiles - 1) {             // uniform; rows past the end: any readable data, their norm -> NaN
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = wave_u * 4 + j, r = 4 * i + sub;
                int lr = t * SB_TR + r;
                lr = lr < nrows ? lr : nrows - 1;
                gm_dma16(rows + (size_t)lr * 256 + ((pos ^ (r & 15)) << 4), st + i * 1024);
            }
            int lr = t * SB_TR + wave_u * 16 + pos;    // norms of rows 16 w .. 16 w + 15: lanes 0-15 only
            lr = lr < nrows ? lr : nrows - 1;
            if (lane < 16) gm_dma4(ddb + lr, st + SB_TR * 256 + wave_u * 64);
        } else {
            const size_t off = (size_t)t * (SB_TR * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j) gm_dma16(src0[j] + off, st + (wave_u * 4 + j) * 1024);
            if (lane < 16) gm_dma4(nsrc0 + (size_t)t * SB_TR, st + SB_TR * 256 + wave_u * 64);
        }
    };
#pragma unroll
    for (int t0 = 0; t0 < SB_NS - 1; ++t0) issue(t0 < ntiles ? t0 : ntiles - 1, t0);
    prepare();
    const unsigned char *frag[8];                     // fragment s2 of block 0 in stage 0, this lane
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) frag[s2] = ring + l31 * 256 + (((2 * s2 + half) ^ (l31 & 15)) << 4);
    const float *nrm0 = reinterpret_cast<const float *>(ring + SB_TR * 256) + 4 * half;
    auto tile = [&](auto stage_c, int t) {
        constexpr int STAGE = decltype(stage_c)::value;
        gm_wait_vm<(SB_NS - 2) * SB_PER>();                   // this wave's part of tile t has landed
        __syncthreads();  // everybody's part has; every wave is done with tile t-1, whose stage is free again
        on_tile(t);       // quiescent point: no wave is inside on_block, so workgroup state is uniform here
        if (partial && t == ntiles - 1) {                     // uniform: the slice ends inside this tile
            float *sd = reinterpret_cast<float *>(ring + STAGE * SB_STAGE + SB_TR * 256);
            if (tid < SB_TR && t * SB_TR + tid >= nrows) sd[tid] = __builtin_nanf("");
            __syncthreads();
        }
        constexpr int FREE = STAGE == 0 ? SB_NS - 1 : STAGE - 1;
        if (!(ABL & 4)) issue(t + SB_NS - 1 < ntiles ? t + SB_NS - 1 : ntiles - 1, FREE);
        // (Measured and dropped, round 4: both blocks of the tile as one software pipeline in the 4x1 shape -- 16 fragment
        // reads up front, the two MFMA chains interleaved, block 0's arithmetic behind them; 168 registers.  4096 queries
        // 1.45 ms against 1.39, 2048 0.76 against 0.73, 512 0.243 against 0.262.)
        auto block = [&](auto rb_c) {
            constexpr int RB = decltype(rb_c)::value;
            constexpr int OFF = STAGE * SB_STAGE + RB * 32 * 256, NOFF = STAGE * SB_STAGE + RB * 32 * 4;
            // all LDS reads of the block -- eight row fragments and the 16 norms of this lane's accumulator rows -- go
            // out as one batch ahead of the MFMA chain (left alone, the compiler recycles ONE fragment register quad:
            // read, wait, MFMA, eight times over, i.e. eight exposed LDS latencies per block)
            bf16x8 a[8];
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
                if (ABL & 8) a[s2] = bq[0][s2];
                else a[s2] = *reinterpret_cast<const bf16x8 *>(frag[s2] + OFF);
            }
            // ... and the 16 norms of the lane's accumulator rows: the C operand of the chains' first MFMA
            f32x4 hv4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                hv4[g] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const unsigned char *>(nrm0 + 8 * g) + NOFF);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 cinit;                      // cinit[r]: norm of row mfma_row(r, half) of the block (NaN past the end)
#pragma unroll
            for (int r = 0; r < 16; ++r) cinit[r] = hv4[r >> 2][r & 3];
            // NQS independent accumulator chains share every row fragment: with two query sets per wave a fragment
            // read feeds two MFMAs and neither chain waits for the other's result
            f32x16 acc[NQS];
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
#pragma unroll
                for (int j = 0; j < NQS; ++j) {
                    if (ABL & 1) {
                        if (s2 == 0) acc[j] = cinit;
                        acc[j][s2] += __builtin_bit_cast(float, (int)a[s2][0] | ((int)a[s2][7] << 16));
                    } else {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s2], bq[j][s2], s2 == 0 ? cinit : acc[j], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NQS; ++j) {
                const float *nb = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(nrm0) + NOFF);
                if (ABL & 2) { if (acc[j][0] == 12345.678f && acc[j][3] == acc[j][5]) on_block(t, RB, j, acc[j], nb); }
                else on_block(t, RB, j, acc[j], nb);
            }
        };
        // (Also measured and dropped for the two-query-set form: the reads of both blocks in front of the first chain,
        // 236-256 registers -- 4096 queries 1.201 ms against 1.203.)
        if constexpr (RW == 1) {
            block(std::integral_constant<int, 0>{});
            block(std::integral_constant<int, 1>{});
        } else {                                              // RW == 2: row-wave rw takes block rw
            if (rw == 0) block(std::integral_constant<int, 0>{});
            else block(std::integral_constant<int, 1>{});
        }
    };
    static_assert(SB_NS == 3 && (RW == 1 || RW == 2), "the tile loop is unrolled by hand over three ring stages");
    for (int t = 0; t < ntiles; t += SB_NS) {
        tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntiles) tile(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < ntiles) tile(std::integral_constant<int, 2>{}, t + 2);
    }
    gm_wait_vm<0>();      // nothing of this wave's is in flight when the caller reuses LDS or the wave ends
}

template <int QW, int NQS, int ABL = 0>
__global__ __launch_bounds__(256, NQS >= 2 ? SB_WGS_NQS2 : 3) void search_bound_bf16_kernel(
    const unsigned short *__restrict__ dbh, const float *__restrict__ dd, int64_t n_sample, const float *__restrict__ q,
    int nq, int64_t rows_per_split, float *__restrict__ gmax, int ngroups) {
    constexpr int RW = 4 / QW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned char *ring = reinterpret_cast<unsigned char *>(smem);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int qw = wave % QW, rw = wave / QW;
    int split, qgroup;
    search_block(split, qgroup);
    const int64_t row_begin = (int64_t)split * rows_per_split;
    const int64_t row_end = (row_begin + rows_per_split < n_sample) ? row_begin + rows_per_split : n_sample;
    const float kplus = 1.0f + SB_SLACK;
    bf16x8 bq[NQS][8];
    int qi[NQS];
    float bestm[NQS];
#pragma unroll
    for (int j = 0; j < NQS; ++j) {
        qi[j] = ((qgroup * QW + qw) * NQS + j) * 32 + l31;
        bestm[j] = INFINITY;
    }
    // d~ + SLACK (qq + dd) = qq kplus - 2 (<q^,x^> - dd kplus / 2), and the bracket is -(kplus / 2) acc for the stream's
    // acc = dd - (2 / kplus) <q^,x^> (query operand scaled by -2 / kplus): the lane keeps the MINIMUM of acc, three at a
    // time, no arithmetic per element; rows past the end carry NaN and are ignored by fminf
    const float nhk = -0.5f * kplus;
    stream_tiles_bf16<QW, NQS, ABL>(dbh, dd, row_begin, row_end, ring, bq, [&]() {
#pragma unroll
        for (int j = 0; j < NQS; ++j) load_queries_bf16(q, qi[j], qi[j] < nq, half, bq[j], -2.0f / kplus);
    }, [&](int, int, int j, const f32x16 &acc, const float *) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) bestm[j] = fminf(fminf(bestm[j], acc[r]), acc[r + 1]);
    }, [](int) {});
    // every (split, row-wave, half) group of a query has ONE writer: the maximum of the bracket goes out as it is
    // (-inf when the group saw no row) and search_thr_pre_kernel folds the groups into the 64 classes -- no atomics, so no
    // launch in front of this one to initialise them (round 4: the init kernel was 5 us of a 79 us search)
#pragma unroll
    for (int j = 0; j < NQS; ++j)
        if (qi[j] < nq)
            gmax[(size_t)qi[j] * ngroups + ((split * RW + rw) * 2 + half)] = bestm[j] < INFINITY ? nhk * bestm[j] : -INFINITY;
}

// Hits are rare (a few hundred per query over the whole database) but a returning global atomic costs microseconds,
// so the MFMA loop only appends (query, row) to an LDS queue; a full queue is drained to the per-query candidate lists.
// The queues are WAVE-PRIVATE: the fill count is a wave-uniform scalar, a hit's slot is count + (hit lanes below this
// one) from the ballot -- no LDS atomic (its returning round trip was most of the ~650 cycles a block with a hit
// cost, and every second block holds one), no workgroup barrier around a drain, no shared state at all.
constexpr int HB_CAP = 72;        // entries per wave: 4 x 72 x 13 B = 3.7 KB; with the 49 KB ring three workgroups fit a CU

template <int QW, int NQS>
__global__ __launch_bounds__(256, NQS >= 2 ? SB_WGS_NQS2 : 3) void search_scan_bf16_kernel(
    const unsigned short *__restrict__ dbh, const float *__restrict__ dd, int64_t row0, int64_t n,
    const float *__restrict__ q, const float *__restrict__ qq, int nq, int64_t rows_per_split,
    const float *__restrict__ thr, int *__restrict__ cnt, int *__restrict__ cand_i, float2 *__restrict__ cand_e) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned char *ring = reinterpret_cast<unsigned char *>(smem);
    __shared__ int hb_row[4][HB_CAP];
    __shared__ float hb_e[4][HB_CAP];
    __shared__ float hb_d[4][HB_CAP];
    // (one byte per entry wherever the workgroup's query count allows: the one-set forms run THREE workgroups per CU and
    //  their 49 KB ring + these queues fill the LDS to the last allocation unit -- with two-byte entries only two fit, and
    //  41 queries took 0.086 instead of 0.076 ms, 128 queries 0.103 instead of 0.085)
    typedef typename std::conditional<(QW * NQS * 32 > 256), unsigned short, unsigned char>::type hb_q_t;
    __shared__ hb_q_t hb_q[4][HB_CAP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int qw = wave % QW;
    int split, qgroup;
    search_block(split, qgroup);
    const int qbase = qgroup * QW * NQS * 32;        // the workgroup's queries: qbase + [0, 32 QW NQS)
    const int64_t row_begin = row0 + (int64_t)split * rows_per_split;      // the launch covers rows [row0, n)
    const int64_t row_end = (row_begin + rows_per_split < n) ? row_begin + rows_per_split : n;
    // keep iff d~ - SLACK (qq + dd) <= bound  <=>  <q^,x^> >= A_q + H_row
    const float kminus = 1.0f - SB_SLACK;
    bf16x8 bq[NQS][8];
    float a_q[NQS];
    const int sub = split & (SR_NSUB - 1);
    int *my_row = hb_row[wave];
    float *my_e = hb_e[wave];
    float *my_d = hb_d[wave];
    hb_q_t *my_q = hb_q[wave];
    int fill = 0;                                             // wave-uniform
    auto drain = [&]() {                                      // this wave's queue -> sub-list `sub` of the queries' lists
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int e = lane; e < fill; e += 64) {
            const int qg = qbase + my_q[e];
            const int pos = atomicAdd(&cnt[qg * SR_NSUB + sub], 1);
            if (pos < SR_SUBCAP) {                            // beyond: the select kernel sees the count and rescans
                cand_i[(size_t)qg * SR_CAP + sub * SR_SUBCAP + pos] = my_row[e];
                // E = <q^,x^> - dd kminus / 2 (bounds d later) and the row's norm (the select kernel's upper bound needs it:
                // carried along, the select kernel has no gather between its candidate load and its histogram)
                cand_e[(size_t)qg * SR_CAP + sub * SR_SUBCAP + pos] = make_float2(my_e[e], my_d[e]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        fill = 0;
    };
    auto push = [&](bool hit, int row, float ev, float ddv, int ql) {    // wave-level call
        const unsigned long long mask = __ballot(hit);
        const int add = __popcll(mask);
        if (fill + add > HB_CAP) drain();                     // uniform
        const int slot = fill + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        if (hit) {
            my_row[slot] = row;
            my_e[slot] = ev;
            my_d[slot] = ddv;
            my_q[slot] = (hb_q_t)ql;
        }
        fill += add;
    };
    const float hs = 0.5f * kminus;
    // keep iff <q^,x^> - dd kminus / 2 >= (qq kminus - thr) / 2  <=>  acc <= thr / kminus - qq for the stream's
    // acc = dd - (2 / kminus) <q^,x^> (query operand scaled by -2 / kminus).  Common case: the minimum of the 16
    // accumulators three at a time (NaN past the end of the slice drops out of fminf), ONE compare and one branch per
    // block -- no arithmetic per element, no per-element lane masks (whose 16 dependent scalar ORs behind 16 VALU
    // compares cost as much as the MFMA chain itself).
    stream_tiles_bf16<QW, NQS>(dbh, dd, row_begin, row_end, ring, bq, [&]() {
#pragma unroll
        for (int j = 0; j < NQS; ++j) {
            const int qi = qbase + (qw * NQS + j) * 32 + l31;
            load_queries_bf16(q, qi, qi < nq, half, bq[j], -2.0f / kminus);
            a_q[j] = qi < nq ? thr[qi] / kminus - qq[qi] : -INFINITY;
        }
    }, [&](int t, int rb, int j, const f32x16 &e, const float *nb) {
        float m = fminf(e[0], e[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) m = fminf(fminf(m, e[r]), e[r + 1]);
        if (__ballot(m <= a_q[j]) != 0) {
            // A block holds a hit far more often than "rare" suggests -- 32 x 32 pairs against a few hundred candidates
            // per query in a million rows: every second to fourth block -- so this path must be cheap as well: 16
            // compares into a bit mask (no branches); in the usual case of ONE hit among a lane's 16 rows its value is
            // the maximum already at hand.
            unsigned bits = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) bits |= e[r] <= a_q[j] ? 1u << r : 0u;       // (NaN compares false)
            const int slab0 = (int)(row_begin + (int64_t)t * SB_TR + rb * 32) + 4 * half;
            const int ql = (qw * NQS + j) * 32 + l31;                 // < 32 QW NQS <= 384
            const bool single = (bits & (bits - 1)) == 0;
            const int r0 = bits ? __builtin_ctz(bits) : 0;            // row mfma_row(r, half) of the block
            const int off0 = (r0 & 3) + 8 * (r0 >> 2);
            // the record the select kernel reads: E = <q^,x^> - dd kminus / 2 = -(kminus / 2) acc, and the row's norm
            push(bits != 0 && single, slab0 + off0, -hs * m, nb[off0], ql);  // (the norm: one LDS read at the lane's own index)
            if (__ballot(!single) != 0) {                             // several hits in one lane's 16 rows: rare
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool h = !single && ((bits >> r) & 1);
                    if (__ballot(h) != 0) push(h, slab0 + (r & 3) + 8 * (r >> 2), -hs * e[r], nb[(r & 3) + 8 * (r >> 2)], ql);
                }
            }
        }
    }, [](int) {});
    drain();
}

// Per query: the k best of its candidates by (exact distance, id).
// The scan left (row, E) pairs, E = <q^,x^> - dd[row] (1 - SLACK) / 2 -- the very value its test compared.  With
// t = qq + dd[row] the true distance lies in [lo, hi],
//   lo = t (1 - SLACK) - 2 <q^,x^> = qq (1 - SLACK) - 2 E             (no dd[row]: phase B needs no gather for it),
//   hi = t (1 + SLACK) - 2 <q^,x^> = qq (1 + SLACK) - 2 E + 2 SLACK dd[row]
// (the roundings of these forms are a few 2^-24 (qq + dd), against the 1.7e-4 (qq + dd) SB_SLACK keeps in reserve),
// so phase A takes the k-th smallest hi (at least k rows are truly that close: a bound ~10x tighter than the scan's)
// and phase B evaluates the exact f32 distance -- the oracle's fmaf chain over the 512-byte row -- only for the rows
// whose lo does not exceed it: a few dozen random row reads per query instead of several hundred.
// TIGHTEN: only phase A, over the candidates the first part of a two-part scan left: thr[qi] = min(thr[qi], the k-th
// smallest upper bound) -- the bound the second part then scans with (at least k rows of the first part are that close).
template <bool TIGHTEN>
__global__ __launch_bounds__(256, 4) void search_select_exact_kernel(const float *__restrict__ db,
                                                                  const float *__restrict__ dd, int64_t n,
                                                                  const float *__restrict__ q,
                                                                  const float *__restrict__ qq, int nq, int k,
                                                                  int64_t id_base, float *__restrict__ thr,
                                                                  const int *__restrict__ cnt,
                                                                  const int *__restrict__ cand_i,
                                                                  const float2 *__restrict__ cand_e,
                                                                  float *__restrict__ out_d,
                                                                  int64_t *__restrict__ out_i) {
    __shared__ float pend_d[4][WT_PEND];
    __shared__ int pend_i[4][WT_PEND];
    __shared__ float wtop_d[4][32];
    __shared__ int wtop_i[4][32];
    __shared__ float sq[SR_D];
    __shared__ float s_thr2;
    __shared__ int need_rows[SR_CAP];                      // every listed candidate fits: no overflow case
    __shared__ int s_nneed;
    const int qi = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid < SR_D) sq[tid] = q[(size_t)qi * SR_D + tid];
    __shared__ int s_off[SR_NSUB + 1];
    const float myqq = qq[qi];
    if (wave == 0) {                                       // sub-list fills -> offsets of a flat candidate index
        const int cs = lane < SR_NSUB ? cnt[qi * SR_NSUB + lane] : 0;      // (one load per lane, a 16-lane prefix sum)
        const bool over = cs > SR_SUBCAP;
        const int v = over ? SR_SUBCAP : cs;
        int inc = v;
#pragma unroll
        for (int j = 1; j < SR_NSUB; j <<= 1) {
            const int o = __shfl_up(inc, j);
            if (lane >= j) inc += o;
        }
        const bool any_over = __ballot(over) != 0;
        if (lane < SR_NSUB) s_off[lane] = inc - v;
        if (lane == SR_NSUB - 1) s_off[SR_NSUB] = any_over ? -1 : inc;
    }
    __syncthreads();
    const int c = s_off[SR_NSUB];
    const bool listed = c >= 0;                            // else: a sub-list overflowed -> exact rescan of every row
    auto slot_of = [&](int e) {                            // flat candidate index -> slot in the query's list
        int s2 = 0;
#pragma unroll
        for (int b2 = SR_NSUB / 2; b2 > 0; b2 >>= 1) s2 += (e >= s_off[s2 + b2]) ? b2 : 0;
        return s2 * SR_SUBCAP + (e - s_off[s2]);
    };
    float thr2 = thr[qi];
    const float kminus = 1.0f - SB_SLACK, kplus = 1.0f + SB_SLACK;
    const float qhi = myqq * kplus, dspan = kplus - kminus;     // (the difference of two floats this close is exact)
    WaveTop top;
    float td;
    int ti;
    auto exact = [&](int64_t row) {                        // the oracle's chain over the 512-byte row
        // the whole row is requested before the dependent fmaf chain starts (every lane reads a different row)
        f32x4 xr[SR_D / 4];
        const f32x4 *rp = reinterpret_cast<const f32x4 *>(db) + row * (SR_D / 4);
#pragma unroll
        for (int c4 = 0; c4 < SR_D / 4; ++c4) xr[c4] = rp[c4];
        const float ddr = dd[row];
        float ip = 0.0f;
#pragma unroll
        for (int c4 = 0; c4 < SR_D / 4; ++c4) {
            ip = __builtin_fmaf(xr[c4][0], sq[4 * c4 + 0], ip);
            ip = __builtin_fmaf(xr[c4][1], sq[4 * c4 + 1], ip);
            ip = __builtin_fmaf(xr[c4][2], sq[4 * c4 + 2], ip);
            ip = __builtin_fmaf(xr[c4][3], sq[4 * c4 + 3], ip);
        }
        const float d = (myqq + ddr) - 2.0f * ip;
        return d < 0.0f ? 0.0f : d;
    };
    // ---- the usual case: a few hundred candidates.  No sorting networks at all (a 64-lane bitonic sort is ~1 700 cycles,
    // and the fold + three-sort merge of the general path below ran twice per query: 16 of the kernel's 29 us at 41
    // queries, tools' stop-point timing, round 4):
    //   phase A  the candidates (row, E, hi) stay in registers (up to eight per thread); the bound is read from a 256-bin histogram of hi between its
    //            minimum and maximum -- the upper edge (plus one bin against the rounding of the bin index) of the bin
    //            the k-th smallest falls into: >= the k-th smallest hi, so still a valid bound, a bin or two looser;
    //   phase B  the rows with lo <= bound are compacted, their exact distances computed one per thread, and every
    //            thread finds the RANK of its row by counting the (distance, id) pairs before it -- ranks < k are the
    //            answer, written in place.  O(need^2 / 256) compares per thread: 40 at the usual hundred rows.
    constexpr int SF_CAP = 2048, SF_NEED = 512, SF_PER = SF_CAP / 256;
    __shared__ int f_row[SF_NEED];
    __shared__ float f_hi[SF_NEED];
    __shared__ int f_hist[256];
    __shared__ float f_red[2][4];
    if (listed && c <= SF_CAP) {
        int row[SF_PER];
        float ev[SF_PER], ddv[SF_PER], hi[SF_PER];
#pragma unroll
        for (int u = 0; u < SF_PER; ++u) {                 // every load of a kind in flight at once; the rounds past the
            row[u] = 0;                                    // end of the list (u * 256 >= c: uniform) load nothing
            ev[u] = 0.0f;
            ddv[u] = 0.0f;
            if (u * 256 < c) {
                const int e = u * 256 + tid;
                const size_t sl = (size_t)qi * SR_CAP + slot_of(e < c ? e : c - 1);
                row[u] = cand_i[sl];
                const float2 ed = cand_e[sl];
                ev[u] = ed.x;
                ddv[u] = ed.y;                             // the row's norm rides with the candidate: no gather here
            }
        }
        float mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < SF_PER; ++u) {
            hi[u] = __builtin_fmaf(dspan, ddv[u], __builtin_fmaf(-2.0f, ev[u], qhi));
            hi[u] = hi[u] < 0.0f ? 0.0f : hi[u];
            if (u * 256 + tid < c) {
                mn = fminf(mn, hi[u]);
                mx = fmaxf(mx, hi[u]);
            }
        }
        f_hist[tid] = 0;
#pragma unroll
        for (int j = 1; j < 64; j <<= 1) {
            mn = fminf(mn, __shfl_xor(mn, j));
            mx = fmaxf(mx, __shfl_xor(mx, j));
        }
        if (lane == 0) {
            f_red[0][wave] = mn;
            f_red[1][wave] = mx;
        }
        __syncthreads();
        const float lo_h = fminf(fminf(f_red[0][0], f_red[0][1]), fminf(f_red[0][2], f_red[0][3]));
        const float hi_h = fmaxf(fmaxf(f_red[1][0], f_red[1][1]), fmaxf(f_red[1][2], f_red[1][3]));
        const float w = (hi_h - lo_h) * (1.0f / 256.0f);
        const float inv = (w > 0.0f && w < INFINITY) ? 1.0f / w : 0.0f;
#pragma unroll
        for (int u = 0; u < SF_PER; ++u) {
            if (u * 256 + tid < c) {
                int b = (int)((hi[u] - lo_h) * inv);
                b = b < 0 ? 0 : (b > 255 ? 255 : b);
                atomicAdd(&f_hist[b], 1);
            }
        }
        __syncthreads();
        if (wave == 0) {                                   // bins 4 lane .. 4 lane + 3: inclusive scan over the lanes
            const int h0 = f_hist[4 * lane], h1 = f_hist[4 * lane + 1], h2 = f_hist[4 * lane + 2], h3 = f_hist[4 * lane + 3];
            const int mine = h0 + h1 + h2 + h3;
            int cum = mine;
#pragma unroll
            for (int j = 1; j < 64; j <<= 1) {
                const int o = __shfl_up(cum, j);
                if (lane >= j) cum += o;
            }
            const int before = cum - mine;                 // candidates in the bins of the lanes below
            if (c < k) {
                if (lane == 0) s_thr2 = INFINITY;          // fewer than k candidates exist
            } else if (before < k && cum >= k) {           // exactly one lane
                int b = 4 * lane, run = before + h0;
                if (run < k) { ++b; run += h1; }
                if (run < k) { ++b; run += h2; }
                if (run < k) { ++b; }
                s_thr2 = fminf(hi_h, __builtin_fmaf((float)(b + 2), w, lo_h));
            }
        }
        __syncthreads();
        thr2 = fminf(thr2, s_thr2);
        if (TIGHTEN) {
            if (tid == 0) thr[qi] = thr2;
            return;
        }
        const float qlo = myqq * kminus;
        if (tid == 0) s_nneed = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SF_PER; ++u)
            if (u * 256 + tid < c && __builtin_fmaf(-2.0f, ev[u], qlo) <= thr2)          // lo: no dd[row] needed
                need_rows[atomicAdd(&s_nneed, 1)] = row[u];
        __syncthreads();
        const int nneed = s_nneed;
        if (nneed <= SF_NEED) {                            // (uniform) else: the general path below, from the lists
            float dmine[SF_NEED / 256];
            int rmine[SF_NEED / 256];
#pragma unroll
            for (int v = 0; v < SF_NEED / 256; ++v) {
                const int i = v * 256 + tid;
                rmine[v] = i < nneed ? need_rows[i] : 0;
                dmine[v] = (v * 256 < nneed) ? (i < nneed ? exact(rmine[v]) : INFINITY) : INFINITY;
                if (i < nneed) {
                    f_hi[i] = dmine[v];
                    f_row[i] = rmine[v];
                }
            }
            __syncthreads();
#pragma unroll
            for (int v = 0; v < SF_NEED / 256; ++v) {
                const int i = v * 256 + tid;
                if (v * 256 < nneed) {                     // uniform
                    int rank = 0;
                    for (int j = 0; j < nneed; ++j) rank += lex_lt(f_hi[j], f_row[j], dmine[v], rmine[v]) ? 1 : 0;
                    if (i < nneed && rank < k) {
                        out_d[(size_t)qi * k + rank] = dmine[v];
                        out_i[(size_t)qi * k + rank] = id_base + (int64_t)rmine[v];
                    }
                }
            }
            if (tid >= nneed && tid < k) {                 // fewer rows than k (k <= 32 < 256)
                out_d[(size_t)qi * k + tid] = INFINITY;
                out_i[(size_t)qi * k + tid] = -1;
            }
            return;
        }
        thr2 = thr[qi];                                    // the general path recomputes its own bound
        __syncthreads();
    }
    if (listed) {                                          // phase A: k-th smallest upper bound
        top.init(pend_d[wave], pend_i[wave], INFINITY);
        // four candidates per thread and round, every load of a kind in flight at once (indices clamped to the last
        // candidate instead of branching around the loads): a list of 600 costs two dependent round trips, not six
        for (int e0 = 0; e0 < c; e0 += 4 * 256) {
            int row[4];
            float ev[4], ddv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * 256 + tid;
                const size_t sl = (size_t)qi * SR_CAP + slot_of(e < c ? e : c - 1);
                row[u] = cand_i[sl];
                ev[u] = cand_e[sl].x;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) ddv[u] = dd[row[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (e0 + u * 256 < c) {                    // uniform: push is a wave-level call
                    float hi = __builtin_fmaf(dspan, ddv[u], __builtin_fmaf(-2.0f, ev[u], qhi));
                    hi = hi < 0.0f ? 0.0f : hi;
                    top.push(e0 + u * 256 + tid < c, hi, row[u], k, lane);
                }
            }
        }
        if (top.pc > 0) top.fold(k, lane);
        block_merge_tops(top, wtop_d, wtop_i, wave, lane, td, ti);
        if (wave == 0 && lane == k - 1) s_thr2 = td;      // +inf when fewer than k candidates exist
        __syncthreads();
        thr2 = fminf(thr2, s_thr2);
    }
    if (TIGHTEN) {
        if (tid == 0 && listed) thr[qi] = thr2;
        return;
    }
    __syncthreads();                                       // sq visible; the merge buffers are free again
    // phase B: exact distances of the rows that can still be among the k best
    top.init(pend_d[wave], pend_i[wave], thr2);
    if (listed) {
        // the rows still in question (lo <= the bound of phase A: a few dozen of the several hundred) are compacted
        // into an LDS list first, then fetched one per thread side by side -- picked out of the candidate rounds where
        // they stand, every round paid a full row-fetch latency for its two or three scattered lanes
        const float qlo = myqq * kminus;
        if (tid == 0) s_nneed = 0;
        __syncthreads();
        for (int e0 = 0; e0 < c; e0 += 4 * 256) {
            int row[4];
            float ev[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * 256 + tid;
                const size_t sl = (size_t)qi * SR_CAP + slot_of(e < c ? e : c - 1);
                row[u] = cand_i[sl];
                ev[u] = cand_e[sl].x;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e0 + u * 256 + tid < c && __builtin_fmaf(-2.0f, ev[u], qlo) <= thr2)    // lo: no dd[row] needed
                    need_rows[atomicAdd(&s_nneed, 1)] = row[u];
        }
        __syncthreads();
        const int nneed = s_nneed;
        for (int i0 = 0; i0 < nneed; i0 += 256) {
            const bool need = i0 + tid < nneed;
            const int row = need ? need_rows[i0 + tid] : 0;
            const float d = need ? exact(row) : INFINITY;
            top.push(need, d, row, k, lane);
        }
    } else {
        for (int64_t e0 = 0; e0 < n; e0 += 256) {
            const int64_t row = e0 + tid;
            const bool need = row < n;
            const float d = need ? exact(row) : INFINITY;
            top.push(need, d, (int)row, k, lane);
        }
    }
    if (top.pc > 0) top.fold(k, lane);
    block_merge_tops(top, wtop_d, wtop_i, wave, lane, td, ti);
    if (wave == 0 && lane < k) {
        out_d[(size_t)qi * k + lane] = td;
        out_i[(size_t)qi * k + lane] = ti == SR_EMPTY ? (int64_t)-1 : id_base + (int64_t)ti;
    }
}

// ---- merge of P partial lists per query: one workgroup per query ------------------------------------------
// 256 threads sweep the P*k entries 2048 at a time (8 independent loads per thread), keep only entries that are
// valid and <= the running threshold (+inf, tightened by every fold), append them to an LDS buffer,
// and wave 0 folds the buffer into the sorted top list 32 entries at a time with the 64-lane bitonic sort.
constexpr int MG_ITEMS = 8;
constexpr int MG_CAP = 256 * MG_ITEMS + 64;

__global__ __launch_bounds__(256) void search_merge_kernel(const float *__restrict__ part_d,
                                                           const int64_t *__restrict__ part_i, int P, int nq, int k,
                                                           float *__restrict__ out_d,
                                                           int64_t *__restrict__ out_i) {
    __shared__ float top_d[32];
    __shared__ long long top_i[32];
    __shared__ float buf_d[MG_CAP];
    __shared__ long long buf_i[MG_CAP];
    __shared__ int s_cnt;
    __shared__ float s_thr;
    constexpr long long EMPTY = 0x7fffffffffffffffll;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    if (tid < 32) {
        top_d[tid] = INFINITY;
        top_i[tid] = EMPTY;
    }
    if (tid == 0) {
        s_thr = INFINITY;
        s_cnt = 0;
    }
    __syncthreads();
    const int total = P * k;
    for (int base = 0; base < total; base += 256 * MG_ITEMS) {
        const float thr = s_thr;
        float d[MG_ITEMS];
        long long id[MG_ITEMS];
#pragma unroll
        for (int u = 0; u < MG_ITEMS; ++u) {
            const int e = base + u * 256 + tid;
            d[u] = INFINITY;
            id[u] = -1;
            if (e < total) {
                const int p = e / k, j = e - p * k;
                const size_t o = ((size_t)p * nq + q) * k + j;
                id[u] = part_i[o];
                d[u] = part_d[o];
            }
        }
#pragma unroll
        for (int u = 0; u < MG_ITEMS; ++u)
            if (id[u] >= 0 && d[u] <= thr) {
                const int pos = atomicAdd(&s_cnt, 1);
                buf_d[pos] = d[u];
                buf_i[pos] = id[u];
            }
        __syncthreads();
        const int cnt = s_cnt;
        if (tid < 64 && cnt > 0) {      // wave 0 folds the buffer, 32 entries per sort
            float td = lane < 32 ? top_d[lane] : INFINITY;
            long long ti = lane < 32 ? top_i[lane] : EMPTY;
            for (int off = 0; off < cnt; off += 32) {
                float sd = td;
                long long si = ti;
                if (lane >= 32) {
                    const int e = off + lane - 32;
                    sd = e < cnt ? buf_d[e] : INFINITY;
                    si = e < cnt ? buf_i[e] : EMPTY;
                }
                wave_sort64(sd, si, lane);
                td = sd;                     // lanes 0..31 now hold the 32 best so far
                ti = si;
            }
            if (lane < 32) {
                top_d[lane] = td;
                top_i[lane] = ti;
            }
            const float kth = __shfl(td, k - 1);
            if (lane == 0) {
                s_thr = fminf(s_thr, kth);
                s_cnt = 0;
            }
        }
        __syncthreads();
    }
    if (tid < k) {
        const long long i = top_i[tid];
        out_d[(size_t)q * k + tid] = top_d[tid];
        out_i[(size_t)q * k + tid] = i == EMPTY ? (int64_t)-1 : (int64_t)i;
    }
}

struct SearchPlan {
    int qw, rw, qgroups, trows;
    int splits;                   // main pass: grid = (splits, qgroups)
    int64_t rows_per_split;
    int b_splits;                 // pre-pass launch over the first b_rows rows
    int64_t b_rows, b_rows_per_split;
};

static void split_rows(int64_t rows, int trows, int64_t want, int *splits, int64_t *rows_per_split) {
    const int64_t tiles = (rows + trows - 1) / trows;
    if (want > tiles) want = tiles;
    if (want < 1) want = 1;
    const int64_t tps = (tiles + want - 1) / want;        // whole tiles per split
    *rows_per_split = tps * trows;
    *splits = (int)((tiles + tps - 1) / tps);
}

static SearchPlan make_plan(int64_t n, int nq) {
    SearchPlan p;
    p.qw = nq <= 32 ? 1 : (nq <= 64 ? 2 : 4);
    p.rw = 4 / p.qw;
    p.trows = 32 * p.rw;
    p.qgroups = (nq + 32 * p.qw - 1) / (32 * p.qw);
    // two workgroups per CU (256 CUs), three for the 4x1 shape (registers and LDS allow it), when the query
    // groups allow it
    int64_t want = (p.qw == 4 ? 768 : 512) / p.qgroups;
    split_rows(n, p.trows, want < 1 ? 1 : want, &p.splits, &p.rows_per_split);
    // pre-pass: the first max(64k, n/16) rows; >= 32 (split, row-wave) pairs per query so that all 64 groups
    // (pair x half) see rows, and enough workgroups to fill the chip for a handful of tiles each
    p.b_rows = n / 16 > 65536 ? n / 16 : 65536;
    if (p.b_rows > n) p.b_rows = n;
    int64_t bwant = 1024 / p.qgroups;
    const int64_t need = (32 + p.rw - 1) / p.rw;
    if (bwant < need) bwant = need;
    split_rows(p.b_rows, p.trows, bwant, &p.b_splits, &p.b_rows_per_split);
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
    return align256((size_t)nq * sizeof(float)) * 2 + align256((size_t)nq * SR_NSUB * sizeof(int)) +
           align256((size_t)nq * SR_GROUPS * sizeof(int)) + align256((size_t)nq * SR_CAP * sizeof(float)) +
           align256((size_t)nq * SR_CAP * sizeof(int));
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
    char *w = (char *)ws;
    float *qq = (float *)w;                 w += align256((size_t)nq * sizeof(float));
    float *thr = (float *)w;                w += align256((size_t)nq * sizeof(float));
    int *cnt = (int *)w;                    w += align256((size_t)nq * SR_NSUB * sizeof(int));
    int *gmin = (int *)w;                   w += align256((size_t)nq * SR_GROUPS * sizeof(int));
    float *cand_d = (float *)w;             w += align256((size_t)nq * SR_CAP * sizeof(float));
    int *cand_i = (int *)w;
    const int64_t ng = (int64_t)nq * SR_GROUPS;
    hipLaunchKernelGGL(search_init_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, s, q, nq, qq, gmin, cnt);
    const size_t lds = ((size_t)p.trows * SR_LS + p.trows) * sizeof(float);
    const dim3 grid_b(p.b_splits, p.qgroups), grid(p.splits, p.qgroups);
#define SR_LAUNCH(QW)                                                                                               \
    (void)hipFuncSetAttribute((const void *)search_bound_kernel<QW>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                              (int)lds);                                                                            \
    hipLaunchKernelGGL(search_bound_kernel<QW>, grid_b, dim3(256), lds, s, db, db_sqnorm, p.b_rows, q, qq, nq,      \
                       p.b_rows_per_split, gmin);                                                                   \
    hipLaunchKernelGGL(search_thr_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, (const int *)gmin, nq, k, thr);      \
    (void)hipFuncSetAttribute((const void *)search_scan_kernel<QW>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                              (int)lds);                                                                            \
    hipLaunchKernelGGL(search_scan_kernel<QW>, grid, dim3(256), lds, s, db, db_sqnorm, n, q, qq, nq,                \
                       p.rows_per_split, (const float *)thr, cnt, cand_d, cand_i)
    if (p.qw == 1) { SR_LAUNCH(1); }
    else if (p.qw == 2) { SR_LAUNCH(2); }
    else { SR_LAUNCH(4); }
#undef SR_LAUNCH
    GRAFP_CHECK_LAUNCH("search_bound_kernel / search_scan_kernel");
    hipLaunchKernelGGL(search_select_kernel, dim3(nq), dim3(256), 0, s, db, db_sqnorm, n, q, (const float *)qq, nq, k,
                       id_base, (const float *)thr, (const int *)cnt, (const float *)cand_d, (const int *)cand_i,
                       out_dist, out_ids);
    GRAFP_CHECK_LAUNCH("search_select_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_f32_to_bf16(const float *src, int64_t n_elems, void *dst, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(src && dst && n_elems >= 0, "f32_to_bf16: bad arguments");
    if (n_elems == 0) return GRAFP_OK;
    const int64_t nb = (n_elems + 255) / 256;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0, (hipStream_t)stream,
                       src, n_elems, (unsigned short *)dst);
    GRAFP_CHECK_LAUNCH("f32_to_bf16_kernel");
    return GRAFP_OK;
}

namespace grafp {
// launch plan of the bf16 pre-filter path (shared by the workspace size and the entry)
struct PrePlan {
    int qw, rw, nqs, qgroups, b_splits, ngroups;
    int64_t b_rows, b_rps;
};
static PrePlan pre_plan(int64_t n, int nq) {
    PrePlan p;
    // measured crossovers (1M x 128): the 1x4 shape only pays for a handful of queries (nq=16: 136 vs 127 us, nq=32:
    // 168 vs 143 us for the 2x2 shape)
    p.qw = nq <= 64 ? 2 : 4;                            // (the 1x4 shape went with the 64-row ring stages)
    p.rw = 4 / p.qw;
    // two query sets per wave (256 queries per workgroup, two workgroups per CU) from 1024 queries on: every row fragment
    // read from LDS feeds two independent MFMA chains and a tile's DMA, barrier and loop overhead is shared by twice the
    // MFMAs (1M x 128, round 4, after the loop's address arithmetic went into immediates: 4096 queries 1.32 -> 1.20 ms,
    // 2048 0.69 -> 0.645, 1024 0.380 -> 0.365, 512 0.231 -> 0.237; before that the two forms were level)
    // THREE sets per wave (round 6; 249 registers, still two workgroups per CU): a third fewer fragment reads per MFMA.
    // Same-process A/B (1M x 128, k = 20, scratch/nqs3_check.py): 4096 queries 1.108 -> 1.080 ms, 2500 0.734 -> 0.704, but
    // 2048 0.580 -> 0.601 and 1024 0.318 -> 0.332 -- the gain (~8 %) is paid back by the idle query slots of the last
    // 384-query group, so: from 1536 queries, when the three-set form pads at most 6 % more than the two-set form
    int nqs_auto = nq >= 1024 ? 2 : 1;
    if (nq >= 1536) {
        const int64_t pad3 = (int64_t)((nq + 383) / 384) * 384 - nq, pad2 = (int64_t)((nq + 255) / 256) * 256 - nq;
        if ((pad3 - pad2) * 100 <= (int64_t)6 * nq) nqs_auto = 3;
    }
    p.nqs = p.qw == 4 ? GRAFP_TUNE_INT("GRAFP_SEARCH_NQS", nqs_auto) : 1;
    if (p.nqs < 1 || p.nqs > 3) p.nqs = 1;
    p.qgroups = (nq + 32 * p.qw * p.nqs - 1) / (32 * p.qw * p.nqs);
    p.b_rows = n / 16 > 65536 ? n / 16 : 65536;
    if (p.b_rows > n) p.b_rows = n;
    // pre-pass workgroups: ONE per CU, four or more ring stages each (round 4; 1024 one-tile workgroups paid their prologue
    // -- the query operand, the ring fill -- for a single tile: 41 queries 0.092 -> 0.079 ms, 1 query 0.080 -> 0.070, no
    // batch size slower); never fewer than the 64 (split, row-wave, half) groups the threshold kernel selects from
    int64_t bwant = GRAFP_TUNE_INT("GRAFP_SEARCH_BWANT", 256) / p.qgroups;
    const int64_t bneed = (32 + p.rw - 1) / p.rw;
    if (bwant < bneed) bwant = bneed;
    split_rows(p.b_rows, SB_TR, bwant, &p.b_splits, &p.b_rps);
    p.ngroups = p.b_splits * p.rw * 2;
    return p;
}
}  // namespace grafp

extern "C" size_t grafp_knn_search_pre_workspace(int64_t n, int nq, int d, int k) {
    using namespace grafp;
    if (n <= 0 || nq <= 0 || d != SR_D || k < 1) return 0;
    const PrePlan p = pre_plan(n, nq);
    return align256((size_t)nq * sizeof(float)) * 2 + align256((size_t)nq * SR_NSUB * sizeof(int)) +
           align256((size_t)nq * p.ngroups * sizeof(float)) + align256((size_t)nq * SR_CAP * sizeof(int)) +
           align256((size_t)nq * SR_CAP * sizeof(float2));
}

extern "C" int grafp_knn_search_l2_pre(const float *db, const void *db_bf16, const float *db_sqnorm, int64_t n,
                                       const float *q, int nq, int d, int k, int64_t id_base, float *out_dist,
                                       int64_t *out_ids, void *ws, size_t ws_bytes, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(db && db_bf16 && db_sqnorm && q && out_dist && out_ids, "knn_search_pre: null pointer");
    GRAFP_REQUIRE(d == SR_D, "knn_search_pre: d=%d unsupported (fingerprints are 128-d)", d);
    GRAFP_REQUIRE(n >= 1 && n < 0x7fffffffll && nq >= 1, "knn_search_pre: bad n=%lld nq=%d", (long long)n, nq);
    GRAFP_REQUIRE(k >= 1 && k <= GRAFP_SEARCH_MAX_K, "knn_search_pre: k=%d not in [1, %d]", k, GRAFP_SEARCH_MAX_K);
    GRAFP_REQUIRE((((uintptr_t)db | (uintptr_t)db_bf16) & 15) == 0 && ((uintptr_t)q & 3) == 0,
                  "knn_search_pre: db / db_bf16 must be 16-byte aligned");
    const size_t need = grafp_knn_search_pre_workspace(n, nq, d, k);
    if (!ws || ws_bytes < need) {
        set_error("knn_search_pre: workspace %zu bytes < required %zu", ws_bytes, need);
        return GRAFP_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const PrePlan pl = pre_plan(n, nq);
    const int qw = pl.qw, nqs = pl.nqs, qgroups = pl.qgroups, b_splits = pl.b_splits;
    const int64_t b_rows = pl.b_rows, b_rps = pl.b_rps;
    int splits;
    int64_t rps;
    int64_t want = GRAFP_TUNE_INT("GRAFP_SEARCH_WANT", nqs >= 2 ? 256 * SB_WGS_NQS2 : 768) / qgroups;
    if (want < 1) want = 1;
    // Large batches scan in two parts.  The pre-pass bound (k-th smallest of 64 group minima over n/16 rows) lets a few
    // hundred rows per query through; the first n/4 rows are scanned with it, the k-th smallest UPPER bound among their
    // candidates (at least k rows are truly that close -- the select kernel's own phase A, run early: 35 us at 4096
    // queries) replaces it, and the other three quarters are scanned with the tighter one: their blocks hold a hit a
    // third as often and the select kernel sorts ~40 % fewer candidates.  Measured (1M x 128, k = 20, same process,
    // 30 repetitions): 4096 queries 1.46 -> 1.39 ms, 2048 0.765 -> 0.732, 1024 0.428 -> 0.412; a first part of n/3,
    // n/6, n/8 gives 1.40, 1.40, 1.42 ms.  Below ~700 queries the two extra launches cost what the bound saves (256
    // queries: 0.171 -> 0.183 ms): one part.
    const int first_div = GRAFP_TUNE_INT("GRAFP_SEARCH_FIRST_DIV", 4);
    int64_t n_first = 0;
    if (nq >= GRAFP_TUNE_INT("GRAFP_SEARCH_TWO_PART_NQ", 768) && first_div > 1 && n / first_div >= 65536)
        n_first = (n / first_div) / SB_TR * SB_TR;
    int a_splits = 1;
    int64_t a_rps = SB_TR;
    if (n_first > 0) split_rows(n_first, SB_TR, want, &a_splits, &a_rps);
    split_rows(n - n_first, SB_TR, want, &splits, &rps);
    char *w = (char *)ws;
    float *qq = (float *)w;                 w += align256((size_t)nq * sizeof(float));
    float *thr = (float *)w;                w += align256((size_t)nq * sizeof(float));
    int *cnt = (int *)w;                    w += align256((size_t)nq * SR_NSUB * sizeof(int));
    float *gmax = (float *)w;               w += align256((size_t)nq * pl.ngroups * sizeof(float));
    int *cand_i = (int *)w;                 w += align256((size_t)nq * SR_CAP * sizeof(int));
    float2 *cand_e = (float2 *)w;
    const size_t lds = (size_t)SB_NS * SB_STAGE;
    const dim3 grid_b(b_splits, qgroups), grid_a(a_splits, qgroups), grid(splits, qgroups);
    const unsigned short *dbh = (const unsigned short *)db_bf16;
#define SB_LAUNCH(QW, NQS)                                                                                          \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(search_bound_bf16_kernel<QW, NQS>), grid_b, dim3(256), lds, s, dbh,          \
                       db_sqnorm, b_rows, q, nq, b_rps, gmax, pl.ngroups);                                          \
    hipLaunchKernelGGL(search_thr_pre_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, q, nq, (const float *)gmax,      \
                       pl.ngroups, k, qq, thr, cnt);                                                                \
    if (n_first > 0) {                                                                                              \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(search_scan_bf16_kernel<QW, NQS>), grid_a, dim3(256), lds, s, dbh,       \
                           db_sqnorm, (int64_t)0, n_first, q, (const float *)qq, nq, a_rps, (const float *)thr,     \
                           cnt, cand_i, cand_e);                                                                    \
        hipLaunchKernelGGL(search_select_exact_kernel<true>, dim3(nq), dim3(256), 0, s, db, db_sqnorm, n, q,        \
                           (const float *)qq, nq, k, id_base, thr, (const int *)cnt, (const int *)cand_i,           \
                           (const float2 *)cand_e, out_dist, out_ids);                                               \
    }                                                                                                               \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(search_scan_bf16_kernel<QW, NQS>), grid, dim3(256), lds, s, dbh, db_sqnorm,  \
                       n_first, n, q, (const float *)qq, nq, rps, (const float *)thr, cnt, cand_i, cand_e)
    if (qw == 2) { SB_LAUNCH(2, 1); }
    else if (nqs == 1) { SB_LAUNCH(4, 1); }
    else if (nqs == 3) { SB_LAUNCH(4, 3); }
    else { SB_LAUNCH(4, 2); }
#undef SB_LAUNCH
    GRAFP_CHECK_LAUNCH("search_bound_bf16_kernel / search_scan_bf16_kernel");
    hipLaunchKernelGGL(search_select_exact_kernel<false>, dim3(nq), dim3(256), 0, s, db, db_sqnorm, n, q,
                       (const float *)qq, nq, k, id_base, thr, (const int *)cnt, (const int *)cand_i,
                       (const float2 *)cand_e, out_dist, out_ids);
    GRAFP_CHECK_LAUNCH("search_select_exact_kernel");
    return GRAFP_OK;
}

#ifdef GRAFP_MEASURE
// measurement builds only (tools/search_abl.py): the pre-pass loop over ALL n rows with parts of it removed
extern "C" int grafp_measure_search_loop(const void *db_bf16, const float *db_sqnorm, int64_t n, const float *q,
                                         const float *qq, int nq, int abl, int nqs, int *gmin, grafp_stream_t stream) {
    (void)qq;
    using namespace grafp;
    const int qgroups = (nq + 128 * nqs - 1) / (128 * nqs);
    int splits;
    int64_t rps;
    int64_t want = 768 / qgroups;
    split_rows(n, SB_TR, want < 1 ? 1 : want, &splits, &rps);
    const size_t lds = (size_t)SB_NS * SB_STAGE;
    const dim3 grid(splits, qgroups);
    const unsigned short *dbh = (const unsigned short *)db_bf16;
    hipStream_t s = (hipStream_t)stream;
#define ABL_CASE(A)                                                                                                 \
    case A:                                                                                                         \
        if (nqs == 2)                                                                                               \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(search_bound_bf16_kernel<4, 2, A>), grid, dim3(256), lds, s, dbh,    \
                               db_sqnorm, n, q, nq, rps, (float *)gmin, 2 * splits);                                \
        else                                                                                                        \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(search_bound_bf16_kernel<4, 1, A>), grid, dim3(256), lds, s, dbh,    \
                               db_sqnorm, n, q, nq, rps, (float *)gmin, 2 * splits);                                \
        break
    switch (abl) {
        ABL_CASE(0); ABL_CASE(1); ABL_CASE(2); ABL_CASE(3); ABL_CASE(4); ABL_CASE(8); ABL_CASE(9); ABL_CASE(10);
        ABL_CASE(11); ABL_CASE(12); ABL_CASE(14); ABL_CASE(15); ABL_CASE(6); ABL_CASE(7);
        default: set_error("measure_search_loop: abl=%d not built", abl); return GRAFP_ERR_ARG;
    }
#undef ABL_CASE
    GRAFP_CHECK_LAUNCH("search_bound_bf16_kernel (ablation)");
    return GRAFP_OK;
}
#endif

extern "C" int grafp_merge_topk(const float *part_dist, const int64_t *part_ids, int P, int nq, int k, float *out_dist,
                                int64_t *out_ids, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(part_dist && part_ids && out_dist && out_ids, "merge_topk: null pointer");
    GRAFP_REQUIRE(P >= 1 && nq >= 1 && k >= 1 && k <= GRAFP_SEARCH_MAX_K, "merge_topk: bad P=%d nq=%d k=%d", P, nq, k);
    hipLaunchKernelGGL(search_merge_kernel, dim3(nq), dim3(256), 0, (hipStream_t)stream, part_dist, part_ids, P, nq, k,
                       out_dist, out_ids);
    GRAFP_CHECK_LAUNCH("search_merge_kernel");
    return GRAFP_OK;
}
