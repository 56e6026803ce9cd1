// seq_rerank.hip -- sequence-level rerank of the segment search results, gfx950.
//
// Replaces the double Python loop of /root/reference/eval.py:262-290.  For one (test id, query length) item with
// ql query segments q[0..ql) whose top-k database ids are already known (one batched search for all items):
//   * offset compensation (:273-274): candidate start id = id - t for a hit of segment t;
//   * unique non-negative candidates (:277);
//   * score(cid) = mean_t <q[t], index[cid + t]> over the min(ql, n - cid) rows that exist (:280-287 -- np.diag of a
//     non-square product silently uses the shorter side);
//   * the `top` best candidates, score descending, lowest id first among equal scores (:290; the reference's
//     argsort is not stable, so equal scores are unordered there).
// One workgroup per item: the item's query rows and candidate keys live in LDS; candidates are de-duplicated with a
// block-wide bitonic sort of 64-bit keys, scored one per half-wave with coalesced 512-byte row reads out of the
// resident database, and ranked with a second sort of (~score, id) keys.
// Arithmetic order (restated in oracle/csrc/seq_rerank.c): lane l of 32 owns dims 4l..4l+3 and runs ONE fmaf chain
// over (t ascending, e = 0..3); the 32 lane sums are combined by the butterfly s = 16, 8, 4, 2, 1 (x[l] + x[l ^ s]);
// score = sum / rows (IEEE division).
#include <math.h>

#include "common.h"

namespace grafp {

constexpr int RR_D = 128;
constexpr int RR_MAX_LEN = 64;       // query segments per item
constexpr int RR_MAX_CAND = 2048;    // ql * k
constexpr unsigned long long RR_NONE = ~0ull;

// ascending bitonic sort of P (power of two) 64-bit keys in LDS by the whole workgroup
__device__ __forceinline__ void block_sort_u64(unsigned long long *keys, int P, int tid, int nthreads) {
    for (int k2 = 2; k2 <= P; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int e = tid; e < P; e += nthreads) {
                const int partner = e ^ j;
                if (partner > e) {
                    const unsigned long long a = keys[e], b = keys[partner];
                    const bool asc = (e & k2) == 0;
                    if ((a > b) == asc) {
                        keys[e] = b;
                        keys[partner] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// monotone map f32 -> u32 (larger float = larger integer), and back
__device__ __forceinline__ unsigned int f32_ord(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_f32(unsigned int o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// Sharded use: `recon` holds global rows [row_base, row_base + n_rows) of an n-row index; only candidates whose start id
// lies in [id_lo, id_hi) are scored here (the shard that owns the start row, which also holds a halo of the next
// shard's first rows so that whole sequences are local); the per-shard top lists are merged by score afterwards.
__global__ __launch_bounds__(256) void seq_rerank_kernel(const float *__restrict__ recon, int64_t n,
                                                         int64_t row_base, int64_t n_rows, int64_t id_lo,
                                                         int64_t id_hi, const float *__restrict__ q_rows,
                                                         const int64_t *__restrict__ ids, int k,
                                                         const int64_t *__restrict__ item_row,
                                                         const int *__restrict__ item_len, int max_len, int top,
                                                         int64_t *__restrict__ out_ids,
                                                         float *__restrict__ out_scores) {
    __shared__ __attribute__((aligned(16))) float sq[RR_MAX_LEN * RR_D];
    __shared__ unsigned long long keys[RR_MAX_CAND];
    __shared__ int cand[RR_MAX_CAND];
    __shared__ int s_ncand;
    const int item = blockIdx.x, tid = threadIdx.x;
    const int64_t r0 = item_row[item];
    const int ql = item_len[item] < max_len ? item_len[item] : max_len;   // max_len * k <= RR_MAX_CAND (host-checked)
    const int total = ql * k;
    int P = 64;
    while (P < total) P <<= 1;
    if (tid == 0) s_ncand = 0;
    // query rows -> LDS (coalesced), candidate start ids -> keys
    {
        const float4 *src = reinterpret_cast<const float4 *>(q_rows + r0 * RR_D);
        float4 *dst = reinterpret_cast<float4 *>(sq);
        for (int i = tid; i < ql * (RR_D / 4); i += 256) dst[i] = src[i];
    }
    for (int e = tid; e < P; e += 256) {
        unsigned long long key = RR_NONE;
        if (e < total) {
            const int t = e / k;
            const int64_t id = ids[(r0 + t) * k + (e - t * k)];
            const int64_t c = id - t;                         // eval.py:273-274
            if (id >= 0 && c >= 0 && c >= id_lo && c < id_hi) key = (unsigned long long)c;
        }
        keys[e] = key;
    }
    __syncthreads();
    block_sort_u64(keys, P, tid, 256);
    // unique (eval.py:277); the order of the compacted list is irrelevant, ranking re-sorts with the id in the key
    for (int e = tid; e < P; e += 256) {
        const unsigned long long key = keys[e];
        if (key != RR_NONE && (e == 0 || keys[e - 1] != key)) cand[atomicAdd(&s_ncand, 1)] = (int)key;
    }
    __syncthreads();
    const int ncand = s_ncand;
    int P2 = 64;
    while (P2 < ncand) P2 <<= 1;
    __syncthreads();                                           // keys are rewritten below
    for (int e = tid; e < P2; e += 256) keys[e] = RR_NONE;
    __syncthreads();
    // scores: one candidate per half-wave at a time
    const int hw = tid >> 5, l = tid & 31;
    const float4 *rc4 = reinterpret_cast<const float4 *>(recon);
    const float4 *sq4 = reinterpret_cast<const float4 *>(sq);
    for (int c = hw; c < ncand; c += 8) {
        const int cid = cand[c];
        const int64_t left = n - (int64_t)cid;
        int m = left < ql ? (int)left : ql;
        const int64_t have = row_base + n_rows - (int64_t)cid;        // rows of this sequence present locally
        if (have < m) m = have > 0 ? (int)have : 0;                   // (never with a halo of max_len - 1 rows)
        float acc = 0.0f;
        for (int t = 0; t < m; ++t) {
            const float4 r = rc4[((int64_t)cid - row_base + t) * (RR_D / 4) + l];
            const float4 qv = sq4[t * (RR_D / 4) + l];
            acc = __builtin_fmaf(qv.x, r.x, acc);
            acc = __builtin_fmaf(qv.y, r.y, acc);
            acc = __builtin_fmaf(qv.z, r.z, acc);
            acc = __builtin_fmaf(qv.w, r.w, acc);
        }
#pragma unroll
        for (int s = 16; s > 0; s >>= 1) acc += __shfl_xor(acc, s);      // stays inside the 32-lane half
        const float score = acc / (float)m;
        if (l == 0) keys[c] = ((unsigned long long)(~f32_ord(score)) << 32) | (unsigned int)cid;
    }
    __syncthreads();
    block_sort_u64(keys, P2, tid, 256);
    if (tid < top) {
        const unsigned long long key = tid < P2 ? keys[tid] : RR_NONE;
        const bool have = key != RR_NONE;
        out_ids[(size_t)item * top + tid] = have ? (int64_t)(key & 0xffffffffull) : (int64_t)-1;
        out_scores[(size_t)item * top + tid] = have ? ord_f32(~(unsigned int)(key >> 32)) : -INFINITY;
    }
}

}  // namespace grafp

extern "C" int grafp_seq_rerank_shard_f32(const float *index_rows, int64_t n_rows, int64_t row_base, int64_t n,
                                          int64_t id_lo, int64_t id_hi, const float *q_rows, int64_t n_qrows,
                                          const int64_t *topk_ids, int k, const int64_t *item_row,
                                          const int *item_len, int n_items, int max_len, int top, int64_t *out_ids,
                                          float *out_scores, grafp_stream_t stream) {
    using namespace grafp;
    GRAFP_REQUIRE(index_rows && q_rows && topk_ids && item_row && item_len && out_ids && out_scores,
                  "seq_rerank: null pointer");
    GRAFP_REQUIRE(n >= 1 && n < 0x7fffffffll && n_qrows >= 1 && n_items >= 0, "seq_rerank: bad n=%lld n_qrows=%lld",
                  (long long)n, (long long)n_qrows);
    GRAFP_REQUIRE(n_rows >= 1 && row_base >= 0 && row_base + n_rows <= n && id_lo >= row_base && id_lo <= id_hi &&
                  id_hi <= row_base + n_rows, "seq_rerank: bad shard rows [%lld, +%lld) ids [%lld, %lld) of %lld",
                  (long long)row_base, (long long)n_rows, (long long)id_lo, (long long)id_hi, (long long)n);
    GRAFP_REQUIRE(k >= 1 && max_len >= 1 && max_len <= RR_MAX_LEN && (int64_t)max_len * k <= RR_MAX_CAND,
                  "seq_rerank: max_len=%d k=%d exceed %d segments / %d candidates per item", max_len, k, RR_MAX_LEN,
                  RR_MAX_CAND);
    GRAFP_REQUIRE(top >= 1 && top <= 64, "seq_rerank: top=%d not in [1, 64]", top);
    GRAFP_REQUIRE((((uintptr_t)index_rows | (uintptr_t)q_rows) & 15) == 0, "seq_rerank: rows must be 16-byte aligned");
    if (n_items == 0) return GRAFP_OK;
    hipLaunchKernelGGL(seq_rerank_kernel, dim3(n_items), dim3(256), 0, (hipStream_t)stream, index_rows, n, row_base,
                       n_rows, id_lo, id_hi, q_rows, topk_ids, k, item_row, item_len, max_len, top, out_ids, out_scores);
    GRAFP_CHECK_LAUNCH("seq_rerank_kernel");
    return GRAFP_OK;
}

extern "C" int grafp_seq_rerank_f32(const float *index_rows, int64_t n, const float *q_rows, int64_t n_qrows,
                                    const int64_t *topk_ids, int k, const int64_t *item_row, const int *item_len,
                                    int n_items, int max_len, int top, int64_t *out_ids, float *out_scores,
                                    grafp_stream_t stream) {
    return grafp_seq_rerank_shard_f32(index_rows, n, 0, n, 0, n, q_rows, n_qrows, topk_ids, k, item_row, item_len,
                                      n_items, max_len, top, out_ids, out_scores, stream);
}
