/* oracle/csrc/seq_rerank.c -- sequence-level rerank, CPU restatement.  TEST INFRASTRUCTURE (see oracle/__init__.py).
 *
 * Follows /root/reference/eval.py:272-290 for one (test id, query length) item:
 *   :273-274  I[offset, :] -= offset                       candidate start id = id - t
 *   :277      candidates = np.unique(I[I >= 0])            (ascending)
 *   :280-287  score = np.mean(np.diag(np.dot(q, index[cid:cid+sl].T)))    over min(len(q), rows that exist) terms
 *   :290      pred = candidates[np.argsort(-scores)[:10]]
 * The reference evaluates the products in float32 BLAS order, which is not reproducible; this file FIXES an order
 * (the one the HIP kernel uses, grafp_amd/csrc/seq_rerank.hip) so that kernel and oracle agree bit for bit:
 *   32 lanes; lane l owns dims 4l..4l+3 and runs one fmaf chain over (t ascending, e = 0..3);
 *   the lane sums are combined by the butterfly s = 16, 8, 4, 2, 1: x[l] = x[l] + x[l ^ s] for all l at once;
 *   score = x[0] / rows  (IEEE f32 division).
 * Ranking: score descending, lowest id first among equal scores (np.argsort's default sort is not stable, so the
 * reference leaves equal scores unordered; oracle/retrieval.py uses the same rule with kind="stable").
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct { float score; int64_t id; } cand_t;

static int cmp_id(const void *a, const void *b)
{
    const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

static int cmp_rank(const void *a, const void *b)
{
    const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
    if (x->score > y->score) return -1;
    if (x->score < y->score) return 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}

static float seq_score(const float *q, const float *index_rows, int64_t cid, int rows, int d)
{
    float x[32], y[32];
    for (int l = 0; l < 32; ++l) x[l] = 0.0f;
    for (int t = 0; t < rows; ++t) {
        const float *qr = q + (size_t)t * d, *rr = index_rows + (size_t)(cid + t) * d;
        for (int l = 0; l < 32; ++l)
            for (int e = 0; e < 4; ++e) x[l] = fmaf(qr[4 * l + e], rr[4 * l + e], x[l]);
    }
    for (int s = 16; s > 0; s >>= 1) {
        for (int l = 0; l < 32; ++l) y[l] = x[l] + x[l ^ s];
        for (int l = 0; l < 32; ++l) x[l] = y[l];
    }
    return x[0] / (float)rows;
}

/* index_rows (n,128); q_rows (n_qrows,128); topk_ids (n_qrows,k); item i = rows [item_row[i], +item_len[i]).
 * out_ids / out_scores (n_items, top): -1 / -inf padded.  Returns 0, or -1 on bad arguments. */
int oracle_seq_rerank(const float *index_rows, int64_t n, const float *q_rows, const int64_t *topk_ids, int k,
                      const int64_t *item_row, const int32_t *item_len, int n_items, int top, int64_t *out_ids,
                      float *out_scores)
{
    const int d = 128;
    if (n < 1 || k < 1 || top < 1) return -1;
    for (int it = 0; it < n_items; ++it) {
        const int64_t r0 = item_row[it];
        const int ql = item_len[it];
        int64_t *c = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ql * k + 1));
        int nc = 0;
        for (int t = 0; t < ql; ++t)
            for (int j = 0; j < k; ++j) {
                const int64_t id = topk_ids[(size_t)(r0 + t) * k + j];
                if (id >= 0 && id - t >= 0) c[nc++] = id - t;
            }
        qsort(c, (size_t)nc, sizeof(int64_t), cmp_id);
        int nu = 0;
        for (int i = 0; i < nc; ++i)
            if (i == 0 || c[i] != c[i - 1]) c[nu++] = c[i];
        cand_t *cs = (cand_t *)malloc(sizeof(cand_t) * (size_t)(nu + 1));
        for (int i = 0; i < nu; ++i) {
            const int64_t left = n - c[i];
            const int rows = left < ql ? (int)left : ql;
            cs[i].id = c[i];
            cs[i].score = seq_score(q_rows + (size_t)r0 * d, index_rows, c[i], rows, d);
        }
        qsort(cs, (size_t)nu, sizeof(cand_t), cmp_rank);
        for (int j = 0; j < top; ++j) {
            out_ids[(size_t)it * top + j] = j < nu ? cs[j].id : -1;
            out_scores[(size_t)it * top + j] = j < nu ? cs[j].score : -INFINITY;
        }
        free(c);
        free(cs);
    }
    return 0;
}
