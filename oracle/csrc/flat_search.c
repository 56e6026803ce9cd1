/* oracle/csrc/flat_search.c -- TEST INFRASTRUCTURE (see oracle/__init__.py).
 *
 * Restatement of faiss==1.7.2 IndexFlatL2.add/search as used at
 * /root/reference/eval.py:54,212-213,269-270 (faiss is a pinned, un-vendored dependency:
 * /root/reference/requirements.txt:7).  Published semantics: exact squared-L2 k nearest, ascending
 * distance, int64 ids in insertion order, id -1 for missing results.  faiss's own accumulation and
 * tie order are unspecified (BLAS path for nq >= 20, SIMD path below); this file FIXES them so the
 * HIP kernel can match bit-for-bit:
 *
 *   qq_i = fmaf-chain over c of q[i][c]^2 ; dd_j likewise ; ip_ij = fmaf-chain of q[i][c]*db[j][c]
 *   (all chains c ascending, start 0)
 *   dis_ij = (qq_i + dd_j) - 2*ip_ij ; negative -> 0         (the norm-expansion form faiss uses)
 *   result = k smallest by (dis, id) lexicographic.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define QB 64            /* queries processed together (vector lanes run over queries) */

void oracle_row_sqnorm(const float *m, int64_t n, int d, float *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < n; ++j) {
        float s = 0.0f;
        for (int c = 0; c < d; ++c) s = fmaf(m[j * d + c], m[j * d + c], s);
        out[j] = s;
    }
}

static inline int lex_less(float d1, int64_t i1, float d2, int64_t i2)
{
    return d1 < d2 || (d1 == d2 && i1 < i2);
}

static void insert(float *bd, int64_t *bi, int k, float d, int64_t id)
{
    if (!lex_less(d, id, bd[k - 1], bi[k - 1])) return;
    int p = k - 1;
    while (p > 0 && lex_less(d, id, bd[p - 1], bi[p - 1])) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; --p; }
    bd[p] = d; bi[p] = id;
}

/* db (n,d), q (nq,d) row-major f32; out_d (nq,k) f32, out_i (nq,k) int64.  id_base is added to ids. */
int oracle_flat_search_l2(const float *db, int64_t n, const float *q, int nq, int d, int k,
                          int64_t id_base, float *out_d, int64_t *out_i)
{
    if (k < 1 || d < 1) return -1;
    float *dd = (float *)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
    float *qq = (float *)malloc((size_t)(nq > 0 ? nq : 1) * sizeof(float));
    oracle_row_sqnorm(db, n, d, dd);
    oracle_row_sqnorm(q, nq, d, qq);
    for (int64_t t = 0; t < (int64_t)nq * k; ++t) { out_d[t] = INFINITY; out_i[t] = -1; }

    for (int q0 = 0; q0 < nq; q0 += QB) {
        const int nb = nq - q0 < QB ? nq - q0 : QB;
        float *qt = (float *)calloc((size_t)d * QB, sizeof(float));       /* (d, QB) transposed block */
        for (int i = 0; i < nb; ++i)
            for (int c = 0; c < d; ++c) qt[(size_t)c * QB + i] = q[(size_t)(q0 + i) * d + c];
#pragma omp parallel
        {
            float *bd = (float *)malloc((size_t)QB * k * sizeof(float));
            int64_t *bi = (int64_t *)malloc((size_t)QB * k * sizeof(int64_t));
            for (int t = 0; t < QB * k; ++t) { bd[t] = INFINITY; bi[t] = INT64_MAX; }
            float ip[QB];
#pragma omp for schedule(static)
            for (int64_t j = 0; j < n; ++j) {
                const float *row = db + j * d;
                for (int i = 0; i < QB; ++i) ip[i] = 0.0f;
                for (int c = 0; c < d; ++c) {
                    const float a = row[c];
                    const float *qc = qt + (size_t)c * QB;
                    for (int i = 0; i < QB; ++i) ip[i] = fmaf(qc[i], a, ip[i]);
                }
                for (int i = 0; i < nb; ++i) {
                    float dis = (qq[q0 + i] + dd[j]) - 2.0f * ip[i];
                    if (dis < 0.0f) dis = 0.0f;
                    if (dis <= bd[(size_t)i * k + k - 1]) insert(bd + (size_t)i * k, bi + (size_t)i * k, k, dis, id_base + j);
                }
            }
#pragma omp critical
            {
                for (int i = 0; i < nb; ++i)
                    for (int t = 0; t < k; ++t)
                        if (bi[(size_t)i * k + t] != INT64_MAX) {
                            float *od = out_d + (size_t)(q0 + i) * k; int64_t *oi = out_i + (size_t)(q0 + i) * k;
                            /* treat empty output slots (-1) as +inf/INT64_MAX for ordering */
                            float d1 = bd[(size_t)i * k + t]; int64_t i1 = bi[(size_t)i * k + t];
                            int p = k - 1;
                            int64_t lastid = oi[p] < 0 ? INT64_MAX : oi[p];
                            if (!lex_less(d1, i1, od[p], lastid)) continue;
                            while (p > 0) {
                                int64_t pid = oi[p - 1] < 0 ? INT64_MAX : oi[p - 1];
                                if (!lex_less(d1, i1, od[p - 1], pid)) break;
                                od[p] = od[p - 1]; oi[p] = oi[p - 1]; --p;
                            }
                            od[p] = d1; oi[p] = i1;
                        }
            }
            free(bd); free(bi);
        }
        free(qt);
    }
    free(dd); free(qq);
    return 0;
}

/* Merge P partial lists (P, nq, k) [dist asc, id; id<0 = empty] into (nq, k). */
void oracle_merge_topk(const float *pd, const int64_t *pi, int P, int nq, int k, float *out_d, int64_t *out_i)
{
    for (int i = 0; i < nq; ++i) {
        float *od = out_d + (size_t)i * k; int64_t *oi = out_i + (size_t)i * k;
        for (int t = 0; t < k; ++t) { od[t] = INFINITY; oi[t] = INT64_MAX; }
        for (int p = 0; p < P; ++p)
            for (int t = 0; t < k; ++t) {
                const int64_t id = pi[((size_t)p * nq + i) * k + t];
                if (id >= 0) insert(od, oi, k, pd[((size_t)p * nq + i) * k + t], id);
            }
        for (int t = 0; t < k; ++t) if (oi[t] == INT64_MAX) oi[t] = -1;
    }
}
