/* oracle/csrc/ivfpq.c -- TEST INFRASTRUCTURE (see oracle/__init__.py).
 *
 * Restatement of the IVF-PQ index the published protocol searches with: faiss==1.7.2
 * IndexIVFPQ(IndexFlatL2(d), d, 64, 64, 8), nprobe = 20 (/root/reference/eval.py:65-69,122; test_fp.py:276 defaults to
 * it).  faiss is a pinned, un-vendored dependency (/root/reference/requirements.txt:7): PARITY UNPINNED against faiss
 * itself.  The published algorithm (inverted file over a k-means coarse quantiser, product quantisation of the residuals,
 * asymmetric distances at search time) leaves accumulation order, tie order and the k-means initialisation open; this file
 * FIXES them so that csrc/ivfpq.hip can match bit for bit:
 *
 *   residual r = x[row] - base[base_idx[row]]  (base == NULL: r = x[row]); G sub-spaces of d = D / G dims
 *   dist(row, g, j) = fmaf chain over c ascending of (r[g d + c] - cent[g][j][c])^2, start 0; argmin = lowest j on ties
 *   k-means: cent <- residuals of rows init_rows[0..k); niter times { assign; per cluster the f32 sum of its residuals
 *            in 1024-row chunks (row order), chunk sums added in chunk order; cent = sum / (float)count, empty keeps }
 *   probe:   the nprobe smallest (dist, list id), dist = fmaf chain of (q_c - centroid_c)^2
 *   search:  estimate(code) = sum over m ascending (plain f32 adds) of table[m][code[m]],
 *            table[m][j] = fmaf chain over e of ((q - centroid)[m dsub + e] - codebook[m][j][e])^2;
 *            result = k smallest by (estimate, id), ids = insertion order; (+inf, -1) when fewer exist.
 * What pins the definition itself: ADC(q, code) == ||q - reconstruct(code)||^2 (tests/test_oracle.py, oracle/ivfpq.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KM_CHUNK 1024

static inline float resid(const float *x, int64_t row, int D, const float *base, const int32_t *base_idx, int col)
{
    const float v = x[row * D + col];
    return base ? v - base[(int64_t)base_idx[row] * D + col] : v;
}

void oracle_pq_assign(const float *x, int64_t n, int D, int G, const float *base, const int32_t *base_idx,
                      const float *cent, int k, int32_t *out)
{
    const int d = D / G;
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < n; ++row) {
        for (int g = 0; g < G; ++g) {
            float best = INFINITY;
            int bj = 0;
            for (int j = 0; j < k; ++j) {
                const float *cj = cent + ((int64_t)g * k + j) * d;
                float acc = 0.0f;
                for (int c = 0; c < d; ++c) {
                    const float diff = resid(x, row, D, base, base_idx, g * d + c) - cj[c];
                    acc = fmaf(diff, diff, acc);
                }
                if (acc < best) {
                    best = acc;
                    bj = j;
                }
            }
            out[row * G + g] = bj;
        }
    }
}

int oracle_kmeans(const float *x, int64_t n, int D, int G, const float *base, const int32_t *base_idx,
                  const int64_t *init_rows, int k, int niter, float *cent)
{
    const int d = D / G;
    const int64_t ne = (int64_t)G * k * d;
    const int nchunks = (int)((n + KM_CHUNK - 1) / KM_CHUNK);
    int32_t *asg = malloc((size_t)n * G * sizeof(int32_t));
    float *partial = malloc((size_t)nchunks * ne * sizeof(float));
    int32_t *pcnt = malloc((size_t)nchunks * G * k * sizeof(int32_t));
    if (!asg || !partial || !pcnt) {
        free(asg); free(partial); free(pcnt);
        return -1;
    }
    for (int64_t e = 0; e < ne; ++e) {
        const int c = (int)(e % d), j = (int)((e / d) % k), g = (int)(e / ((int64_t)d * k));
        cent[e] = resid(x, init_rows[j], D, base, base_idx, g * d + c);
    }
    for (int it = 0; it < niter; ++it) {
        oracle_pq_assign(x, n, D, G, base, base_idx, cent, k, asg);
#pragma omp parallel for schedule(static) collapse(2)
        for (int ch = 0; ch < nchunks; ++ch) {
            for (int g = 0; g < G; ++g) {
                const int64_t lo = (int64_t)ch * KM_CHUNK, hi = lo + KM_CHUNK < n ? lo + KM_CHUNK : n;
                float *po = partial + ((int64_t)ch * G + g) * k * d;
                int32_t *pc = pcnt + ((int64_t)ch * G + g) * k;
                memset(po, 0, (size_t)k * d * sizeof(float));
                memset(pc, 0, (size_t)k * sizeof(int32_t));
                for (int64_t row = lo; row < hi; ++row) {           /* row order per (cluster, dim): one order */
                    const int j = asg[row * G + g];
                    for (int c = 0; c < d; ++c) po[j * d + c] += resid(x, row, D, base, base_idx, g * d + c);
                    pc[j] += 1;
                }
            }
        }
        for (int64_t e = 0; e < ne; ++e) {
            const int64_t gj = e / d;
            float s = 0.0f;
            int cnt = 0;
            for (int ch = 0; ch < nchunks; ++ch) {
                s += partial[(int64_t)ch * ne + e];
                cnt += pcnt[(int64_t)ch * G * k + gj];
            }
            if (cnt > 0) cent[e] = s / (float)cnt;
        }
    }
    free(asg); free(partial); free(pcnt);
    return 0;
}

static inline int lex_less(float d1, int64_t i1, float d2, int64_t i2)
{
    return d1 < d2 || (d1 == d2 && i1 < i2);
}

void oracle_ivfpq_probe(const float *q, int nq, int d, const float *cent, int nlist, int nprobe, int32_t *probe)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nq; ++i) {
        float *dist = malloc((size_t)nlist * sizeof(float));
        char *used = calloc((size_t)nlist, 1);
        for (int l = 0; l < nlist; ++l) {
            float acc = 0.0f;
            for (int c = 0; c < d; ++c) {
                const float diff = q[(int64_t)i * d + c] - cent[(int64_t)l * d + c];
                acc = fmaf(diff, diff, acc);
            }
            dist[l] = acc;
        }
        for (int s = 0; s < nprobe; ++s) {
            int bl = -1;
            for (int l = 0; l < nlist; ++l)
                if (!used[l] && (bl < 0 || lex_less(dist[l], l, dist[bl], bl))) bl = l;
            probe[(int64_t)i * nprobe + s] = bl;
            if (bl >= 0) used[bl] = 1;
        }
        free(dist); free(used);
    }
}

/* codes (n, M) and ids (n) in list order, list_start (nlist + 1) */
int oracle_ivfpq_search(const float *q, int nq, int d, const float *cent, const float *books, int M,
                        const uint8_t *codes, const int64_t *list_start, const int64_t *ids, const int32_t *probe,
                        int nprobe, int k, float *out_d, int64_t *out_i)
{
    const int dsub = d / M;
#pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < nq; ++i) {
        float *tab = malloc((size_t)M * 256 * sizeof(float));
        float *bd = out_d + (int64_t)i * k;
        int64_t *bi = out_i + (int64_t)i * k;
        for (int t = 0; t < k; ++t) {
            bd[t] = INFINITY;
            bi[t] = INT64_MAX;
        }
        for (int s = 0; s < nprobe; ++s) {
            const int list = probe[(int64_t)i * nprobe + s];
            if (list < 0) continue;
            for (int m = 0; m < M; ++m)
                for (int j = 0; j < 256; ++j) {
                    float acc = 0.0f;
                    for (int e = 0; e < dsub; ++e) {
                        const float r = q[(int64_t)i * d + m * dsub + e] - cent[(int64_t)list * d + m * dsub + e];
                        const float diff = r - books[((int64_t)m * 256 + j) * dsub + e];
                        acc = fmaf(diff, diff, acc);
                    }
                    tab[m * 256 + j] = acc;
                }
            for (int64_t p = list_start[list]; p < list_start[list + 1]; ++p) {
                float acc = 0.0f;
                for (int m = 0; m < M; ++m) acc += tab[m * 256 + codes[p * M + m]];
                const int64_t id = ids[p];
                if (!lex_less(acc, id, bd[k - 1], bi[k - 1])) continue;
                int t = k - 1;
                while (t > 0 && lex_less(acc, id, bd[t - 1], bi[t - 1])) {
                    bd[t] = bd[t - 1];
                    bi[t] = bi[t - 1];
                    --t;
                }
                bd[t] = acc;
                bi[t] = id;
            }
        }
        for (int t = 0; t < k; ++t)
            if (bi[t] == INT64_MAX) bi[t] = -1;
        free(tab);
    }
    return 0;
}
