"""IVF-PQ (faiss==1.7.2 IndexIVFPQ as configured at /root/reference/eval.py:65-69: 64 lists, M = 64 sub-quantisers,
8 bits, by_residual, nprobe = 20 at :122) restated in numpy from its published algorithm.  TEST INFRASTRUCTURE (see
oracle/__init__.py).  faiss is not vendored and not installable here: PARITY UNPINNED against faiss itself.  What pins
the restatement: ADC(q, code) == || q - reconstruct(code) ||^2 exactly (tests/test_oracle.py), i.e. the scan is the
squared distance to the de-quantised vector, which is the definition.

Quantisers (coarse centroids, codebooks) are INPUTS here: the k-means that produces them is part of the product
(grafp_amd/ivfpq.py, seeded) and of faiss (not reproducible without it); encode / search are deterministic given them.
"""
import numpy as np


def assign(x, centroids):
    """Nearest coarse centroid (squared L2, lowest id on ties)."""
    d = (centroids * centroids).sum(1)[None, :] - 2.0 * x.astype(np.float64) @ centroids.astype(np.float64).T
    return d.argmin(axis=1)


def encode(x, centroids, codebooks):
    """x (n, d) -> (list ids (n), codes (n, M) uint8): per sub-space the codeword nearest to the residual."""
    M, ksub, dsub = codebooks.shape
    a = assign(x, centroids)
    res = (x.astype(np.float64) - centroids[a].astype(np.float64)).reshape(len(x), M, dsub)
    codes = np.empty((len(x), M), dtype=np.uint8)
    for m in range(M):
        diff = res[:, m, None, :] - codebooks[m].astype(np.float64)[None]            # (n, 256, dsub)
        codes[:, m] = (diff * diff).sum(-1).argmin(axis=1)
    return a, codes


def reconstruct(a, codes, centroids, codebooks):
    M = codebooks.shape[0]
    rec = centroids[a].astype(np.float64).copy().reshape(len(a), M, -1)
    for m in range(M):
        rec[:, m, :] += codebooks[m][codes[:, m]]
    return rec.reshape(len(a), -1)


def search(q, a, codes, centroids, codebooks, nprobe, k):
    """Asymmetric-distance search: (D (nq, k) float64, I (nq, k) int64 insertion ids; -1 / inf when fewer found)."""
    M, ksub, dsub = codebooks.shape
    coarse = (centroids * centroids).sum(1)[None, :] - 2.0 * q.astype(np.float64) @ centroids.astype(np.float64).T
    probe = np.argsort(coarse, axis=1, kind="stable")[:, :nprobe]
    D = np.full((len(q), k), np.inf)
    I = np.full((len(q), k), -1, dtype=np.int64)
    for i in range(len(q)):
        cand_d, cand_i = [], []
        for lst in probe[i]:
            members = np.nonzero(a == lst)[0]
            if members.size == 0:
                continue
            r = (q[i].astype(np.float64) - centroids[lst].astype(np.float64)).reshape(M, dsub)
            tab = ((r[:, None, :] - codebooks.astype(np.float64)) ** 2).sum(-1)      # (M, 256)
            cand_d.append(tab[np.arange(M)[None, :], codes[members]].sum(axis=1))
            cand_i.append(members)
        if cand_d:
            cd, ci = np.concatenate(cand_d), np.concatenate(cand_i)
            order = np.lexsort((ci, cd))[:k]
            D[i, :len(order)], I[i, :len(order)] = cd[order], ci[order]
    return D, I
