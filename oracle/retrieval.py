"""Retrieval evaluation restated in numpy.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows /root/reference/eval.py:170-332 (`eval_faiss` with index_type='l2'): index = dummy_db rows
followed by db rows (:212-213); ground truth id = test_id + n_dummy (:250); for every test id and
query length sl: top-k_probe search of the sl query segments (:269), offset compensation (:273-274),
unique non-negative candidates (:277), score = mean of the diagonal of q . seq^T over the candidate
sequence fake_recon_index[cid:cid+sl] (:280-287), predictions = 10 best scores (:290), hit flags
(:294-301), rates in percent (:305-310).
"""
import numpy as np

from . import native


def sequence_scores(q, index_rows, candidates, sl):
    """eval.py:280-287.  A candidate whose sequence would run past the end of the index makes the
    reference's np.dot/np.diag silently use the shorter block (diag of a non-square product); the
    mean is then over min(sl, rows) terms -- reproduced here."""
    scores = np.zeros(len(candidates))
    for ci, cid in enumerate(candidates):
        seq = index_rows[cid:cid + sl, :]
        scores[ci] = np.mean(np.diag(np.dot(q, seq.T)))
    return scores


def eval_l2(query, db, dummy_db, test_ids, test_seq_len, k_probe=20, search=None):
    """Returns (hit_rates (4, n_len) in %, raw flags (n_test, 4*n_len), top1 predictions (n_test, n_len))."""
    search = search or (lambda index, q, k: native.flat_search_l2(index, q, k))
    index_rows = np.concatenate([dummy_db, db], axis=0).astype(np.float32)
    n_dummy = dummy_db.shape[0]
    test_ids = np.asarray(test_ids)
    gt_ids = test_ids + n_dummy
    n_test, n_len = len(test_ids), len(test_seq_len)
    flags = np.zeros((4, n_test, n_len), dtype=int)
    top1 = np.full((n_test, n_len), -1, dtype=np.int64)
    for ti, test_id in enumerate(test_ids):
        for si, sl in enumerate(test_seq_len):
            q = query[test_id:test_id + sl, :]
            _, I = search(index_rows, q, k_probe)
            I = I.copy()
            for off in range(len(I)):
                I[off, :] -= off
            cand = np.unique(I[np.where(I >= 0)])
            scores = sequence_scores(q, index_rows, cand, sl)
            pred = cand[np.argsort(-scores, kind="stable")[:10]]
            gt = gt_ids[ti]
            top1[ti, si] = pred[0]
            flags[0, ti, si] = int(gt == pred[0])
            flags[1, ti, si] = int(pred[0] in (gt - 1, gt, gt + 1))
            flags[2, ti, si] = int(gt in pred[:3])
            flags[3, ti, si] = int(gt in pred[:10])
    rates = 100.0 * flags.mean(axis=1)
    return rates, np.concatenate(list(flags), axis=1), top1


def exact_search_f64(db, q, k):
    """Independent float64 exact search: sum((q-d)^2), ties -> lowest id (checks flat_search.c)."""
    db64, q64 = db.astype(np.float64), q.astype(np.float64)
    d = ((q64[:, None, :] - db64[None, :, :]) ** 2).sum(-1)
    order = np.lexsort((np.broadcast_to(np.arange(db.shape[0]), d.shape), d), axis=1)[:, :k]
    return np.take_along_axis(d, order, axis=1), order
