#!/usr/bin/env python3
"""Golden vectors for the sequence rerank + hit-rate logic of the reference's eval_faiss (eval.py:170-332).

Run in the build container only (needs /root/reference):  python tests/golden/make_eval_golden.py
Writes tests/golden/eval_faiss.npz = OUTPUTS of the reference's own eval_faiss (hit_rates, raw_score flags, test_ids)
on a small synthetic {query, db, dummy_db} memmap set whose contents are regenerated from tests/_hashfill.py
(`eval_case()` in tests/_common.py builds the same arrays).

What this pins and what it does not: eval.py imports `faiss` (faiss-gpu==1.7.2, requirements.txt:7), which is not
installed here and not installable.  The stand-in below supplies ONLY `IndexFlatL2(d)` with `train/add/search/ntotal`
as an exact float64 squared-L2 search (ties -> lowest id) -- the documented meaning of IndexFlatL2 -- so that the
reference's code around it runs unmodified: index build order (:212-213), ground-truth ids (:250), offset
compensation (:273-274), unique candidates (:277), sequence scores (:280-287), top-10 and hit flags (:290-301),
rates (:305-310), side-effect files (:324-329).  faiss's own arithmetic stays "parity unpinned" (oracle/__init__.py).
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

from _common import eval_case, write_eval_case  # noqa: E402


class _IndexFlatL2:
    def __init__(self, d):
        self.d, self.x, self.ntotal, self.nprobe = d, np.zeros((0, d), np.float32), 0, 1

    def train(self, x):
        pass

    def add(self, x):
        self.x = np.concatenate([self.x, np.asarray(x, dtype=np.float32)], axis=0)
        self.ntotal = len(self.x)

    def search(self, q, k):
        q64, x64 = np.asarray(q, dtype=np.float64), self.x.astype(np.float64)
        d = ((q64[:, None, :] - x64[None, :, :]) ** 2).sum(-1)
        order = np.lexsort((np.broadcast_to(np.arange(len(x64)), d.shape), d), axis=1)[:, :k]
        return np.take_along_axis(d, order, axis=1).astype(np.float32), order.astype(np.int64)


def main():
    faiss = types.ModuleType("faiss")
    faiss.IndexFlatL2 = _IndexFlatL2
    sys.modules["faiss"] = faiss
    sys.path.insert(0, "/root/reference")
    import eval as ref_eval                                            # the reference's eval.py, unmodified

    case = eval_case()
    with tempfile.TemporaryDirectory() as tmp:
        write_eval_case(tmp, case)
        ids_path = os.path.join(tmp, "golden_test_ids.npy")
        np.save(ids_path, case["test_ids"])
        rates = ref_eval.eval_faiss(tmp, index_type="l2", nogpu=True, test_ids=ids_path,
                                    test_seq_len=case["test_seq_len"], k_probe=case["k_probe"])
        sub = [d for d in os.listdir(tmp) if os.path.isdir(os.path.join(tmp, d))]
        assert len(sub) == 1
        raw = np.load(os.path.join(tmp, sub[0], "raw_score.npy"))
        saved_rates = np.load(os.path.join(tmp, sub[0], "hit_rates.npy"))
        saved_ids = np.load(os.path.join(tmp, "test_ids.npy"))
    assert np.array_equal(rates, saved_rates) and np.array_equal(saved_ids, case["test_ids"])
    np.savez_compressed(os.path.join(HERE, "eval_faiss.npz"), hit_rates=rates, raw_score=raw, test_ids=saved_ids)
    print("hit rates (rows: top1 exact, top1 near, top3, top10; cols: lengths", case["test_seq_len"], ")")
    print(rates)


if __name__ == "__main__":
    main()
