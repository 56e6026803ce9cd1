"""tools/ivfpq_bench.py [--n ROWS] [--nq QUERIES]: the protocol's IVF-PQ index (64 lists, 64 x 8-bit codes, nprobe 20;
/root/reference/eval.py:65-69,122) at database scale, every step on csrc/ivfpq.hip: training time (coarse + 64 sub-space
k-means on 65 536 rows, 25 iterations each), add rate, search rate, and recall of the exact nearest neighbour -- beside the
exact index (ops.FlatL2Index) on the same data."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd import ops  # noqa: E402
from grafp_amd.ivfpq import IVFPQIndex  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--nq", type=int, default=4096)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(2)
    # clustered unit vectors (fingerprints of neighbouring segments are close): 2000 centres + noise
    centres = torch.nn.functional.normalize(torch.randn(2000, 128, generator=gen, device=dev), dim=1)
    db = torch.nn.functional.normalize(centres[torch.randint(0, 2000, (a.n,), generator=gen, device=dev)]
                                       + 0.6 * torch.randn(a.n, 128, generator=gen, device=dev) / 128 ** 0.5 * 3, dim=1)
    rows = torch.randint(0, a.n, (a.nq,), generator=gen, device=dev)
    q = torch.nn.functional.normalize(db[rows] + 0.05 * torch.randn(a.nq, 128, generator=gen, device=dev), dim=1)

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t0

    idx = IVFPQIndex(128, nlist=64, M=64)
    _, t_train = timed(lambda: idx.train(db))
    _, t_add = timed(lambda: idx.add(db))
    idx.nprobe = 20
    idx.search(q[:64], 20)
    (D, I), t_s = timed(lambda: idx.search(q, 20))
    sq, dbh = ops.row_sqnorm(db), ops.rows_to_bf16(db)
    ops.search_l2(db, sq, q[:64], 20, db_bf16=dbh)
    (De, Ie), t_e = timed(lambda: ops.search_l2(db, sq, q, 20, db_bf16=dbh))
    print(f"# {a.n} x 128 clustered unit vectors, {a.nq} planted queries, k = 20")
    print(f"train (65 536 rows, 65 k-means x 25 iterations)  {t_train * 1e3:9.1f} ms")
    print(f"add   ({a.n} rows: list id + 64 code bytes)       {t_add * 1e3:9.1f} ms  {a.n / t_add / 1e6:.2f} M rows/s")
    print(f"search IVF-PQ nprobe 20/64                        {t_s * 1e3:9.1f} ms  {a.nq / t_s:10.0f} QPS  "
          f"top-1 = planted row {float((I[:, 0] == rows).float().mean()):.3f}  "
          f"exact NN in top-20 {float((I == Ie[:, :1]).any(1).float().mean()):.3f}")
    print(f"search exact (FlatL2Index, bf16 pre-filter)       {t_e * 1e3:9.1f} ms  {a.nq / t_e:10.0f} QPS  "
          f"top-1 = planted row {float((Ie[:, 0] == rows).float().mean()):.3f}")


if __name__ == "__main__":
    main()
