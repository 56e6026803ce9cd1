"""tools/search_bench.py [--n ROWS] [--pre|--f32] [NQ ...]: latency and throughput of ops.search_l2 per batch size on a
random unit-norm database (planted neighbours at sigma 0.05: top-1 must be 1.000).  --pre (default) = the bf16
pre-filter path (grafp_knn_search_l2_pre), --f32 = the all-f32 path.  With GRAFP_HIP_LIB pointing at the measurement
build the plan knobs of csrc/tuning.h (GRAFP_SEARCH_NQS) can be swept from the environment."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--f32", action="store_true")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("nq", nargs="*", type=int)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(2)
    db = torch.nn.functional.normalize(torch.randn(a.n, 128, generator=gen, device=dev), dim=1)
    rows = torch.randint(0, a.n, (4096,), generator=gen, device=dev)
    q = torch.nn.functional.normalize(db[rows] + 0.05 * torch.randn(4096, 128, generator=gen, device=dev), dim=1)
    sq = ops.row_sqnorm(db)
    dbh = None if a.f32 else ops.rows_to_bf16(db)
    print(f"# {a.n} x 128, k = {a.k}, {'all-f32 path' if a.f32 else 'bf16 pre-filter + exact rescoring'}")
    for nq in a.nq or [1, 8, 41, 128, 512, 1024, 4096]:
        ops.search_l2(db, sq, q[:nq], a.k, db_bf16=dbh)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            D, I = ops.search_l2(db, sq, q[:nq], a.k, db_bf16=dbh)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.reps
        top1 = (I[:, 0] == rows[:nq]).float().mean().item()
        print(f"nq={nq:5d}  {dt * 1e3:8.3f} ms  {nq / dt:10.0f} QPS  top1 {top1:.3f}")


if __name__ == "__main__":
    main()
