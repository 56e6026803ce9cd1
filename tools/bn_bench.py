"""Timing of the single-pass BatchNorm backward (bn_bwd1_kernel) on the layer shapes of one training step.

    python tools/bn_bench.py [--clips 512] [--spin N]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd import ops  # noqa: E402
from grafp_amd._lib import lib  # noqa: E402


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=512)
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--spin", type=int, default=None)
    ap.add_argument("--pad", type=int, default=None, help="elements between the three tensors (one allocation)")
    ap.add_argument("--affine", action="store_true", help="time the normalise pass (bn_affine_bf16_kernel) instead")
    args = ap.parse_args()
    spin = -1 if args.spin is None else args.spin
    dev, tot = "cuda:0", 0.0
    depth = (2, 2, 6, 2)
    if args.affine:
        for stage, (C, N) in enumerate(((64, 1024), (128, 512), (256, 256), (512, 128))):
            M = args.clips * N
            for name, rows, n, with_res in (("C", C, 1, False), ("C+res", C, 2, True), ("2C", 2 * C, 1, False), ("4C", 4 * C, 1, False)):
                y = torch.randn(rows, M, device=dev).to(torch.bfloat16)
                res = torch.randn(rows, M, device=dev).to(torch.bfloat16) if with_res else None
                tab = torch.rand(rows, args.views, 2, device=dev)
                t = timeit(lambda: ops.bn_affine(y, tab, args.views, res, ops.ACT_RELU, 0.0))
                tot += t * n * depth[stage]
                by = (3.0 if with_res else 2.0) * rows * M * 2
                print(f"s{stage} rows {rows:5d} {name:6s} M {M:8d}: {t:8.1f} us  {by / t / 1e6:5.2f} TB/s", flush=True)
        print(f"per step: {tot / 1e3:.2f} ms")
        return
    for stage, (C, N) in enumerate(((64, 1024), (128, 512), (256, 256), (512, 128))):
        M = args.clips * N
        for name, rows, n in (("C", C, 3), ("2C", 2 * C, 1), ("4C", 4 * C, 1)):
            y = torch.randn(rows, M, device=dev).to(torch.bfloat16)
            dz = torch.randn(rows, M, device=dev).to(torch.bfloat16)
            g, b = torch.rand(rows, device=dev) + 0.5, torch.randn(rows, device=dev)
            mean, invstd = torch.zeros(rows, args.views, device=dev), torch.ones(rows, args.views, device=dev)
            if args.pad is None:
                t = timeit(lambda: ops._bn_bwd(y, dz, rows, M, args.views, None, g, b, mean, invstd, ops.ACT_RELU, 0.0, True))
            else:
                n = rows * M
                buf = torch.empty(3 * n + 2 * args.pad, dtype=torch.bfloat16, device=dev)
                yy, dd, dx = buf[:n].view(rows, M), buf[n + args.pad:2 * n + args.pad].view(rows, M), buf[2 * n + 2 * args.pad:].view(rows, M)
                yy.copy_(y), dd.copy_(dz)
                dg, db = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
                nb = lib.grafp_bn_workspace(rows, M)
                ws = torch.empty(nb, dtype=torch.uint8, device=dev)
                sync = ops._bn_sync(yy.device, rows, M)
                P = ops._p
                t = timeit(lambda: ops.check(lib.grafp_bn_bwd_1pass(P(yy), P(dd), ops._DT[yy.dtype], rows, M, args.views, None, P(g), P(b), P(mean), P(invstd), ops.ACT_RELU, 0.0, 1, P(dx), P(dg), P(db), None, P(ws), nb, P(sync), spin, ops._stream()), "bn_bwd"))
                del buf
            tot += t * n * depth[stage]
            print(f"s{stage} rows {rows:5d} M {M:8d}: {t:8.1f} us  {3.0 * rows * M * 2 / t / 1e6:5.2f} TB/s", flush=True)
    print(f"per step: {tot / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
