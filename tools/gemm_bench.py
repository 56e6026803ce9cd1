"""Shape-by-shape timing of the hand-written 1x1-conv GEMM (gemm.hip) against torch.mm (hipBLASLt) on the layer shapes
of one training step at 256 pairs (512 clip-views): forward and data-gradient products of every stage.

    python tools/gemm_bench.py [--stats] [--pro]      # on the GPU box
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd import ops  # noqa: E402


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3          # us


def wgrad(args, dev):
    """dW = G X^T per layer shape (GRAFP_WGRAD_TILE=T/S/L/old selects a tile configuration / the register-staged kernel)."""
    depth, tot = (2, 2, 6, 2), 0.0
    for stage, (C, N) in enumerate(((64, 1024), (128, 512), (256, 256), (512, 128))):
        M = args.clips * N
        if str(stage) not in args.stages:
            continue
        for name, co, ci, g in (("fc1", C, C, 1), ("gconv g4", 2 * C, 2 * C, 4), ("gfc2", C, 2 * C, 1),
                                ("ffn1", 4 * C, C, 1), ("ffn2", C, 4 * C, 1)):
            G = torch.randn(co, M, device=dev).to(torch.bfloat16)
            X = torch.randn(ci, M, device=dev).to(torch.bfloat16)
            tab = torch.rand(ci, args.views, 2, device=dev)
            t0 = timeit(lambda: ops.conv1x1_wgrad(G, X, co, ci, g, M, args.views))
            t1 = 0.0 if args.no_pro else timeit(lambda: ops.conv1x1_wgrad(G, X, co, ci, g, M, args.views, tab, 1))
            by, fl = (co + ci) * M * 2.0, 2.0 * co * (ci // g) * M
            tot += t0 * depth[stage]
            print(f"s{stage} {name:9s} {co:5d} x {ci:5d} g={g} M={M:7d}  wgrad {t0:7.1f} us (+pro {t1:7.1f}) | "
                  f"{by / t0 / 1e6:5.2f} TB/s {fl / t0 / 1e6:7.1f} TF/s", flush=True)
    print(f"weighted by blocks per stage: {tot / 1e3:.2f} ms")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--clips", type=int, default=512)
    ap.add_argument("--wgrad", action="store_true", help="time the weight-gradient kernel instead")
    ap.add_argument("--stages", default="0123", help="--wgrad: encoder stages to run")
    ap.add_argument("--no-pro", action="store_true", help="--wgrad: skip the normalise-on-load variant")
    args = ap.parse_args()
    dev = "cuda:0"
    if args.wgrad:
        return wgrad(args, dev)
    rows = []
    for stage, (C, N) in enumerate(((64, 1024), (128, 512), (256, 256), (512, 128))):
        M = args.clips * N
        shapes = [("fc1  CxC", C, C, 1), ("gconv 2Cx2C g4", 2 * C, 2 * C, 4), ("gfc2 Cx2C", C, 2 * C, 1),
                  ("ffn1 4CxC", 4 * C, C, 1), ("ffn2 Cx4C", C, 4 * C, 1), ("d_gfc2 2CxC", 2 * C, C, 1)]
        for name, R, K, g in shapes:
            w = (0.1 * torch.randn(R, K // g, device=dev)).to(torch.bfloat16)
            x = torch.randn(K, M, device=dev).to(torch.bfloat16)
            tab = torch.rand(K, args.views, 2, device=dev)
            t_plain = timeit(lambda: ops.conv1x1_gemm(w, x, g, args.views))
            t_stats = timeit(lambda: ops.conv1x1_gemm(w, x, g, args.views, stats=True))
            t_pro = timeit(lambda: ops.conv1x1_gemm(w, x, g, args.views, pro_tab=tab, pro_act=1, stats=True))
            if g == 1:
                t_lib = timeit(lambda: torch.mm(w, x))
            else:
                w3 = w.reshape(g, R // g, K // g)
                t_lib = timeit(lambda: torch.bmm(w3, x.reshape(g, K // g, M)))
            by = (R + K) * M * 2.0
            fl = 2.0 * R * (K // g) * M
            rows.append((stage, name, R, K, g, M, t_lib, t_plain, t_stats, t_pro, by, fl))
            print(f"s{stage} {name:16s} R={R:5d} K={K:5d} g={g} M={M:7d}  lib {t_lib:7.1f} us | gemm {t_plain:7.1f} "
                  f"(+stats {t_stats:7.1f}, +pro {t_pro:7.1f}) us | {by / t_plain / 1e6:6.2f} TB/s {fl / t_plain / 1e6:7.1f} TF/s",
                  flush=True)
    # blocks per stage: 2, 2, 6, 2
    depth = (2, 2, 6, 2)
    tot_lib = sum(r[6] * depth[r[0]] for r in rows)
    tot = sum(r[7] * depth[r[0]] for r in rows)
    print(f"weighted by blocks per stage: library {tot_lib / 1e3:.2f} ms, gemm.hip {tot / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
