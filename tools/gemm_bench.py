"""Shape-by-shape timing of the hand-written 1x1-conv GEMM (gemm.hip) on the layer shapes of one training step: the
forward products (with the statistics epilogue), the data-gradient products and the concatenated-operand data
gradients of every stage, per tile configuration.

    make -C grafp_amd/csrc measure                                   # libgrafp_hip_measure.so: plan overrides compiled in
    GRAFP_HIP_LIB=$PWD/grafp_amd/libgrafp_hip_measure.so python tools/gemm_bench.py [--clips 2048] [--cfgs auto,S,L,N]
    ... --wgrad [--stages 0123]                                       # the weight-gradient kernel instead

Without the measurement library only the `auto` column (the shipping plan) can be timed.
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd import ops  # noqa: E402
from grafp_amd._lib import lib  # noqa: E402

CFG_IDS = {"S": 0, "L": 1, "N32": 2, "N64": 3, "N128": 4, "XL": 5}
CFG_NAMES = {v: k for k, v in CFG_IDS.items()}


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


def plan_of(R, K, g, M, views):
    info = (ctypes.c_int * 8)()
    lib.grafp_conv1x1_gemm_plan(R, K, g, M, views, info)
    return CFG_NAMES.get(info[0], "?"), info[1], info[2]


def force(cfg):
    if cfg is None:
        os.environ.pop("GRAFP_GEMM_CFG", None)
    else:
        os.environ["GRAFP_GEMM_CFG"] = str(cfg)


def wgrad(args, dev):
    """dW = G X^T per layer shape (GRAFP_WGRAD_TILE in a measurement build selects a tile configuration)."""
    depth, tot, tot_floor = (2, 2, 6, 2), 0.0, 0.0
    for stage, (C, N) in enumerate(((64, 1024), (128, 512), (256, 256), (512, 128))):
        M = args.clips * N
        if str(stage) not in args.stages:
            continue
        for name, co, ci, g in (("fc1", C, C, 1), ("gconv g4", 2 * C, 2 * C, 4), ("gfc2", C, 2 * C, 1),
                                ("ffn1", 4 * C, C, 1), ("ffn2", C, 4 * C, 1)):
            G = torch.randn(co, M, device=dev).to(torch.bfloat16)
            X = torch.randn(ci, M, device=dev).to(torch.bfloat16)
            t0 = timeit(lambda: ops.conv1x1_wgrad(G, X, co, ci, g, M, 1))
            by, fl = (co + ci) * M * 2.0, 2.0 * co * (ci // g) * M
            info = (ctypes.c_int * 8)()
            lib.grafp_conv1x1_wgrad_plan(co, ci, g, M, 1, info)
            # floors: the operands once at the part's measured read rate (vmem_pipe_bench: 6.25 TB/s), the products at
            # the bf16 matrix peak (2.5 PFLOP/s) and at the rate measured on non-constant operands (1.67, mfma_data_bench)
            f_rd, f_mx, f_md = by / 6.25e6, fl / 2.5e9, fl / 1.67e9
            floor = max(f_rd, f_md)
            tot += t0 * depth[stage]
            tot_floor += floor * depth[stage]
            print(f"s{stage} {name:9s} {co:5d} x {ci:5d} g={g} M={M:7d}  wgrad {t0:7.1f} us | "
                  f"{by / t0 / 1e6:5.2f} TB/s {fl / t0 / 1e6:7.1f} TF/s | cfg {info[0]:2d} tiles {info[1]}x{info[2]} slices {info[3]:4d} | "
                  f"read floor {f_rd:6.1f} us, matrix floor {f_mx:6.1f} (data-dependent {f_md:6.1f}) -> {floor / t0:4.2f} of the "
                  f"larger; lost {depth[stage] * (t0 - floor):7.1f} us per step", flush=True)
    print(f"weighted by blocks per stage: {tot / 1e3:.2f} ms measured, {tot_floor / 1e3:.2f} ms at the floors")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--clips", type=int, default=2048)
    ap.add_argument("--cfgs", default="auto,S,L,N", help="auto = the shipping plan; S, L, N (the 512-column tile that "
                    "fits the rows: N32 / N64 / N128) need the measurement library")
    ap.add_argument("--wgrad", action="store_true", help="time the weight-gradient kernel instead")
    ap.add_argument("--stages", default="0123", help="encoder stages to run")
    ap.add_argument("--lib", action="store_true", help="also time torch.mm / torch.bmm (hipBLASLt) on the shape")
    args = ap.parse_args()
    dev = "cuda:0"
    if args.wgrad:
        return wgrad(args, dev)
    cfgs = args.cfgs.split(",")
    depth = (2, 2, 6, 2)
    totals = {c: 0.0 for c in cfgs}
    floor_tot = 0.0
    for stage, (C, N) in enumerate(((64, 1024), (128, 512), (256, 256), (512, 128))):
        if str(stage) not in args.stages:
            continue
        M = args.clips * N
        # (name, R, K, groups, stats?, cat: rows of the second operand)
        shapes = [("fc1", C, C, 1, True, 0), ("gconv g4", 2 * C, 2 * C, 4, True, 0), ("gfc2", C, 2 * C, 1, True, 0),
                  ("ffn1", 4 * C, C, 1, True, 0), ("ffn2", C, 4 * C, 1, True, 0),
                  ("d_fc1 cat", C, C, 1, False, C), ("d_gconv g4", 2 * C, 2 * C, 4, False, 0),
                  ("d_gfc2", 2 * C, C, 1, False, 0), ("d_ffn1 cat", C, 4 * C, 1, False, C), ("d_ffn2", 4 * C, C, 1, False, 0)]
        for name, R, K, g, stats, cat in shapes:
            w = (0.1 * torch.randn(R, (K + cat) // g, device=dev)).to(torch.bfloat16)
            x = torch.randn(K, M, device=dev).to(torch.bfloat16)
            x2 = torch.randn(cat, M, device=dev).to(torch.bfloat16) if cat else None
            views = args.views if stats else 1
            by = (R + K + cat) * M * 2.0
            fl = 2.0 * R * ((K + cat) // g) * M
            floor = max(by / 8e12, fl / 2.5e15) * 1e6
            floor_tot += floor * depth[stage]
            rg = R // g
            ncfg = "N32" if rg <= 32 else "N64" if rg <= 64 else "N128"
            cols, t_auto = [], None
            for c in cfgs:
                if c == "N" and rg > 128:
                    cols.append(f"{'':>15s}")
                    totals[c] += (t_auto or 0.0) * depth[stage]       # not applicable: the shipping plan's time
                    continue
                force(None if c == "auto" else CFG_IDS[ncfg if c == "N" else c])
                if cat:
                    t = timeit(lambda: ops.conv1x1_gemm_cat(w, x, x2))
                else:
                    t = timeit(lambda: ops.conv1x1_gemm(w, x, g, views, stats=stats))
                totals[c] += t * depth[stage]
                if c == "auto":
                    t_auto = t
                tag = plan_of(R, K + cat, g, M, views)[0] if c == "auto" else (ncfg if c == "N" else c)
                cols.append(f"{tag:>4s} {t:7.1f} us")
            force(None)
            extra = ""
            if args.lib and not cat:
                if g == 1:
                    t_lib = timeit(lambda: torch.mm(w, x))
                else:
                    w3 = w.reshape(g, R // g, K // g)
                    t_lib = timeit(lambda: torch.bmm(w3, x.reshape(g, K // g, M)))
                extra = f" | lib {t_lib:7.1f}"
            best = min(float(c.split()[1]) for c in cols if c.strip())
            print(f"s{stage} {name:11s} R={R:5d} K={K + cat:5d} g={g} M={M:7d} {'st' if stats else '  '} | "
                  + " | ".join(cols) + f" | floor {floor:6.1f} | best {by / best / 1e6:5.2f} TB/s {fl / best / 1e6:7.1f} TF/s"
                  + extra, flush=True)
    print("weighted by blocks per stage (ms): " + ", ".join(f"{c} {v / 1e3:.2f}" for c, v in totals.items())
          + f", two-ceiling floor {floor_tot / 1e3:.2f}")


if __name__ == "__main__":
    main()
