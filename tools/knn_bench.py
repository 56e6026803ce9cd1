"""k-NN graph build per encoder stage: the exact-f32 MFMA kernel (knn_graph.hip) against the split-bf16 certified path
(knn_split.hip), random unit features and -- with --model -- the encoder's own features.

    python tools/knn_bench.py [--clips 2048] [--model]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd import ops  # noqa: E402


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=2048)
    ap.add_argument("--model", action="store_true", help="also on the features a random-init encoder feeds its graphs")
    ap.add_argument("--dtype", default="bf16", choices=("bf16", "f32"),
                    help="bf16: (C, B, N) bf16 rows as the training path hands them over (RAW form of the certified path); "
                         "f32: (B, C, N) f32 (split hi/lo planes)")
    args = ap.parse_args()
    dev = "cuda:0"
    depth, tot = (2, 2, 6, 2), [0.0, 0.0]
    feats = {}
    if args.model:
        from grafp_amd.train import build_model, synthetic_batch
        from grafp_amd.util import load_config
        from grafp_amd.modules.transformations import GPUTransformNeuralfp
        cfg = load_config()
        nb = min(args.clips // 2, 128)
        cfg["bsz_train"] = nb
        torch.manual_seed(0)
        model = build_model(cfg, device=dev).train()
        aug = GPUTransformNeuralfp(cfg, None, None, train=True)
        x_i, x_j = synthetic_batch(nb, 5, dev)
        orig = ops.knn_graph

        def rec(x, k, normalize=True, layout="bcn", index_dtype=torch.int64, prefilter=None):
            feats.setdefault(x.shape[0] if layout == "cbn" else x.shape[1], (x.detach().clone(), layout))
            return orig(x, k, normalize, layout, index_dtype, prefilter)
        ops.knn_graph = rec
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            model(*aug(x_i, x_j))
        ops.knn_graph = orig
    for stage, (C, N) in enumerate(((64, 1024), (128, 512), (256, 256), (512, 128))):
        if args.dtype == "bf16":
            x, lay = torch.randn(C, args.clips, N, device=dev).to(torch.bfloat16), "cbn"
        else:
            x, lay = torch.randn(args.clips, C, N, device=dev), "bcn"
        t_f32 = timeit(lambda: ops.knn_graph(x, 3, layout=lay, index_dtype=torch.int32, prefilter=False))
        t_split = timeit(lambda: ops.knn_graph_split(x, 3, layout=lay, index_dtype=torch.int32))
        a = ops.knn_graph(x, 3, layout=lay, index_dtype=torch.int32, prefilter=False)
        b, unc = ops.knn_graph_split(x, 3, layout=lay, index_dtype=torch.int32, return_uncertified=True)
        fl = 2.0 * N * N * C * args.clips
        tot[0] += t_f32 * depth[stage]
        tot[1] += t_split * depth[stage]
        line = (f"s{stage} C={C:4d} N={N:5d} clips={args.clips}: exact-f32 {t_f32:8.1f} us ({fl / t_f32 / 1e6:6.1f} TF/s) | split "
                f"{t_split:8.1f} us ({fl / t_split / 1e6:6.1f} TF/s equiv) | equal {bool(torch.equal(a, b))} | uncertified "
                f"{int(unc)} of {args.clips * N} ({100.0 * int(unc) / (args.clips * N):.2f} %)")
        if C in feats:
            xf, layout = feats[C]
            tm_f32 = timeit(lambda: ops.knn_graph(xf, 3, layout=layout, index_dtype=torch.int32, prefilter=False))
            tm_split = timeit(lambda: ops.knn_graph_split(xf, 3, layout=layout, index_dtype=torch.int32))
            a = ops.knn_graph(xf, 3, layout=layout, index_dtype=torch.int32, prefilter=False)
            b, unc = ops.knn_graph_split(xf, 3, layout=layout, index_dtype=torch.int32, return_uncertified=True)
            nq = a.shape[0] * a.shape[1]
            line += (f" || encoder features ({a.shape[0]} clips): {tm_f32:7.1f} vs {tm_split:7.1f} us, equal "
                     f"{bool(torch.equal(a, b))}, uncertified {100.0 * int(unc) / nq:.2f} %")
        print(line, flush=True)
    print(f"per step (blocks per stage 2, 2, 6, 2): exact-f32 {tot[0] / 1e3:.2f} ms, split {tot[1] / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
