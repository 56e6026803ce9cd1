"""What would a MIXED precision mode buy?  (measurement, not a product path.)

The shipped bf16 mode stores every activation in bf16; end to end its embeddings drift from the f32 mode's by ~0.1 relative
L2 on a trained model, because a rounding that flips one k-NN edge changes that node's aggregate for good.  This tool
EMULATES, on the f32 kernels, modes in which the GEMM operands are bf16 (what the matrix cores would consume) while some
or all activations stay f32, and prints the free-running distance of each from the pure f32 mode:

  A  f32 activations everywhere, GEMM operands (weights and activations) rounded to bf16 at the product;
  B  as A, and the wide hidden tensors (grouped-conv output, FFN hidden layer) stored in bf16 -- only the residual stream
     and the tensors the k-NN graph is built from stay f32;
  C  as B with the k-NN features ALSO rounded to bf16 (the residual stream alone in f32);
  bf16  the shipped mode.

    python tools/mixed_mode_probe.py            # on the GPU box; trained synthetic-retrieval model, 256 segments
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from _retrieval_case import build_case  # noqa: E402


def rb(t):
    return t.to(torch.bfloat16).float()


def main():
    from grafp_amd import ops
    from grafp_amd.encoder import _dense
    from grafp_amd.encoder.gcn_lib import torch_edge
    dev = torch.device("cuda:0")
    case = build_case(dev, n_tracks=8, seconds=20, train_steps=40, n_test=50)
    model = case["model"]
    segs = case["db"][:256]

    def embed():
        with torch.no_grad():
            return model.embed(segs)[1].float()
    z32 = embed()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        z16 = embed()

    orig_rows, orig_cba = ops.conv1x1_rows, _dense.conv_bn_act
    orig_nb = torch_edge.DenseDilatedKnnGraph.neighbours
    mode = {"hidden": False, "knn": False}

    def rows(x, w, groups=1, w_lowp=None):                       # bf16 operands, f32 accumulation and result
        return orig_rows(rb(x), rb(w.detach()), groups, None)

    def cba(conv, bn, x, residual=None, **kw):
        out = orig_cba(conv, bn, x, residual=residual, **kw)
        wide = conv.out_channels > conv.in_channels or conv.groups > 1       # FFN hidden layer, grouped graph conv
        return rb(out) if (mode["hidden"] and wide and residual is None) else out

    def nb(self, x, *a, **k):
        return orig_nb(self, rb(x) if mode["knn"] else x, *a, **k)

    def run(hidden, knn):
        mode["hidden"], mode["knn"] = hidden, knn
        ops.conv1x1_rows, _dense.conv_bn_act = rows, cba
        _dense.ops.conv1x1_rows = rows
        torch_edge.DenseDilatedKnnGraph.neighbours = nb
        import grafp_amd.encoder.graph_encoder as ge
        import grafp_amd.encoder.gcn_lib.torch_nn as tn
        import grafp_amd.encoder.gcn_lib.torch_vertex as tv
        saved = [(m, m.conv_bn_act) for m in (ge, tn, tv)]
        for m, _ in saved:
            m.conv_bn_act = cba
        try:
            return embed()
        finally:
            ops.conv1x1_rows, _dense.conv_bn_act = orig_rows, orig_cba
            torch_edge.DenseDilatedKnnGraph.neighbours = orig_nb
            for m, f in saved:
                m.conv_bn_act = f

    def rel(z):
        r = torch.linalg.norm(z - z32, dim=1) / torch.linalg.norm(z32, dim=1)
        return f"max {float(r.max()):.4f}  mean {float(r.mean()):.4f}  median {float(r.median()):.5f}"
    print("free-running embedding distance from the f32 mode (relative L2 per segment, 256 segments, trained model):")
    print("  shipped bf16 mode                                   :", rel(z16))
    print("  A  bf16 GEMM operands, every activation f32          :", rel(run(False, False)))
    print("  B  A + wide hidden tensors stored in bf16            :", rel(run(True, False)))
    print("  C  B + k-NN features rounded to bf16                 :", rel(run(True, True)))


if __name__ == "__main__":
    main()
