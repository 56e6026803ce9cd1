"""The four-wave 256 x 256 tile (gemm_xl.h) against the eight-wave one on the same operands (measurement library: the
tile is forced through GRAFP_GEMM_CFG): y must be BIT-EQUAL (same k order inside every output), the statistics partials
must finalise to the same mean / invstd, the concatenated-operand and affine-epilogue forms likewise.  Repeated launches
must be bit-identical (a race between the ring and a fragment read shows up as run-to-run differences).

    make -C grafp_amd/csrc measure
    GRAFP_HIP_LIB=$PWD/grafp_amd/libgrafp_hip_measure.so python tools/gemm_xl_check.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd import ops  # noqa: E402


def force(cfg):
    if cfg is None:
        os.environ.pop("GRAFP_GEMM_CFG", None)
    else:
        os.environ["GRAFP_GEMM_CFG"] = str(cfg)


def main():
    dev = "cuda:0"
    torch.manual_seed(0)
    bad = 0
    junk = torch.full((1 << 28,), float("nan"), device=dev)
    del junk
    # (R, K, groups, M, views, cat rows)
    shapes = [(256, 64, 1, 256 * 8, 1, 0), (256, 256, 1, 256 * 64, 2, 0), (1024, 256, 1, 256 * 256, 2, 0),
              (256, 1024, 1, 256 * 128, 2, 0), (512, 512, 1, 256 * 96, 2, 0), (1024, 1024, 4, 256 * 40, 2, 0),
              (256, 256, 1, 256 * 72, 1, 256), (512, 2048, 1, 256 * 36, 1, 512), (2048, 512, 1, 256 * 1024, 2, 0),
              (768, 128, 1, 256 * 24, 1, 0), (256, 128, 1, 256, 1, 0), (256, 160, 1, 256 * 3, 2, 0), (512, 192, 2, 256 * 5, 1, 0)]
    for R, K, g, M, views, cat in shapes:
        w = (0.1 * torch.randn(R, (K + cat) // g, device=dev)).to(torch.bfloat16)
        x = torch.randn(K, M, device=dev).to(torch.bfloat16)
        x[:, ::7] *= 3.0
        x2 = torch.randn(cat, M, device=dev).to(torch.bfloat16) if cat else None
        outs = {}
        tabe = torch.stack((torch.rand(R, views, device=dev) + 0.5, torch.randn(R, views, device=dev)), dim=2).contiguous()
        for name, cfg in (("L", 1), ("XL", 5)):
            force(cfg)
            if cat:
                y = ops.conv1x1_gemm_cat(w, x, x2)
                st = None
            else:
                y, part = ops.conv1x1_gemm(w, x, g, views, stats=True)
                gamma = torch.ones(R, device=dev); beta = torch.zeros(R, device=dev)
                mean, invstd, tab = ops.bn_finalize(part, R, K, g, M, views, gamma, beta, None, None, None, True, 0.1, 1e-5)
                st = (mean, invstd)
                y_plain = ops.conv1x1_gemm(w, x, g, views)
                assert torch.equal(y_plain, y), (name, "stats form changed y")
                z = ops.conv1x1_gemm_affine(w, x, tabe, g, views, act=1)
                outs[name + "_z"] = z
            again = ops.conv1x1_gemm_cat(w, x, x2) if cat else ops.conv1x1_gemm(w, x, g, views)
            if not torch.equal(again, y):
                print("  NOT REPRODUCIBLE", name, (again != y).sum().item())
                bad += 1
            outs[name] = (y, st)
        force(None)
        (yl, sl), (yx, sx) = outs["L"], outs["XL"]
        ref = (w.float().reshape(g, R // g, -1) @ (torch.cat((x, x2)) if cat else x).float().reshape(g, (K + cat) // g, M)).reshape(R, M)
        err_l = ((yl.float() - ref).norm() / ref.norm()).item()
        err_x = ((yx.float() - ref).norm() / ref.norm()).item()
        ok = torch.equal(yl, yx)
        msg = f"R={R} K={K}+{cat} g={g} M={M} views={views}: y equal {ok} (rel err vs f32 {err_l:.2e} / {err_x:.2e})"
        if not ok:
            bad += 1
            d = (yl != yx)
            msg += f" differing {d.sum().item()} of {d.numel()}; rows {d.any(1).nonzero().flatten()[:8].tolist()} cols {d.any(0).nonzero().flatten()[:8].tolist()}"
        if sl is not None:
            dm = (sl[0] - sx[0]).abs().max().item()
            di = ((sl[1] - sx[1]).abs() / sl[1].abs()).max().item()
            zok = torch.equal(outs["L_z"], outs["XL_z"])
            msg += f" | mean diff {dm:.2e} invstd rel {di:.2e} | affine epilogue equal {zok}"
            if dm > 1e-5 or di > 1e-5 or not zok:
                bad += 1
        print(msg, flush=True)
    print("FAILED" if bad else "all equal")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
