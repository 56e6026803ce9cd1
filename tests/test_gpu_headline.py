"""The N = 1 bench shape under the oracle: ONE bf16 training step at 1024 pairs = 2048 clip-views on one GPU (the
workload of bench.py's headline line, BASELINE.json metric "clips/sec contrastive step @ batch 1024").  `pytest -m gpu`.

Several launch plans fire ONLY at this size -- the tile rule for rows of >= 2^20 columns per view, four rounds of
workgroups for launches that stream >= 750 MB (gemm.hip: gemm_plan), the register-staged weight-gradient
tiles and their split-K slice counts beyond 750 MB of operands (wgrad.hip: wgrad_dma_plan), 8-vector chunks in the
single-pass BatchNorm backward -- so every hand-written kernel of the dense chain is checked INSIDE that step, on the
operands the step itself produced, the first time each distinct launch shape occurs:

  * conv1x1_gemm / conv1x1_gemm_cat (forward products, data gradients, [W^T | I][dY; dZ]): sampled 128-column slabs
    (first, last, both sides of the view boundary, random) against an f32 product of the same bf16 operands;
  * the statistics epilogue + bn_finalize_affine: per-view mean / invstd against float64 over the whole rows, z slabs;
  * bn_bwd1 (_bn_bwd): d(gamma), d(beta) against float64 over the whole rows, dY on slabs;
  * conv1x1_wgrad: the whole dW against a float64 product;
  * the 12 k-NN graphs of sampled clips bit-exact against oracle/csrc/knn_graph.c (torch_edge.py:7-18,70-103).
Reference: /root/reference/train.py:66-74 (one optimisation step).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PAIRS = 1024


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _slabs(M, views, seed):
    """128-column slabs: both ends, both sides of every view boundary, four random ones."""
    Mg = M // views
    starts = {0, M - 128}
    for v in range(1, views):
        starts |= {v * Mg - 128, v * Mg}
    rng = np.random.default_rng(seed)
    starts |= {int(s) * 128 for s in rng.integers(0, M // 128, size=4)}
    cols = torch.cat([torch.arange(s, s + 128) for s in sorted(starts)])
    return cols


def _check_product(y, wf, xf, groups, operand_rebuilt=False):
    """y (R, n) bf16 vs the f32 product of the same operands: one bf16 rounding step of slack plus the f32
    accumulation-order error, bounded by 2^-18 of sum |w||x| (K <= 4096 terms of 2^-24 each, generously).
    operand_rebuilt: xf is this test's own rebuild of an operand the kernel derives in LDS (normalise-on-load): the kernel's
    fma against torch's mul + add can move single operand elements across a bf16 rounding boundary, so (as in
    tests/test_gpu_gemm.py::test_gemm_normalise_on_load) the bars are a relative L2 error and an outlier fraction."""
    R, Kg = wf.shape
    Rg = R // groups
    ref = torch.cat([wf[g * Rg:(g + 1) * Rg] @ xf[g * Kg:(g + 1) * Kg] for g in range(groups)], dim=0)
    mag = torch.cat([wf[g * Rg:(g + 1) * Rg].abs() @ xf[g * Kg:(g + 1) * Kg].abs() for g in range(groups)], dim=0)
    refq = ref.to(torch.bfloat16).float()
    err = (y.float() - refq).abs()
    tol = refq.abs() * 2.0 ** -7 + mag * 2.0 ** -18 + 1e-30
    if operand_rebuilt:
        assert float((y.float() - refq).norm() / refq.norm()) < 2e-3, tuple(y.shape)
        assert float((err > tol).float().mean()) < 1e-3, (float((err > tol).float().mean()), tuple(y.shape))
        return
    assert bool((err <= tol).all()), (float((err / tol).max()), tuple(y.shape))
    assert float((y.float() != refq).float().mean()) < 2e-2


class _Checks:
    def __init__(self):
        self.seen = {"gemm": set(), "cat": set(), "affine": set(), "bn_bwd": set(), "wgrad": set()}
        self.graphs = []


def _pro(xs, tab, act, slope, view_of):
    """act(x * scale + shift) per operand row and view, rounded to bf16: what conv1x1_gemm / the weight gradient build from
    the raw operand when the producer deferred its BatchNorm (ops.DeferredNorm)."""
    t = tab.reshape(xs.shape[0], -1, 2).float()
    v = view_of.long()
    z = torch.addcmul(t[:, v, 1], xs, t[:, v, 0])
    z = torch.relu(z) if act == 1 else (torch.where(z > 0, z, z * slope) if act == 2 else z)
    return z.to(torch.bfloat16).float()


def test_headline_step_kernels_vs_references(dev):
    from grafp_amd import ops
    from grafp_amd.train import Trainer, build_model, synthetic_batch
    from grafp_amd.util import load_config
    from oracle import native
    cfg = load_config()
    cfg["bsz_train"] = PAIRS
    torch.manual_seed(3)
    model = build_model(cfg, device=dev)
    tr = Trainer(cfg, model, dev, amp_dtype=torch.bfloat16)
    x_i, x_j = synthetic_batch(PAIRS, seed=9, device=dev)
    tr.step(x_i, x_j)                      # one plain step first: BatchNorm statistics and Adam moments are warm
    ck = _Checks()
    orig = {n: getattr(ops, n) for n in ("conv1x1_gemm", "conv1x1_gemm_cat", "bn_finalize_affine", "_bn_bwd",
                                          "_wgrad_bf16", "knn_graph")}
    clips = [0, 17, PAIRS - 1, PAIRS, PAIRS + 476, 2 * PAIRS - 1]

    def gemm(w, x, groups=1, views=1, pro_tab=None, pro_act=0, pro_slope=0.0, stats=False):
        out = orig["conv1x1_gemm"](w, x, groups, views, pro_tab, pro_act, pro_slope, stats)
        key = (w.shape[0], x.shape[0], groups, x.shape[1], views, stats, pro_tab is not None)
        if key not in ck.seen["gemm"]:
            ck.seen["gemm"].add(key)
            y = out[0] if stats else out
            cols = _slabs(x.shape[1], views, len(ck.seen["gemm"])).to(x.device)
            xs = x[:, cols].float()
            if pro_tab is not None:        # normalise-on-load (stages 0-1): the operand the kernel builds in LDS
                xs = _pro(xs, pro_tab, pro_act, pro_slope, cols // (x.shape[1] // views))
            _check_product(y[:, cols], w.float(), xs, groups, operand_rebuilt=pro_tab is not None)
        return out

    def gemm_cat(w, x1, x2):
        y = orig["conv1x1_gemm_cat"](w, x1, x2)
        key = (w.shape[0], x1.shape[0], x2.shape[0], x1.shape[1])
        if key not in ck.seen["cat"]:
            ck.seen["cat"].add(key)
            cols = _slabs(x1.shape[1], 2, len(ck.seen["cat"])).to(y.device)
            _check_product(y[:, cols], w.float(), torch.cat((x1[:, cols], x2[:, cols]), dim=0).float(), 1)
        return y

    def affine(y, part, K, groups, views, gamma, beta, pre_bias, running_mean, running_var, momentum, eps,
               residual=None, act=0, slope=0.0):
        out = orig["bn_finalize_affine"](y, part, K, groups, views, gamma, beta, pre_bias, running_mean, running_var,
                                         momentum, eps, residual, act, slope)
        z, mean, invstd, tab = out
        C, M = y.shape
        key = (C, K, groups, M, views, residual is not None, act)
        if key not in ck.seen["affine"]:
            ck.seen["affine"].add(key)
            yv = y.reshape(C, views, M // views)
            m64 = torch.stack([yv[:, v].double().mean(dim=1) for v in range(views)], dim=1)
            v64 = torch.stack([yv[:, v].double().var(dim=1, unbiased=False) for v in range(views)], dim=1)
            pb = 0.0 if pre_bias is None else pre_bias.double()[:, None]
            np.testing.assert_allclose(mean.double().cpu().numpy(), (m64 + pb).cpu().numpy(), rtol=1e-4, atol=1e-4)       # f32 partial sums over 2^20 columns
            np.testing.assert_allclose(invstd.double().cpu().numpy(), (1.0 / torch.sqrt(v64 + eps)).cpu().numpy(), rtol=2e-4)   # f32 Chan combination over 2^20 columns
            cols = _slabs(M, views, 7 * len(ck.seen["affine"])).to(y.device)
            t = tab.reshape(C, views, 2)
            view_of = (cols // (M // views)).long()
            sc, sh = t[:, view_of, 0], t[:, view_of, 1]                        # (C, n)
            zr = torch.addcmul(sh, y[:, cols].float(), sc)
            zr = torch.relu(zr) if act == 1 else (torch.where(zr > 0, zr, zr * slope) if act == 2 else zr)
            if residual is not None:
                zr = zr + residual.reshape(C, M)[:, cols].float()
            err = (z[:, cols].float() - zr.to(torch.bfloat16).float()).abs()
            assert bool((err <= zr.abs() * 2.0 ** -7 + 1e-6).all()), float(err.max())
        return out

    def bn_bwd(y, dz, C, M, views, pb, g32, b32, mean, invstd, act, slope, training):
        out = orig["_bn_bwd"](y, dz, C, M, views, pb, g32, b32, mean, invstd, act, slope, training)
        dy, dgamma, dbeta, dpb = out
        key = (C, M, views, act, pb is not None)
        if key not in ck.seen["bn_bwd"]:
            ck.seen["bn_bwd"].add(key)
            Mg = M // views
            dg64 = torch.zeros(C, dtype=torch.float64, device=y.device)
            db64 = torch.zeros(C, dtype=torch.float64, device=y.device)
            cols = _slabs(M, views, 11 * len(ck.seen["bn_bwd"])).to(y.device)
            want_dy = torch.empty((C, cols.numel()), dtype=torch.float64, device=y.device)
            for v in range(views):
                yv = y[:, v * Mg:(v + 1) * Mg].double()
                if pb is not None:
                    yv = yv + pb.double()[:, None]
                xhat = (yv - mean[:, v].double()[:, None]) * invstd[:, v].double()[:, None]
                pre = xhat * g32.double()[:, None] + b32.double()[:, None]
                g = dz[:, v * Mg:(v + 1) * Mg].double()
                if act == 1:
                    g = g * (pre > 0)
                elif act == 2:
                    g = torch.where(pre > 0, g, g * slope)
                s1, s2 = g.sum(dim=1), (g * xhat).sum(dim=1)
                db64 += s1
                dg64 += s2
                sel = (cols >= v * Mg) & (cols < (v + 1) * Mg)
                cv = cols[sel] - v * Mg
                want_dy[:, sel] = (g32.double() * invstd[:, v].double())[:, None] * (
                    g[:, cv] - (s1 / Mg)[:, None] - xhat[:, cv] * (s2 / Mg)[:, None])
                del yv, xhat, pre, g
            scale_g = float(dg64.abs().max()) + 1e-30
            scale_b = float(db64.abs().max()) + 1e-30
            assert float((dgamma.double() - dg64).abs().max()) <= 1e-3 * scale_g, (key, "dgamma")
            assert float((dbeta.double() - db64).abs().max()) <= 1e-3 * scale_b, (key, "dbeta")
            err = (dy[:, cols].double() - want_dy).abs()
            tol = want_dy.abs() * 2.0 ** -7 + float(want_dy.abs().max()) * 2.0 ** -14
            assert bool((err <= tol).all()), (key, float((err / tol).max()))
        return out

    def wgrad(g, x, cout, cin, groups, M, views=1, pro_tab=None, pro_act=0, pro_slope=0.0, tile=-1, may_defer=True,
              out=None):
        # (checked right here against the reference: reduced at once, whatever the caller would allow; `out` -- the
        #  parameter's slice of a data-parallel flat buffer -- is honoured as the product does)
        dw = orig["_wgrad_bf16"](g, x, cout, cin, groups, M, views, pro_tab, pro_act, pro_slope, tile, may_defer=False,
                                 out=out)
        key = (cout, cin, groups, M, views, pro_tab is not None)
        if key not in ck.seen["wgrad"]:
            ck.seen["wgrad"].add(key)
            og, cg = cout // groups, cin // groups
            want = torch.zeros((cout, cg), dtype=torch.float64, device=g.device)
            step = max(128, (1 << 27) // max(cout, cin))                 # <= 1 GiB of float64 per operand chunk
            for m0 in range(0, M, step):
                gd, xd = g[:, m0:m0 + step].double(), x[:, m0:m0 + step].double()
                if pro_tab is not None:
                    view_of = torch.arange(m0, min(M, m0 + step), device=g.device) // (M // views)
                    xd = _pro(x[:, m0:m0 + step].float(), pro_tab, pro_act, pro_slope, view_of).double()
                for i in range(groups):
                    want[i * og:(i + 1) * og] += gd[i * og:(i + 1) * og] @ xd[i * cg:(i + 1) * cg].t()
            assert float((dw.double() - want).abs().max()) <= 2e-3 * float(want.abs().max()), key
        return dw

    def knn(x, k, normalize=True, layout="bcn", index_dtype=torch.int64, prefilter=None):
        idx = orig["knn_graph"](x, k, normalize, layout, index_dtype, prefilter)
        xs = x.detach()[:, clips].float().permute(1, 0, 2) if layout == "cbn" else x.detach()[clips].float()
        ck.graphs.append((xs.cpu().numpy(), idx[clips].cpu().numpy().astype(np.int64)))
        return idx

    ops.conv1x1_gemm, ops.conv1x1_gemm_cat, ops.bn_finalize_affine = gemm, gemm_cat, affine
    ops._bn_bwd, ops._wgrad_bf16, ops.knn_graph = bn_bwd, wgrad, knn
    try:
        w0 = model.encoder.backbone[0][1].fc1[0].weight.detach().clone()
        loss = tr.step(x_i, x_j)
        torch.cuda.synchronize()
    finally:
        for n, f in orig.items():
            setattr(ops, n, f)
    assert np.isfinite(float(loss)) and 0.0 < float(loss) < 20.0
    # every distinct launch shape of the dense chain went through its reference
    M0 = 2 * PAIRS * 1024
    assert len(ck.seen["gemm"]) >= 30 and len(ck.seen["cat"]) == 8, {k: len(v) for k, v in ck.seen.items()}
    assert len(ck.seen["affine"]) >= 16 and len(ck.seen["bn_bwd"]) >= 16 and len(ck.seen["wgrad"]) >= 20
    # stages 0-1: gfc2 and ffn2 normalise their operand on load (4 products, 4 weight gradients went through _pro)
    assert sum(k[-1] for k in ck.seen["gemm"]) == 4 and sum(k[-1] for k in ck.seen["wgrad"]) == 4
    assert any(k[3] == M0 for k in ck.seen["gemm"]) and any(k[3] == M0 for k in ck.seen["wgrad"])
    assert len(ck.graphs) == 12
    for feats, idx in ck.graphs:
        np.testing.assert_array_equal(native.knn_graph(feats, 3), idx)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    assert not torch.equal(model.encoder.backbone[0][1].fc1[0].weight.detach(), w0)
    # the launch plans that exist only at this size did fire (the plan queries mirror what the launches used)
    from grafp_amd._lib import lib
    import ctypes
    info = (ctypes.c_int * 8)()
    tiles, four_rounds = set(), 0
    for (R, K, groups, M, views, stats, _pro_on) in ck.seen["gemm"]:
        assert lib.grafp_conv1x1_gemm_plan(R, K, groups, M, views, info) == 0
        tiles.add(int(info[0]))
        four_rounds += int(info[4] >= 1024)
    assert tiles >= {0, 1, 3, 4} and four_rounds >= 1, (tiles, four_rounds)         # S, L, N64, N128 all ran
    cfgs = set()
    for (cout, cin, groups, M, views, _pro_on) in ck.seen["wgrad"]:
        assert lib.grafp_conv1x1_wgrad_plan(cout, cin, groups, M, views, info) == 0
        cfgs.add(int(info[0]))
    assert {6, 7} <= cfgs, cfgs                      # SG and LG: the register-staged tiles of >= 750 MB / wide layers
