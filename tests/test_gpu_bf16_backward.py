"""Backward pass of the bf16 mode (the headline mode of bench.py) against an INDEPENDENT autograd.  `pytest -m gpu`.

The forward bars of the bf16 mode live in test_gpu_bf16.py.  Here the chain the training step actually runs --
ops._ConvBnAct.backward: bn_bwd1<bf16> -> data-gradient GEMM (or the [W^T | I] concatenated-operand GEMM that also
carries the shortcut's gradient, handed over by ops.ShortcutToken) -> split-K weight gradient; ops.max_relative's
backward; the stem's two-step BatchNorm; the Downsample tap gather -- is compared with torch's CPU autograd run on
`oracle.model.simclr_forward(..., q=round_bf16)` (train.py:66-74: loss.backward()).  `round_bf16` is two casts, so autograd
rounds the gradient passing through it to bf16 too -- exactly where the HIP path stores its gradients (dY behind the
BatchNorm backward, dX behind the data-gradient product); weight gradients stay f32 on both sides.

Layer by layer with the inputs held equal (teacher forcing, as in test_gpu_bf16.py): every stored activation of the HIP
forward pass is replaced by the oracle's version (a fresh leaf), and every layer's backward is then driven with the
ORACLE's gradient of that layer's output, last layer first -- so block b's ShortcutToken is filled by the block's last
layer and consumed by its first, as in a real step.  Compared per tensor (relative L2):
  * the gradient that arrives at every stored activation (= dX of its consumers, shortcut included),
  * dW, d(gamma), d(beta), d(conv bias) of every layer, the peak extractor's and the projector's parameters.
A block that dropped or double-counted its shortcut gradient is off by O(1) on its input leaf.
"""
import numpy as np
import pytest
import torch

from _common import RecordedGraphs, ReplayGraphs, filled_state_dict, simclr_inputs

pytestmark = pytest.mark.gpu

# measured on MI355X: activation gradients <= 2.4e-3 relative L2 (median 1.7e-3), parameter gradients <= 3.3e-3 (median
# 1.5e-5); one bf16 rounding step is 2^-9 = 2e-3 relative per element, and the two sides round sums formed in
# different orders -- the bars leave a factor of two
BAR_ACT = 5e-3
BAR_PARAM = 7e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _model(dev, B=4):
    from grafp_amd.train import build_model
    from grafp_amd.util import load_config
    cfg = load_config()
    cfg["bsz_train"] = B
    model = build_model(cfg)
    sd = model.state_dict()
    sd.update(filled_state_dict())
    model.load_state_dict(sd)
    return cfg, model.to(dev)


class _ForcedWithGrad:
    """Like test_gpu_bf16._Stored with `forced`, but the graph is kept: every wrapped op's output is recorded WITH its
    autograd history and replaced by a fresh leaf holding the oracle's activation."""

    def __init__(self, forced):
        self.forced, self.outs, self.leaves, self.errs = forced, [], [], []

    def __enter__(self):
        from grafp_amd import ops
        self._ops = ops
        self._orig = (ops.conv_bn_act, ops.bn_act, ops.max_relative)

        def wrap(fn):
            def inner(*a, **k):
                out = fn(*a, **k)
                want = self.forced[len(self.outs)].reshape(out.shape)
                got = out.detach().float().cpu()
                self.errs.append(float((got - want).norm() / want.norm()))
                self.outs.append(out)
                leaf = want.to(out.device, dtype=out.dtype).requires_grad_(True)
                self.leaves.append(leaf)
                return leaf
            return inner
        ops.conv_bn_act, ops.bn_act, ops.max_relative = (wrap(f) for f in self._orig)
        return self

    def __exit__(self, *exc):
        self._ops.conv_bn_act, self._ops.bn_act, self._ops.max_relative = self._orig


def _stack_views(ts):
    """The oracle runs the views one after the other: [layers of view i] + [layers of view j], each (B, C, N, 1) ->
    per layer the (C, 2B, N) tensor the HIP path holds."""
    nl = len(ts) // 2
    return [torch.cat([ts[i], ts[nl + i]], dim=0).squeeze(-1).permute(1, 0, 2).contiguous() for i in range(nl)]


def _rel(got, want):
    return float((got - want).norm() / want.norm())


@pytest.mark.parametrize("fused", [True, False], ids=["shortcut_token", "autograd_accumulate"])
def test_bf16_backward_layer_by_layer_vs_oracle_autograd(dev, monkeypatch, fused):
    from grafp_amd.simclr.ntxent import ntxent_loss
    from oracle import model as om
    from grafp_amd import ops
    monkeypatch.setattr(ops.switches, "shortcut_fusion", bool(fused))
    monkeypatch.setattr(ops.switches, "defer_norm", False)     # layer-by-layer: every layer stores its own output
    cfg, model = _model(dev)
    model.train()
    xi, xj = simclr_inputs()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    # the graphs the HIP path builds on this input (free-running pass), held equal on both sides afterwards
    with RecordedGraphs() as rg, torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        model(xi.to(dev), xj.to(dev))
    model.load_state_dict({k: v.to(dev) for k, v in sd.items()})

    # ---- the independent side: CPU autograd through the bf16-storage restatement ----
    sd_o = {k: v.clone() for k, v in sd.items()}
    for k in om.trainable(sd_o):
        if sd_o[k].is_floating_point():
            sd_o[k].requires_grad_(True)
    taps = []

    def tap(name, t):
        t.retain_grad()
        taps.append(t)
        return t
    _, _, oz_i, oz_j = om.simclr_forward(sd_o, xi, xj, True, idx_fn=rg.replay_fn(), q=om.round_bf16, tap=tap)
    loss_o = om.ntxent(oz_i, oz_j, cfg["tau"])
    loss_o.backward()
    stored = _stack_views([t.detach() for t in taps])
    stored_grad = _stack_views([t.grad for t in taps])
    assert len(stored) == 1 + 12 * 6 + 3

    # ---- the HIP side: forward with the oracle's activations substituted, graph kept ----
    with ReplayGraphs(rg.graphs), _ForcedWithGrad(stored) as f, torch.autocast("cuda", dtype=torch.bfloat16):
        _, _, z_i, z_j = model(xi.to(dev), xj.to(dev))
        loss = ntxent_loss(z_i, z_j, cfg)
    assert len(f.outs) == len(stored) and max(f.errs) <= 1e-3, max(f.errs)
    assert abs(float(loss) - float(loss_o)) <= 1e-4 * abs(float(loss_o))
    loss.backward()                                    # readout + projector + NT-Xent: down to the last forced leaf
    if fused:
        tokens_used = []
        orig_cat = ops.conv1x1_gemm_cat
        monkeypatch.setattr(ops, "conv1x1_gemm_cat", lambda *a, **k: (tokens_used.append(1), orig_cat(*a, **k))[1])
    for i in reversed(range(len(f.outs))):
        g = stored_grad[i].reshape(f.outs[i].shape).to(dev, dtype=f.outs[i].dtype)
        f.outs[i].backward(g)
    torch.cuda.synchronize()
    if fused:
        assert len(tokens_used) == 24                  # every Grapher and every FFN block handed its shortcut over

    # ---- gradients arriving at the stored activations ----
    errs = np.array([_rel(f.leaves[i].grad.float().cpu(), stored_grad[i].reshape(f.leaves[i].shape))
                     for i in range(len(stored))])
    print("activation gradients: max %.3e (layer %d), median %.3e" % (errs.max(), int(errs.argmax()), np.median(errs)))
    assert errs.max() <= BAR_ACT, (int(errs.argmax()), errs.max())

    # ---- parameter gradients ----
    got = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}
    want = {n: t.grad for n, t in sd_o.items() if t.requires_grad and t.grad is not None}
    assert set(got) == set(want), set(got) ^ set(want)
    top = max(float(v.norm()) for v in want.values())
    rows = []
    for n in sorted(want):
        wn = float(want[n].norm())
        if wn < 1e-4 * top:
            # mathematically zero gradients (a conv bias in front of batch statistics; the BatchNorm bias of a Grapher's
            # fc1, whose per-channel constant the max-relative differences and the next BatchNorm remove): rounding
            # noise on both sides -- only required to BE small
            assert float(got[n].norm()) <= 1e-3 * top, (n, float(got[n].norm()), top)
            continue
        rows.append((_rel(got[n].reshape(want[n].shape), want[n]), n))
    rows.sort(reverse=True)
    print("parameter gradients: %d compared, worst %.3e (%s), median %.3e" % (len(rows), rows[0][0], rows[0][1],
                                                                              rows[len(rows) // 2][0]))
    assert len(rows) >= 150
    assert rows[0][0] <= BAR_PARAM, rows[:5]
