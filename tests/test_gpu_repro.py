"""Bit-reproducibility of the training step.  `pytest -m gpu`.

The reference's step (/root/reference/train.py:56-82) run twice from one seed on one machine gives the same weights only
if every reduction has a fixed order.  Here every float sum of the step is a fixed expression: per-workgroup partials
added in index order (peak-extractor weight gradient, split-K weight gradients, BatchNorm statistics), fixed-point integer
atomics where a scatter is needed (max-relative backward), and no float atomics anywhere
(`grep atomicAdd grafp_amd/csrc` lists LDS counters, integer tickets and the fixed-point scatter only).  So two runs from
the same seed must end in the SAME BITS -- eager and as a replayed HIP graph, f32 and bf16 -- and the statistical tests
(tests/test_gpu_bf16.py) score the same models on every box.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _run(dev, steps, graph, amp, pairs=64, seed=0):
    from grafp_amd.train import Trainer, build_model, synthetic_batch
    from grafp_amd.util import load_config
    cfg = load_config()
    cfg["bsz_train"] = pairs
    torch.manual_seed(seed)
    model = build_model(cfg, device=dev)
    tr = Trainer(cfg, model, dev, amp_dtype=amp, lr=2e-4)
    losses = []
    for it in range(steps):
        x_i, x_j = synthetic_batch(pairs, 100 + it, dev)
        losses.append((tr.step_graph if graph else tr.step)(x_i, x_j).clone())
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    return torch.stack(losses), state, grads


def _differing(a, b):
    return [k for k in a if not torch.equal(a[k], b[k])]


@pytest.mark.parametrize("amp", [torch.bfloat16, None], ids=["bf16", "f32"])
@pytest.mark.parametrize("graph", [False, True], ids=["eager", "step_graph"])
def test_training_is_bit_reproducible(dev, graph, amp):
    l0, s0, g0 = _run(dev, 20, graph, amp)
    l1, s1, g1 = _run(dev, 20, graph, amp)
    assert torch.isfinite(l0).all()
    assert torch.equal(l0, l1), (l0 - l1).abs().max()
    bad_g = _differing(g0, g1)
    assert not bad_g, ("gradients of the last step differ", bad_g[:8], len(bad_g))
    bad = _differing(s0, s1)
    assert not bad, ("state differs after 20 steps", bad[:8], len(bad))


def test_one_backward_twice_gives_the_same_gradients(dev):
    """One forward/backward at 256 pairs (more clip-views than the peak extractor's 512 workgroups, the split-K plans of the
    larger batch) repeated from the same weights: every gradient tensor bit-equal."""
    from grafp_amd import ops
    from grafp_amd.simclr.ntxent import ntxent_loss
    from grafp_amd.train import build_model, synthetic_batch
    from grafp_amd.util import load_config
    cfg = load_config()
    cfg["bsz_train"] = 256
    torch.manual_seed(3)
    model = build_model(cfg, device=dev).train()
    x_i, x_j = synthetic_batch(256, 5, dev)
    X_i = ops.logmel(x_i, cfg["fs"], cfg["n_fft"], cfg["win_len"], cfg["hop_len"], cfg["n_mels"])
    X_j = ops.logmel(x_j, cfg["fs"], cfg["n_fft"], cfg["win_len"], cfg["hop_len"], cfg["n_mels"])
    bufs = {k: v.clone() for k, v in model.named_buffers()}
    runs = []
    for _ in range(2):
        for k, v in model.named_buffers():
            v.copy_(bufs[k])
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            _, _, z_i, z_j = model(X_i, X_j)
        with ops.defer_wgrad_reduce():
            ntxent_loss(z_i, z_j, cfg).backward()
        runs.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    bad = _differing(runs[0], runs[1])
    assert not bad, (bad[:8], len(bad))
