"""GPU tests of the assembled path: module mirror + HIP ops against the reference's golden outputs and the
oracle; retrieval evaluation end to end; database writers; bf16 mode.  `pytest -m gpu`."""
import os

import numpy as np
import pytest
import torch

from _common import (RecordedGraphs, ReplayGraphs, filled_state_dict, golden, hash_normalish, reference_graphs,
                     simclr_inputs)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _filled_model(dev, B=4):
    from grafp_amd.train import build_model
    from grafp_amd.util import load_config
    cfg = load_config()
    cfg["bsz_train"] = B
    model = build_model(cfg)
    sd = model.state_dict()
    sd.update(filled_state_dict())
    model.load_state_dict(sd)
    return cfg, model.to(dev)


def _rel_l2(got, want):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    want = want.detach().cpu().numpy() if torch.is_tensor(want) else want
    return (np.linalg.norm(got - want, axis=-1) / np.linalg.norm(want, axis=-1)).max()


def test_simclr_forward_vs_oracle_with_equal_edges(dev):
    """f32 parity bar: with the k-NN edges held equal (replayed in the oracle), every embedding agrees to
    1e-4 relative L2 (BASELINE bar: 1e-3), train and eval mode."""
    from oracle import model as om
    cfg, model = _filled_model(dev)
    xi, xj = simclr_inputs()
    for train in (True, False):
        model.train(train)
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        with RecordedGraphs() as rg, torch.no_grad():
            h_i, h_j, z_i, z_j = model(xi.to(dev), xj.to(dev))
        assert len(rg.graphs) == 12                 # one graph per block over the two stacked views
        with torch.no_grad():
            o = om.simclr_forward(sd, xi, xj, train, idx_fn=rg.replay_fn())
        for got, want in zip((h_i, h_j, z_i, z_j), o):
            assert _rel_l2(got, want) <= 1e-4
        assert rg.flips <= 200                      # near-ties only: <= 0.2% of the ~98k node decisions
        if train:
            np.testing.assert_allclose(model.encoder.stem[1].running_mean.cpu().numpy(),
                                       sd["encoder.stem.1.running_mean"].numpy(), rtol=1e-5, atol=1e-6)


def test_simclr_forward_matches_reference_golden(dev):
    """The reference's own SimCLR.forward output (tests/golden/simclr_forward.npz, produced on CPU) with the
    reference's edges replayed into the HIP model: 1e-4 relative L2 per vector, train and eval mode.
    (With its own graphs the random-weight network is chaotic: one near-tie neighbour flipping in block 8 of 12
    grows to 0.1 at the output -- measured -- so own-graph outputs are only sanity-checked for direction; the
    k-NN decision itself is verified bit-exactly on identical inputs in test_gpu_kernels.py / smoke().)"""
    g = golden("simclr_forward.npz")
    cfg, model = _filled_model(dev)
    xi, xj = simclr_inputs()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    model.train()
    with ReplayGraphs(reference_graphs(sd, xi, xj, True), views=2), torch.no_grad():
        h_i, h_j, z_i, z_j = model(xi.to(dev), xj.to(dev))
    for got, want in ((z_i, g["z_i"]), (z_j, g["z_j"]), (h_i, g["h_i"]), (h_j, g["h_j"])):
        assert _rel_l2(got, want) <= 1e-4
    np.testing.assert_allclose(model.encoder.stem[1].running_mean.cpu().numpy(), g["stem_running_mean"], rtol=1e-5, atol=1e-6)
    model.eval()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with ReplayGraphs(reference_graphs(sd, xi, xj, False), views=2), torch.no_grad():
        eh_i, _, ez_i, ez_j = model(xi.to(dev), xj.to(dev))
    # eval mode runs on the running statistics this model accumulated on the GPU (equal to the reference's only to
    # rounding), so the replayed graphs can differ from the golden run's by a near-tie: 5e-3 instead of 1e-4
    for got, want in ((ez_i, g["eval_z_i"]), (ez_j, g["eval_z_j"]), (eh_i, g["eval_h_i"])):
        assert _rel_l2(got, want) <= 5e-3
    with torch.no_grad():                                   # own graphs: direction only
        _, _, oz_i, oz_j = model(xi.to(dev), xj.to(dev))
    cos = torch.nn.functional.cosine_similarity(torch.cat([oz_i, oz_j]).cpu(), torch.from_numpy(np.concatenate([g["eval_z_i"], g["eval_z_j"]])))
    assert float(cos.min()) > 0.97, cos


def test_train_step_vs_oracle_with_equal_edges(dev):
    """One full step (fwd + NT-Xent + bwd) against the oracle's autograd with edges held equal: loss 1e-5,
    every parameter gradient within 3e-2 relative L2 (median 8e-3)."""
    from grafp_amd.simclr.ntxent import ntxent_loss
    from oracle import model as om
    cfg, model = _filled_model(dev)
    model.train()
    xi, xj = simclr_inputs()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for k, v in sd.items():
        if v.is_floating_point() and k.rsplit(".", 1)[-1] not in ("running_mean", "running_var", "relative_pos"):
            v.requires_grad_(True)
    with RecordedGraphs() as rg:
        _, _, z_i, z_j = model(xi.to(dev), xj.to(dev))
    loss = ntxent_loss(z_i, z_j, cfg)
    loss.backward()
    _, _, oz_i, oz_j = om.simclr_forward(sd, xi, xj, True, idx_fn=rg.replay_fn())
    oloss = om.ntxent(oz_i, oz_j, cfg["tau"])
    oloss.backward()
    np.testing.assert_allclose(loss.item(), oloss.item(), rtol=1e-4)
    # Metric: relative L2 error per gradient tensor.  (Max-abs is the wrong yardstick: a pre-activation within
    # rounding of zero flips a ReLU / arg-max mask -- about one per layer among ~1e6 activations, also between two
    # CPU formulations of the same network -- and moves single entries by a few % while leaving norms intact.)
    # Conv biases feeding a train-mode BatchNorm have a mathematically ZERO gradient (pure rounding noise on both
    # sides): tensors whose norm is < 1e-4 of the largest are compared on that absolute scale.
    gnorm = max(float(sd[n].grad.norm()) for n, p in model.named_parameters() if p.requires_grad)
    rows = []
    for name, p in model.named_parameters():
        if p.requires_grad:
            want = sd[name].grad
            rows.append((float((p.grad.cpu() - want).norm()) / max(float(want.norm()), 1e-4 * gnorm), name))
    rows.sort(reverse=True)
    assert rows[0][0] <= 3e-2, rows[:5]          # earliest layers accumulate every downstream mask flip
    assert np.median([r[0] for r in rows]) <= 8e-3


def test_train_step_matches_reference_golden(dev):
    """The reference's own train step (golden produced on CPU), reference edges replayed: loss 1e-4, gradient
    norms 1e-2, two full gradient tensors 2e-2 relative L2, post-Adam parameter sums."""
    from grafp_amd.simclr.ntxent import ntxent_loss
    g = golden("train_step.npz")
    cfg, model = _filled_model(dev)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=8e-5)
    xi, xj = simclr_inputs()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    opt.zero_grad()
    with ReplayGraphs(reference_graphs(sd, xi, xj, True), views=2):
        _, _, z_i, z_j = model(xi.to(dev), xj.to(dev))
    loss = ntxent_loss(z_i, z_j, cfg)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    probe = [str(p) for p in g["probe"]]
    params = dict(model.named_parameters())
    gn = np.array([params[k].grad.double().norm().item() for k in probe])
    np.testing.assert_allclose(gn, g["grad_norm"], rtol=1e-2)
    for k in probe[:2]:
        want = g["grad:" + k]
        assert np.linalg.norm(params[k].grad.cpu().numpy() - want) <= 2e-2 * np.linalg.norm(want)
    opt.step()
    # Adam's first step moves every weight by ~lr * sign(grad): entries whose gradient is ~0 can take the other
    # sign after a single mask flip, so the sums agree to a few 1e-3 absolute rather than to rounding
    ps = np.array([params[k].detach().double().sum().item() for k in probe])
    np.testing.assert_allclose(ps, g["param_sum_after"], rtol=1e-4, atol=2e-2)


def test_block_golden(dev):
    """One Grapher+FFN block (C=16, N=64) against the reference, train and eval."""
    from torch import nn
    from grafp_amd.encoder.gcn_lib.torch_vertex import Grapher
    from grafp_amd.encoder.graph_encoder import FFN
    g = golden("block.npz")
    C, N = 16, 64
    blk = nn.Sequential(Grapher(C, 3, 1, "mr", "relu", "batch", True, False, 0.2, 1, n=N, drop_path=0.0, relative_pos=True),
                        FFN(C, 4 * C, C, act="relu"))
    assert sorted(blk.state_dict().keys()) == [str(k) for k in g["keys"]]
    shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
    sd = blk.state_dict(); sd.update(filled_state_dict(shapes, "blk")); blk.load_state_dict(sd)
    blk = blk.to(dev)
    x = torch.from_numpy(hash_normalish("in:blk.x", (3, C, N, 1))).to(dev)
    blk.train()
    with torch.no_grad():
        y = blk(x)
    assert y.shape == (3, C, N, 1)
    np.testing.assert_allclose(y.cpu().numpy(), g["y_train"], rtol=2e-5, atol=2e-5)
    blk.eval()
    with torch.no_grad():
        y = blk(x)
    np.testing.assert_allclose(y.cpu().numpy(), g["y_eval"], rtol=2e-5, atol=2e-5)


def test_bf16_autocast_mode(dev):
    """Throughput mode: bf16 GEMMs (autocast), f32 graph build / gather / loss.  The 1e-3 embedding bar is an
    f32 property: under bf16 ~13% of the nodes change a neighbour in block 1 and nearly all by block 12, and a
    random-init network amplifies the rounding itself (measured, scratch analysis in DESIGN.md).  Checked here:
    the mode runs end to end, stays finite, and with the f32 graphs replayed stays directionally close."""
    cfg, model = _filled_model(dev)
    xi, xj = simclr_inputs()
    model.train()
    with RecordedGraphs() as rg, torch.no_grad():
        _, _, z32, _ = model(xi.to(dev), xj.to(dev))
    from grafp_amd import ops
    with ReplayGraphs(rg.graphs), torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        _, _, z16r, _ = model(xi.to(dev), xj.to(dev))
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        _, _, z16, _ = model(xi.to(dev), xj.to(dev))
    cos_r = torch.nn.functional.cosine_similarity(z32.float(), z16r.float(), dim=1)
    cos = torch.nn.functional.cosine_similarity(z32.float(), z16.float(), dim=1)
    print("bf16 vs f32 cosine: own graphs", cos.cpu().numpy(), "f32 graphs replayed", cos_r.cpu().numpy())
    assert torch.isfinite(z16).all() and float(cos_r.min()) > 0.95
    np.testing.assert_allclose(z16.float().norm(dim=1).cpu().numpy(), 1.0, rtol=1e-2)


def test_trainer_reduces_loss(dev):
    from grafp_amd.train import Trainer, build_model, synthetic_batch
    from grafp_amd.util import load_config
    cfg = load_config(); cfg["bsz_train"] = 16
    torch.manual_seed(0)
    model = build_model(cfg, device=dev)
    tr = Trainer(cfg, model, dev, lr=2e-4)
    x_i, x_j = synthetic_batch(16, 5, dev)
    losses = [float(tr.step(x_i, x_j)) for _ in range(16)]
    assert all(np.isfinite(losses)) and np.mean(losses[-4:]) < np.mean(losses[:4]), losses
    ck = tr.checkpoint(1, losses, [])
    assert set(ck) == {"epoch", "loss", "valid_acc", "hit_rate", "state_dict", "optimizer", "scheduler"}
    assert len(ck["state_dict"]) == 443


def test_eval_faiss_end_to_end_vs_oracle(dev, tmp_path):
    """query/db/dummy_db memmaps -> eval_faiss (GPU search, batched) == the oracle's restatement of
    eval.py (per-item search + rerank): identical hit-rate table and raw flags."""
    from grafp_amd.eval import eval_faiss
    from grafp_amd.fpdb import _write_memmap
    from oracle import retrieval
    dummy = hash_normalish("ev:dummy", (3000, 128)); dummy /= np.linalg.norm(dummy, axis=1, keepdims=True)
    db = hash_normalish("ev:db", (400, 128)); db /= np.linalg.norm(db, axis=1, keepdims=True)
    query = db + 0.12 * hash_normalish("ev:noise", (400, 128)); query /= np.linalg.norm(query, axis=1, keepdims=True)
    for name, arr in (("dummy_db", dummy), ("db", db), ("query", query)):
        _write_memmap(str(tmp_path / name), arr.astype(np.float32))
    test_ids = np.arange(0, 350, 7)
    np.save(tmp_path / "ids.npy", test_ids)
    rates = eval_faiss(str(tmp_path), test_ids=str(tmp_path / "ids.npy"), test_seq_len="1 3 5 11", index_type="l2",
                       nogpu=True)
    want, raw, _ = retrieval.eval_l2(query.astype(np.float32), db.astype(np.float32), dummy.astype(np.float32),
                                     test_ids, [1, 3, 5, 11])
    assert rates.shape == (4, 4)
    np.testing.assert_array_equal(rates, want)
    res_dirs = [d for d in os.listdir(tmp_path) if os.path.isdir(tmp_path / d)]
    assert len(res_dirs) == 1
    np.testing.assert_array_equal(np.load(tmp_path / res_dirs[0] / "raw_score.npy"), raw)
    assert np.array_equal(np.load(tmp_path / "test_ids.npy"), test_ids)
    # 'all' and numeric id selections run too
    r2 = eval_faiss(str(tmp_path), test_ids="20", test_seq_len=[1, 5], index_type="ivfpq")
    assert r2.shape == (4, 2) and (r2[3] >= r2[0]).all()


def test_eval_faiss_matches_reference_golden(dev, tmp_path):
    """eval_faiss (one batched search + one rerank launch on the GPU) on the hash-filled case == the hit-rate table
    and raw flags the reference's own eval_faiss produced (tests/golden/eval_faiss.npz)."""
    from _common import eval_case, golden, write_eval_case
    from grafp_amd.eval import eval_faiss
    g = golden("eval_faiss.npz")
    case = eval_case()
    write_eval_case(str(tmp_path), case)
    np.save(tmp_path / "ids.npy", case["test_ids"])
    rates = eval_faiss(str(tmp_path), test_ids=str(tmp_path / "ids.npy"), test_seq_len=case["test_seq_len"],
                       index_type="l2", k_probe=case["k_probe"])
    np.testing.assert_array_equal(rates, g["hit_rates"])
    res_dirs = [d for d in os.listdir(tmp_path) if os.path.isdir(tmp_path / d)]
    np.testing.assert_array_equal(np.load(tmp_path / res_dirs[0] / "raw_score.npy"), g["raw_score"])


def test_db_writers(dev, tmp_path):
    """create_dummy_db / create_fp_db / create_db: on-disk format and contents == direct model output."""
    from grafp_amd.fpdb import create_db, create_dummy_db, create_fp_db
    from grafp_amd.eval import load_memmap_data
    from grafp_amd.modules.transformations import GPUTransformNeuralfp
    cfg, model = _filled_model(dev)
    model.eval()
    aug = GPUTransformNeuralfp(cfg, None, None, train=False)
    tracks = [torch.from_numpy(0.1 * hash_normalish(f"dbw:{i}", (1, 16000 * 3 + 500 * i))) for i in range(3)]
    create_dummy_db(tracks, augment=aug, model=model, output_root_dir=str(tmp_path), verbose=False)
    create_fp_db(tracks, augment=aug, model=model, output_root_dir=str(tmp_path), verbose=False)
    create_db(tracks, model, aug, str(tmp_path))
    dd, shape = load_memmap_data(str(tmp_path), "dummy_db", display=False)
    n_seg = sum((1 + t.shape[1] // 512 - 32) // 3 + 1 for t in tracks)
    assert tuple(shape) == (n_seg, 128)
    q, _ = load_memmap_data(str(tmp_path), "query", display=False)
    d, _ = load_memmap_data(str(tmp_path), "db", display=False)
    fp = np.load(tmp_path / "fingerprints.npy")
    with torch.no_grad():
        want = torch.cat([model.embed(aug(t.to(dev), None)[0])[1] for t in tracks]).cpu().numpy()
    np.testing.assert_allclose(np.asarray(dd), want, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(fp, want, rtol=1e-5, atol=1e-6)
    # create_fp_db embeds (x, x) as one stacked two-view batch: other GEMM shapes -> other rounding -> a near-tie
    # neighbour may flip; each fingerprint still agrees to 5e-3 relative L2, and db == query row for row
    for got in (np.asarray(d), np.asarray(q)):
        assert (np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)).max() <= 5e-3
    np.testing.assert_allclose(np.asarray(q), np.asarray(d), rtol=1e-5, atol=1e-6)     # identity augmentation
    np.testing.assert_allclose(np.linalg.norm(want, axis=1), 1.0, rtol=1e-5)
    # train mode (what the reference's scripts run in): one model call per track chunk, BatchNorm on that call's
    # statistics -- the writers must reproduce exactly that batching
    model.train()
    create_dummy_db(tracks, augment=aug, model=model, output_root_dir=str(tmp_path), fname="dummy_train", verbose=False)
    dt, _ = load_memmap_data(str(tmp_path), "dummy_train", display=False)
    with torch.no_grad():
        want_t = torch.cat([model.embed(aug(t.to(dev), None)[0])[1] for t in tracks]).cpu().numpy()
    np.testing.assert_allclose(np.asarray(dt), want_t, rtol=1e-5, atol=1e-6)
    assert np.abs(want_t - want).max() > 1e-4                                          # the two modes do differ
    model.eval()
    # opt-in packing of several tracks per model call (eval mode): same fingerprints up to GEMM rounding / near-tie flips
    create_db(tracks, model, aug, str(tmp_path), concat=False, max_segments=1024)
    parts = np.load(tmp_path / "fingerprints.npy", allow_pickle=True)
    assert len(parts) == len(tracks) and sum(len(p) for p in parts) == n_seg
    packed = np.concatenate(list(parts), axis=0)
    with torch.no_grad():        # the train-mode pass above advanced the running statistics: fresh eval reference
        want_e = torch.cat([model.embed(aug(t.to(dev), None)[0])[1] for t in tracks]).cpu().numpy()
    assert (np.linalg.norm(packed - want_e, axis=1) / np.linalg.norm(want_e, axis=1)).max() <= 5e-3


def test_trainer_step_graph_equals_eager(dev):
    """Trainer.step_graph (whole step replayed from one HIP graph): from the SAME weights and optimizer state, a
    replayed step returns the loss an eager forward pass computes on the new batch and leaves the parameters where
    an eager step leaves them."""
    from grafp_amd.simclr.ntxent import ntxent_loss
    from grafp_amd.train import Trainer, build_model, synthetic_batch
    from grafp_amd.util import load_config
    cfg = load_config()
    cfg["bsz_train"] = 16
    torch.manual_seed(7)
    model = build_model(cfg, device=dev)
    tr = Trainer(cfg, model, dev, amp_dtype=None)
    tr.step_graph(*synthetic_batch(16, 50, dev))                  # 3 eager warm-up steps, capture, first replay

    def snapshot():
        return ([p.detach().clone() for p in model.parameters()], [b.detach().clone() for b in model.buffers()],
                [{k: v.detach().clone() for k, v in st.items() if torch.is_tensor(v)} for st in tr.opt.state.values()])

    def restore(snap):                                            # in place: the graph holds these addresses
        with torch.no_grad():
            for p, v in zip(model.parameters(), snap[0]):
                p.copy_(v)
            for b, v in zip(model.buffers(), snap[1]):
                b.copy_(v)
            for st, sv in zip(tr.opt.state.values(), snap[2]):
                for k, v in sv.items():
                    st[k].copy_(v)

    for seed in (51, 52):
        xi, xj = synthetic_batch(16, seed, dev)
        snap = snapshot()
        with torch.no_grad():                                     # eager forward loss at these weights
            X_i, X_j = tr.augment(xi, xj)
            _, _, z_i, z_j = model(X_i, X_j)
            want_loss = float(ntxent_loss(z_i, z_j, cfg))
        restore(snap)                                             # (the forward advanced the running statistics)
        loss_e = float(tr.step(xi, xj))
        p_e = torch.cat([p.detach().flatten() for p in model.parameters()]).clone()
        restore(snap)
        loss_g = float(tr.step_graph(xi, xj))
        p_g = torch.cat([p.detach().flatten() for p in model.parameters()]).clone()
        p_0 = torch.cat([v.flatten() for v in snap[0]])
        assert abs(loss_e - want_loss) <= 1e-5 * max(1.0, abs(want_loss))
        assert abs(loss_g - want_loss) <= 1e-5 * max(1.0, abs(want_loss))
        d_e, d_g = p_e - p_0, p_g - p_0
        assert float(d_e.norm()) > 0 and float((d_g - d_e).norm() / d_e.norm()) < 0.05     # atomics: not bit-equal


def test_step_graph_follows_the_scheduler_and_restores_on_failure(dev, monkeypatch):
    """The learning rate of the captured Adam is a device tensor read at REPLAY time: CosineAnnealingLR (train.py:175,
    224) keeps steering a replayed step (lr = 0 moves nothing, the scheduler's next value moves the weights by the
    ratio of the rates).  If the warm-up or the capture raises, the warm-up steps are undone."""
    from grafp_amd.train import Trainer, build_model, synthetic_batch
    from grafp_amd.util import load_config
    cfg = load_config()
    cfg["bsz_train"] = 8
    torch.manual_seed(7)
    model = build_model(cfg, device=dev)
    tr = Trainer(cfg, model, dev, amp_dtype=torch.bfloat16)
    xi, xj = synthetic_batch(8, 60, dev)

    def flat():
        return torch.cat([p.detach().flatten() for p in model.parameters()]).clone()
    # a capture that fails after the warm-up steps leaves weights, BatchNorm statistics and Adam state untouched
    p0, b0 = flat(), [b.detach().clone() for b in model.buffers()]
    calls = {"n": 0}
    orig_step = tr.step

    def failing_step(a, b):
        calls["n"] += 1
        if calls["n"] == 3:
            raise RuntimeError("injected")
        return orig_step(a, b)
    monkeypatch.setattr(tr, "step", failing_step)
    with pytest.raises(RuntimeError, match="injected"):
        tr.step_graph(xi, xj)
    monkeypatch.undo()
    assert tr._graph is None and torch.equal(flat(), p0)
    assert all(torch.equal(a, b) for a, b in zip(model.buffers(), b0))
    # capture for real; then drive the rate
    tr.step_graph(xi, xj)
    lr = tr.opt.param_groups[0]["lr"]
    assert torch.is_tensor(lr) and lr.is_cuda
    lr.fill_(0.0)
    before = flat()
    tr.step_graph(xi, xj)
    assert torch.equal(flat(), before)                       # lr = 0: Adam's update is exactly zero
    lr.fill_(cfg["lr"])
    tr.sched.step()                                          # in place on the same tensor
    assert tr.opt.param_groups[0]["lr"] is lr and 0.0 < float(lr) < cfg["lr"]
    tr.step_graph(xi, xj)
    moved = float((flat() - before).abs().max())
    assert 0.0 < moved <= 10.0 * float(lr)                   # Adam: |update| ~ lr per element


def test_resume_then_step_graph_keeps_the_scheduler_in_charge(dev, tmp_path):
    """optimizer.load_state_dict REPLACES param_group['lr'] (a float from a reference-format checkpoint, another
    tensor from an own one).  Trainer.load_checkpoint -- and util.load_ckp on trainer.opt followed by any step -- put the
    VALUE into the Trainer's own device tensor, so a capture after the resume still reads the rate at replay time; a resume
    AFTER the capture keeps the captured state tensors.  checkpoint() stores the rate as a float (train.py:212-220)."""
    from grafp_amd.train import Trainer, build_model, synthetic_batch
    from grafp_amd.util import load_ckp, load_config, save_ckp
    cfg = load_config()
    cfg["bsz_train"] = 8
    torch.manual_seed(9)
    model = build_model(cfg, device=dev)
    tr = Trainer(cfg, model, dev, amp_dtype=torch.bfloat16)
    xi, xj = synthetic_batch(8, 61, dev)
    tr.step(xi, xj)
    tr.sched.step()
    ckp = tr.checkpoint(1, [0.0], [0.0])
    lr_saved = ckp["optimizer"]["param_groups"][0]["lr"]
    assert isinstance(lr_saved, float) and 0.0 < lr_saved < cfg["lr"]
    save_ckp(ckp, "m", str(tmp_path), "1")
    path = str(tmp_path / "model_m_1.pth")

    def flat(m):
        return torch.cat([p.detach().flatten() for p in m.parameters()]).clone()

    for how in ("load_checkpoint", "util.load_ckp"):
        torch.manual_seed(10)
        m2 = build_model(cfg, device=dev)
        t2 = Trainer(cfg, m2, dev, amp_dtype=torch.bfloat16)
        if how == "load_checkpoint":
            t2.load_checkpoint(path)
        else:
            load_ckp(path, m2, optimizer=t2.opt, scheduler=t2.sched, map_location=dev)
            assert t2.opt.param_groups[0]["lr"] is not t2._lr          # what torch does; the next step repairs it
        assert torch.equal(flat(m2), flat(model))
        t2.step_graph(xi, xj)                                          # capture AFTER the resume
        lr = t2.opt.param_groups[0]["lr"]
        assert lr is t2._lr and abs(float(lr) - lr_saved) < 1e-12 * max(1.0, lr_saved) + 1e-10
        lr.fill_(0.0)
        before = flat(m2)
        t2.step_graph(xi, xj)
        assert torch.equal(flat(m2), before)                           # the replay READS the tensor: lr = 0 moves nothing
        lr.fill_(lr_saved)
        t2.sched.step()
        assert t2.opt.param_groups[0]["lr"] is t2._lr and 0.0 < float(lr) < lr_saved
        t2.step_graph(xi, xj)
        moved = float((flat(m2) - before).abs().max())
        assert 0.0 < moved <= 10.0 * float(lr)
    # resume AFTER the capture: the graph keeps running on its own state tensors, now holding the checkpoint's values
    t2.load_checkpoint(path)
    assert torch.equal(flat(m2), flat(model)) and t2.opt.param_groups[0]["lr"] is t2._lr
    p0 = next(iter(tr.opt.state))
    q0 = next(iter(t2.opt.state))
    assert torch.equal(tr.opt.state[p0]["exp_avg"], t2.opt.state[q0]["exp_avg"])
    t2._lr.fill_(0.0)
    before = flat(m2)
    t2.step_graph(xi, xj)
    assert torch.equal(flat(m2), before)


def test_backward_after_another_forward_refuses_the_shared_weight_copies(dev):
    """The bf16 / transposed weight copies of ops.lowp_weights are shared between passes and rewritten by every encoder
    forward under autocast: a backward pass whose forward saw an older preparation raises instead of
    differentiating against whatever the buffers hold by then; the normal order works."""
    cfg, model = _filled_model(dev)
    model.train()
    xi, xj = simclr_inputs()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        _, _, z_i, _ = model(xi.to(dev), xj.to(dev))
        model(xi.to(dev), xj.to(dev))                      # e.g. an interleaved validation / fingerprint pass
    with pytest.raises(RuntimeError, match="re-prepared"):
        z_i.sum().backward()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        _, _, z_i, _ = model(xi.to(dev), xj.to(dev))
    z_i.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    # the counter belongs to the ENCODER INSTANCE whose buffers it guards -- a forward pass of ANOTHER model
    # (a teacher / EMA copy, a second Trainer) between this model's forward and backward touches none of them
    _, other = _filled_model(dev)
    for p in model.parameters():
        p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        _, _, z_i, _ = model(xi.to(dev), xj.to(dev))
        with torch.no_grad():
            other(xi.to(dev), xj.to(dev))
    z_i.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_eval_mode_fused_affine_epilogue_equals_two_kernels(dev, monkeypatch):
    """Fingerprint generation (eval mode, bf16): the layers without a shortcut take the GEMM with the normalisation in its
    epilogue -- same embeddings, bit for bit, as GEMM + normalise pass."""
    from grafp_amd import ops
    cfg, model = _filled_model(dev)
    model.eval()
    xi, _ = simclr_inputs()
    calls = []
    orig = ops.conv1x1_gemm_affine
    monkeypatch.setattr(ops, "conv1x1_gemm_affine", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        h1, z1 = model.embed(xi.to(dev))
        n_fused = len(calls)
        monkeypatch.setattr(ops.switches, "fused_eval_affine", False)
        h0, z0 = model.embed(xi.to(dev))
    assert n_fused == 12 * 3 + 3 and len(calls) == n_fused        # fc1, grouped conv, ffn1 of every block + 3 Downsamples
    assert torch.equal(h1, h0) and torch.equal(z1, z0)


def test_trainer_with_device_augmentation_eager_and_graph(dev):
    """The training step with the second view augmented on the device (impulse responses + background noise for
    every clip): runs eagerly and replayed from one HIP graph (the per-clip draws use the device generator, which
    advances across replays), the loss stays finite and the augmented view differs from the clean one."""
    from grafp_amd.train import Trainer, build_model, synthetic_batch
    from grafp_amd.util import load_config
    cfg = load_config(); cfg["bsz_train"] = 8
    torch.manual_seed(0)
    model = build_model(cfg, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    irs = torch.randn(4, 3000, generator=gen, device=dev) * torch.exp(-torch.arange(3000, device=dev) / 500.0)
    noise = torch.randn(3, 40000, generator=gen, device=dev)
    tr = Trainer(cfg, model, dev, amp_dtype=None, ir_dir=irs, noise_dir=noise)
    x_i, x_j = synthetic_batch(8, seed=3, device=dev)
    a, b = tr.augment(x_i, x_j)
    assert not torch.equal(b, tr.augment.logmelspec(x_j))
    l0 = float(tr.step(x_i, x_j))
    losses = [float(tr.step_graph(x_i, x_j)) for _ in range(3)]
    assert np.isfinite(l0) and all(np.isfinite(v) for v in losses)
    assert len(set(losses)) > 1                      # new noise draws (and new weights) on every replay


def test_full_step_at_config2_size(dev):
    """BASELINE config 2 at its real size (256 pairs = 512 clip-views, bf16): one full step through Trainer.step --
    finite loss and gradients, every one of the 12 k-NN graphs bit-exact against the C oracle on sampled clips (the
    decision is re-made from the features the HIP path itself fed the graph kernel), weights move."""
    from grafp_amd import ops
    from grafp_amd.train import Trainer, build_model, synthetic_batch
    from grafp_amd.util import load_config
    from oracle import native
    cfg = load_config(); cfg["bsz_train"] = 256
    torch.manual_seed(3)
    model = build_model(cfg, device=dev)
    tr = Trainer(cfg, model, dev, amp_dtype=torch.bfloat16)
    x_i, x_j = synthetic_batch(256, seed=9, device=dev)
    seen, orig = [], ops.knn_graph
    clips = [0, 17, 255, 256, 300, 511]

    def rec(x, k, normalize=True, layout="bcn", index_dtype=torch.int64, prefilter=None):
        idx = orig(x, k, normalize, layout, index_dtype, prefilter)
        xs = x.detach()[:, clips].float().permute(1, 0, 2) if layout == "cbn" else x.detach()[clips].float()
        seen.append((xs.cpu().numpy(), idx[clips].cpu().numpy().astype(np.int64)))
        return idx
    ops.knn_graph = rec
    try:
        w0 = model.encoder.backbone[0][1].fc1[0].weight.detach().clone()
        loss = tr.step(x_i, x_j)
    finally:
        ops.knn_graph = orig
    assert np.isfinite(float(loss)) and 0.0 < float(loss) < 20.0
    assert len(seen) == 12
    for feats, idx in seen:
        np.testing.assert_array_equal(native.knn_graph(feats, 3), idx)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    assert not torch.equal(model.encoder.backbone[0][1].fc1[0].weight.detach(), w0)


@pytest.mark.parametrize("staging", ["0", "1"])
def test_db_writers_device_stream_and_part_files(dev, tmp_path, monkeypatch, staging):
    """fingerprints go from HBM straight into the (pinned) pages of the output memmap, or -- staging = 1 --
    through two pinned staging slots (small windows and slots here, so that window changes, both slots and the file
    growth are exercised); the shard-aware layout (one part file per rank, contiguous track ranges) reads back as the
    same rows through load_memmap_data and as per-rank slices."""
    from grafp_amd import fpdb
    monkeypatch.setattr(fpdb, "FORCE_STAGING", staging == "1")
    from grafp_amd.eval import PartedRows, load_memmap_data
    from grafp_amd.modules.transformations import GPUTransformNeuralfp
    cfg, model = _filled_model(dev)
    model.eval()
    aug = GPUTransformNeuralfp(cfg, None, None, train=False)
    tracks = [torch.from_numpy(0.1 * hash_normalish(f"dbp:{i}", (1, 16000 * 3 + 700 * i))) for i in range(5)]
    old_slot, old_grow = fpdb._MemmapAppender.__init__.__defaults__, fpdb._MemmapAppender.GROW_ROWS
    fpdb._MemmapAppender.__init__.__defaults__ = (16,)          # 16-row staging slots: several per track
    fpdb._MemmapAppender.GROW_ROWS = 64                          # the sparse file grows a few times
    try:
        fpdb.create_dummy_db(tracks, augment=aug, model=model, output_root_dir=str(tmp_path), fname="whole", verbose=False)
        for r in range(2):
            fpdb.create_dummy_db(tracks, augment=aug, model=model, output_root_dir=str(tmp_path), fname="parted",
                                 verbose=False, rank=r, world=2)
        fpdb.write_parts_manifest(str(tmp_path), "parted", 2)
    finally:
        fpdb._MemmapAppender.__init__.__defaults__ = old_slot
        fpdb._MemmapAppender.GROW_ROWS = old_grow
    whole, shape = load_memmap_data(str(tmp_path), "whole", display=False)
    parted, pshape = load_memmap_data(str(tmp_path), "parted", display=False)
    with torch.no_grad():
        want = torch.cat([model.embed(aug(t.to(dev), None)[0])[1] for t in tracks]).cpu().numpy()
    assert tuple(shape) == tuple(pshape) == want.shape and isinstance(parted, PartedRows)
    assert os.path.getsize(tmp_path / "whole.mm") == want.size * 4            # truncated to the rows written
    np.testing.assert_array_equal(np.asarray(whole), want)
    np.testing.assert_array_equal(np.asarray(parted), want)
    np.testing.assert_array_equal(parted[5:want.shape[0] - 3], want[5:-3])
    pick = np.array([0, want.shape[0] - 1, 7, 7, 30])
    np.testing.assert_array_equal(parted[pick], want[pick])
    lo, hi = fpdb.track_range(len(tracks), 0, 2)
    n0 = sum((1 + t.shape[1] // 512 - 32) // 3 + 1 for t in tracks[lo:hi])
    assert len(parted.part_rows(0)) == n0 and len(parted.part_rows(1)) == want.shape[0] - n0


def test_shortcut_gradient_fused_into_data_gradient(dev, monkeypatch):
    """The shortcut's gradient of every Grapher / FFN block rides the data-gradient GEMM of the block's first layer
    (ops.ShortcutToken): parameter gradients equal those of the plain autograd accumulation up to bf16 rounding."""
    from grafp_amd.train import Trainer, build_model, synthetic_batch
    from grafp_amd.util import load_config
    cfg = load_config()
    cfg["bsz_train"] = 8
    torch.manual_seed(3)
    model = build_model(cfg, device=dev)
    tr = Trainer(cfg, model, dev, amp_dtype=torch.bfloat16)
    x_i, x_j = synthetic_batch(8, 11, dev)
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def grads(fused):
        from grafp_amd import ops
        monkeypatch.setattr(ops.switches, "shortcut_fusion", bool(fused))
        model.load_state_dict(state)
        model.train()
        for p in model.parameters():
            p.grad = None
        with torch.no_grad():
            X_i, X_j = tr.augment(x_i, x_j)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            _, _, z_i, z_j = model(X_i, X_j)
        from grafp_amd.simclr.ntxent import ntxent_loss
        loss = ntxent_loss(z_i, z_j, cfg)
        loss.backward()
        return float(loss.detach()), {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}

    l1, g1 = grads(True)
    l0, g0 = grads(False)
    assert l1 == l0                                              # the forward pass is untouched
    assert g1.keys() == g0.keys()
    num = sum(float((g1[n] - g0[n]).pow(2).sum()) for n in g0)
    den = sum(float(g0[n].pow(2).sum()) for n in g0)
    # bf16 gradients, 24 residual blocks deep: the one rounding that changes per block input (2^-9 per element) is
    # amplified to ~2 % of the whole gradient (measured 0.024); a shortcut gradient that went missing or was counted
    # twice would show up at order 1
    assert den > 0 and (num / den) ** 0.5 <= 6e-2, (num / den) ** 0.5
    # per parameter, among those that carry gradient at all (the BatchNorm bias of a Grapher's fc1 has a mathematically
    # ZERO gradient -- max-relative differences and the next BatchNorm remove a per-channel constant -- so what is
    # computed for it is rounding noise, 1e-6 of the total norm, and differs by 100 % between any two roundings)
    floor = 1e-3 * den ** 0.5
    worst = max(float((g1[n] - g0[n]).norm() / g0[n].norm()) for n in g0 if float(g0[n].norm()) >= floor)
    assert worst <= 0.3, worst


def test_deferred_normalisation_equals_separate_passes(dev, monkeypatch):
    """Stages 0-1: the BatchNorm + ReLU of the grouped graph conv and of the FFN's hidden layer are applied by their
    consumer to its operand fragments (ops.DeferredNorm: no normalise pass, the normalised tensors never written).
    Same operand bits, same products: the loss, the embeddings and the running statistics are identical and the
    gradients differ only by the summation order of the consumer's weight gradient (another tile of the same split-K
    kernel).  The allocator is poisoned with NaNs first: a table entry nobody wrote would silently zero an operand row
    behind the ReLU."""
    from grafp_amd import ops
    from grafp_amd.simclr.ntxent import ntxent_loss
    from grafp_amd.train import Trainer, build_model, synthetic_batch
    from grafp_amd.util import load_config
    cfg = load_config()
    cfg["bsz_train"] = 8
    junk = torch.full((1 << 28,), float("nan"), device=dev)
    del junk
    torch.manual_seed(5)
    model = build_model(cfg, device=dev)
    tr = Trainer(cfg, model, dev, amp_dtype=torch.bfloat16)
    x_i, x_j = synthetic_batch(8, 13, dev)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    pro_launches = []
    orig = ops.conv1x1_gemm

    def gemm(w, x, groups=1, views=1, pro_tab=None, pro_act=0, pro_slope=0.0, stats=False):
        if pro_tab is not None:
            pro_launches.append((w.shape[0], x.shape[0]))
        return orig(w, x, groups, views, pro_tab, pro_act, pro_slope, stats)

    def run(defer):
        monkeypatch.setattr(ops.switches, "defer_norm", bool(defer))
        model.load_state_dict(state)
        model.train()
        for p in model.parameters():
            p.grad = None
        with torch.no_grad():
            X_i, X_j = tr.augment(x_i, x_j)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            _, _, z_i, z_j = model(X_i, X_j)
        loss = ntxent_loss(z_i, z_j, cfg)
        loss.backward()
        stats = {k: v.detach().float().clone() for k, v in model.state_dict().items() if "running_" in k}
        return (float(loss.detach()), torch.cat((z_i, z_j)).detach().float().clone(), stats,
                {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None})

    assert ops.switches.defer_norm is True                              # the shipping default
    monkeypatch.setattr(ops, "conv1x1_gemm", gemm)
    l1, z1, s1, g1 = run(True)
    n_pro = len(pro_launches)
    l0, z0, s0, g0 = run(False)
    assert n_pro == 8 and len(pro_launches) == 8, pro_launches        # gfc2 + ffn2 of the four stage 0-1 blocks, only
    assert sorted(set(pro_launches)) == [(64, 128), (64, 256), (128, 256), (128, 512)]
    assert torch.equal(z1, z0) and l1 == l0                            # the forward pass: bit for bit
    assert s1.keys() == s0.keys() and all(torch.equal(s1[k], s0[k]) for k in s0)
    assert g1.keys() == g0.keys()
    num = sum(float((g1[n] - g0[n]).pow(2).sum()) for n in g0)
    den = sum(float(g0[n].pow(2).sum()) for n in g0)
    assert den > 0 and (num / den) ** 0.5 <= 1e-5, (num / den) ** 0.5
