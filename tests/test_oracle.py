"""CPU tests: the oracle against the golden vectors produced from the reference (make_golden.py)."""
import numpy as np
import pytest
import torch

from _common import (filled_state_dict, golden, hash_ints, hash_normalish, hash_uniform,
                     knn_margin_mask, manifest_shapes, simclr_inputs)
from oracle import model as om
from oracle import native, retrieval

CFG = dict(fs=16000, n_fft=1024, hop_len=512, win_len=1024, n_mels=64, n_frames=32, overlap=0.9, tau=0.05)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# ---------------------------------------------------------------- peak extractor
def test_peak_extractor_matches_reference():
    g = golden("peak_extractor.npz")
    shapes = {"peak_extractor.convs.0.weight": (8, 3, 7, 7), "peak_extractor.convs.0.bias": (8,)}
    from _hashfill import fill_state_dict
    # the golden module was filled with prefix 'pe' on its own (un-prefixed) key names
    raw = fill_state_dict({"convs.0.weight": (8, 3, 7, 7), "convs.0.bias": (8,)}, "pe")
    sd = {"peak_extractor." + k: t(v) for k, v in raw.items()}
    assert set(sd) == set(shapes)
    spec = t(40.0 * hash_uniform("in:peak.spec", (2, 64, 32)) - 30.0)
    spec3 = t(40.0 * hash_uniform("in:peak.spec3", (3, 64, 32)) - 30.0)
    np.testing.assert_allclose(om.peak_extract(sd, spec).numpy(), g["out"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(om.peak_extract(sd, spec3).numpy(), g["out3"], rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------- k-NN graph
KNN_SHAPES = [(64, 1024), (128, 512), (256, 256), (512, 128), (24, 100)]


@pytest.mark.parametrize("C,N", KNN_SHAPES)
def test_knn_c_oracle_exact_on_integer_goldens(C, N):
    """Integer features: every product/sum is exact in f32, so the reference's result is unique up to
    genuine ties; on tie-free nodes the C oracle must equal the reference index-for-index."""
    g = golden("knn_graph.npz")
    xi = hash_ints(f"in:knn.int.{C}.{N}", (2, C, N, 1), -8, 8).astype(np.float32)
    ok, _ = knn_margin_mask(xi, 3, tol=0.5, normalize=False)
    assert ok.mean() > 0.9
    got = native.knn_graph(xi[..., 0], 3, normalize=False)
    assert np.array_equal(got[ok], g[f"int_{C}_{N}"][ok].astype(np.int64))


@pytest.mark.parametrize("C,N", KNN_SHAPES)
def test_knn_c_oracle_f32_goldens(C, N):
    """f32 features through normalise + knn: equal to the reference outside near-ties (gap > 1e-5)."""
    g = golden("knn_graph.npz")
    xf = hash_normalish(f"in:knn.f32.{C}.{N}", (2, C, N, 1))
    ok, _ = knn_margin_mask(xf, 3, tol=1e-5)
    assert ok.mean() > 0.98
    ref = g[f"f32_{C}_{N}"].astype(np.int64)
    got = native.knn_graph(xf[..., 0], 3)
    assert np.array_equal(got[ok], ref[ok])
    # self is always the first neighbour
    assert np.array_equal(got[..., 0][ok], np.broadcast_to(np.arange(N), (2, N))[ok])
    # the torch restatement reproduces the reference exactly (same ops, same host)
    tor = om.knn_graph_torch(t(xf[..., 0]), 3).numpy()
    assert np.array_equal(tor[ok], ref[ok])


def test_knn_k5_and_ties():
    g = golden("knn_graph.npz")
    xf = hash_normalish("in:knn.f32.k5", (2, 32, 200, 1))
    ok, _ = knn_margin_mask(xf, 5, tol=1e-5)
    got = native.knn_graph(xf[..., 0], 5)
    assert np.array_equal(got[ok], g["f32_k5"].astype(np.int64)[ok])
    # duplicated nodes: ties resolve to the lowest index
    x = np.zeros((1, 4, 6), dtype=np.float32)
    x[0, :, :] = np.array([[1, 1, 1, 0, 0, 0]] * 4, dtype=np.float32)
    x[0, 0, 3:] = 1.0
    idx = native.knn_graph(x, 3, normalize=False)
    assert idx[0, 0].tolist() == [0, 1, 2] and idx[0, 2].tolist() == [0, 1, 2]
    assert idx[0, 4].tolist() == [3, 4, 5]


# ---------------------------------------------------------------- gather / MRConv
def test_gather_and_max_relative():
    g = golden("mrconv.npz")
    B, C, N, K = 2, 8, 64, 3
    x = t(hash_normalish("in:mr.x", (B, C, N, 1)))[..., 0]
    idx = hash_ints("in:mr.idx", (B, N, K), 0, N - 1).astype(np.int64)
    idx[:, :, 0] = np.arange(N)[None, :]
    idx = t(idx)
    np.testing.assert_array_equal(om.gather_nodes(x, idx).numpy(), g["sel"])
    xr = x.clone().requires_grad_(True)
    inter = om.max_relative(xr, idx)
    np.testing.assert_array_equal(inter.detach().numpy(), g["inter"][..., 0])
    inter.backward(t(hash_normalish("in:mr.gi", (B, 2 * C, N, 1)))[..., 0])
    np.testing.assert_allclose(xr.grad.numpy(), g["dx_inter"][..., 0], rtol=1e-6, atol=1e-6)


def test_mrconv_full():
    g = golden("mrconv.npz")
    B, C, N, K = 2, 8, 64, 3
    shapes = {"nn.0.weight": (16, 4, 1, 1), "nn.0.bias": (16,), "nn.1.weight": (16,), "nn.1.bias": (16,),
              "nn.1.running_mean": (16,), "nn.1.running_var": (16,), "nn.1.num_batches_tracked": ()}
    sd = filled_state_dict(shapes, "mr", requires_grad=True)
    x = t(hash_normalish("in:mr.x", (B, C, N, 1))).requires_grad_(True)
    idx = hash_ints("in:mr.idx", (B, N, K), 0, N - 1).astype(np.int64)
    idx[:, :, 0] = np.arange(N)[None, :]
    m = om.max_relative(x[..., 0], t(idx)).unsqueeze(-1)
    y = torch.relu(om._bn(sd, "nn.1", om._conv1x1(sd, "nn.0", m, groups=4), True))
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=1e-5, atol=1e-5)
    y.backward(t(hash_normalish("in:mr.gy", tuple(y.shape))))
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(sd["nn.0.weight"].grad.numpy(), g["dw"], rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------- Grapher + FFN block
def test_block_train_eval():
    g = golden("block.npz")
    C, N = 16, 64
    keys = [str(k) for k in g["keys"]]
    ref_shapes = {
        "fc1.0.weight": (C, C, 1, 1), "fc1.0.bias": (C,), "fc2.0.weight": (C, 2 * C, 1, 1), "fc2.0.bias": (C,),
        "graph_conv.gconv.nn.0.weight": (2 * C, C // 2, 1, 1), "graph_conv.gconv.nn.0.bias": (2 * C,)}
    shapes = {}
    for k in keys:
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "relative_pos":
            continue
        body = k.split(".", 1)[1]
        if k.startswith("0.") and body in ref_shapes:
            shapes[k] = ref_shapes[body]
        elif k in ("1.fc1.0.weight",):
            shapes[k] = (4 * C, C, 1, 1)
        elif k in ("1.fc2.0.weight",):
            shapes[k] = (C, 4 * C, 1, 1)
        elif leaf == "num_batches_tracked":
            shapes[k] = ()
        else:                                             # BN vectors
            width = {"0.fc1.1": C, "0.fc2.1": C, "0.graph_conv.gconv.nn.1": 2 * C,
                     "1.fc1.1": 4 * C, "1.fc2.1": C}[k.rsplit(".", 1)[0]]
            shapes[k] = (width,)
    sd = filled_state_dict(shapes, "blk")
    xb = t(hash_normalish("in:blk.x", (3, C, N, 1)))
    y = om.ffn(sd, "1.", om.grapher(sd, "0.", xb, 3, True), True)
    np.testing.assert_allclose(y.numpy(), g["y_train"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(sd["0.fc1.1.running_mean"].numpy(), g["fc1_running_mean"], rtol=1e-5, atol=1e-6)
    y = om.ffn(sd, "1.", om.grapher(sd, "0.", xb, 3, False), False)
    np.testing.assert_allclose(y.numpy(), g["y_eval"], rtol=2e-5, atol=2e-5)


# ---------------------------------------------------------------- full model
def test_manifest_schema():
    shapes = manifest_shapes()
    assert len(shapes) == 443
    assert shapes["encoder.backbone.0.0.graph_conv.gconv.nn.0.weight"] == (128, 32, 1, 1)
    assert shapes["encoder.backbone.12.conv.0.weight"] == (512, 256, 3, 3)
    assert shapes["encoder.backbone.0.0.relative_pos"] == (1, 1024, 1024)
    assert shapes["encoder.backbone.13.0.relative_pos"] == (1, 16, 16)
    n_train = sum(int(np.prod(s)) for k, s in shapes.items()
                  if k.rsplit(".", 1)[-1] in ("weight", "bias"))
    assert n_train == 18367264                                   # SURVEY.md section 2a [probe]


def test_simclr_forward_train_and_eval():
    g = golden("simclr_forward.npz")
    sd = filled_state_dict()
    xi, xj = simclr_inputs()
    with torch.no_grad():
        h_i, h_j, z_i, z_j = om.simclr_forward(sd, xi, xj, True)
    np.testing.assert_allclose(h_i.numpy(), g["h_i"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(z_i.numpy(), g["z_i"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(z_j.numpy(), g["z_j"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(sd["encoder.stem.1.running_mean"].numpy(), g["stem_running_mean"],
                               rtol=1e-5, atol=1e-6)
    with torch.no_grad():
        eh_i, _, ez_i, ez_j = om.simclr_forward(sd, xi, xj, False)
    np.testing.assert_allclose(ez_i.numpy(), g["eval_z_i"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(ez_j.numpy(), g["eval_z_j"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(eh_i.numpy(), g["eval_h_i"], rtol=1e-3, atol=1e-4)


def test_simclr_forward_with_c_knn_graph():
    """Swapping torch's k-NN for the fully specified C oracle does not change the embeddings beyond
    f32 noise on this input (no near-tie flips): ties the two k-NN statements together end to end."""
    g = golden("simclr_forward.npz")
    sd = filled_state_dict()
    xi, xj = simclr_inputs()

    def c_knn(x, k):
        return torch.from_numpy(native.knn_graph(x.detach().numpy(), k))
    with torch.no_grad():
        _, _, z_i, z_j = om.simclr_forward(sd, xi, xj, True, idx_fn=c_knn)
    np.testing.assert_allclose(z_i.numpy(), g["z_i"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(z_j.numpy(), g["z_j"], rtol=1e-3, atol=1e-5)


# ---------------------------------------------------------------- NT-Xent
@pytest.mark.parametrize("B", [2, 8, 32])
def test_ntxent_value_and_grads(B):
    g = golden("ntxent.npz")
    for fn in (om.ntxent_loop, om.ntxent):
        a = t(g[f"zi_{B}"]).clone().requires_grad_(True)
        b = t(g[f"zj_{B}"]).clone().requires_grad_(True)
        loss = fn(a, b, 0.05)
        loss.backward()
        np.testing.assert_allclose(loss.item(), g[f"loss_{B}"], rtol=2e-6)
        np.testing.assert_allclose(a.grad.numpy(), g[f"dzi_{B}"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(b.grad.numpy(), g[f"dzj_{B}"], rtol=1e-4, atol=1e-6)


def test_ntxent_unnormalised_inputs():
    g = golden("ntxent.npz")
    a = t(0.3 * hash_normalish("in:nt.raw.zi", (6, 16))).requires_grad_(True)
    b = t(0.3 * hash_normalish("in:nt.raw.zj", (6, 16))).requires_grad_(True)
    loss = om.ntxent(a, b, 0.5)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["loss_raw"], rtol=2e-6)
    np.testing.assert_allclose(a.grad.numpy(), g["dzi_raw"], rtol=1e-4, atol=1e-7)


# ---------------------------------------------------------------- one train step
def test_train_step():
    g = golden("train_step.npz")
    sd = filled_state_dict(requires_grad=True)
    params = om.trainable(sd)
    assert sum(p.numel() for p in params.values()) == 18367264
    opt = torch.optim.Adam(list(params.values()), lr=8e-5)
    xi, xj = simclr_inputs()
    # keep grads for inspection: run the step manually
    opt.zero_grad()
    _, _, z_i, z_j = om.simclr_forward(sd, xi, xj, True)
    loss = om.ntxent(z_i, z_j, 0.05)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    probe = [str(p) for p in g["probe"]]
    gn = np.array([sd[k].grad.double().norm().item() for k in probe])
    np.testing.assert_allclose(gn, g["grad_norm"], rtol=5e-3)
    for k in probe[:2]:
        np.testing.assert_allclose(sd[k].grad.numpy(), g["grad:" + k], rtol=5e-3, atol=1e-6)
    opt.step()
    ps = np.array([sd[k].detach().double().sum().item() for k in probe])
    np.testing.assert_allclose(ps, g["param_sum_after"], rtol=1e-5, atol=1e-3)


# ---------------------------------------------------------------- log-mel (parity unpinned vs torchaudio)
def _logmel_numpy(x, n_fft=1024, hop=512, n_mels=64, fs=16000):
    """Independent float64 numpy implementation of the published definition."""
    x = np.asarray(x, dtype=np.float64)
    pad = n_fft // 2
    xp = np.pad(x, (pad, pad), mode="reflect")
    n_frames = 1 + len(x) // hop
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)          # periodic hann
    frames = np.stack([xp[i * hop:i * hop + n_fft] * win for i in range(n_frames)])
    power = np.abs(np.fft.rfft(frames, axis=1)) ** 2                         # (frames, 513)
    freqs = np.linspace(0, fs // 2, n_fft // 2 + 1)
    mmax = 2595.0 * np.log10(1.0 + (fs / 2) / 700.0)
    fpts = 700.0 * (10.0 ** (np.linspace(0, mmax, n_mels + 2) / 2595.0) - 1.0)
    fb = np.zeros((n_fft // 2 + 1, n_mels))
    for m in range(n_mels):
        lo, ce, hi = fpts[m], fpts[m + 1], fpts[m + 2]
        fb[:, m] = np.maximum(0, np.minimum((freqs - lo) / (ce - lo), (hi - freqs) / (hi - ce)))
    return 10.0 * np.log10(np.maximum(power @ fb, 1e-10)).T                  # (n_mels, frames)


def test_logmel_against_independent_numpy():
    x = 0.1 * hash_normalish("in:logmel.x", (3, 16000))
    got = om.logmel(t(x), CFG).numpy()
    assert got.shape == (3, 64, 32)
    for b in range(3):
        np.testing.assert_allclose(got[b], _logmel_numpy(x[b]), rtol=0, atol=2e-3)   # dB
    fb = om.mel_filterbank().numpy()
    assert fb.shape == (513, 64) and int((fb > 0).sum()) in (1000, 1001)     # sparse: ~3-42 bins per band


def test_val_segments_shape():
    x = 0.1 * hash_normalish("in:logmel.track", (1, 16000 * 5))
    seg = om.val_segments(t(x), CFG)
    frames = 1 + 80000 // 512
    assert seg.shape == ((frames - 32) // 3 + 1, 64, 32)
    full = om.logmel(t(x[0]), CFG)
    np.testing.assert_array_equal(seg[2].numpy(), full[:, 6:38].numpy())


# ---------------------------------------------------------------- flat search + eval (parity unpinned vs faiss)
def test_flat_search_against_f64_and_ties():
    db = hash_normalish("in:fs.db", (3000, 128)); db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = db[100:141] + 0.05 * hash_normalish("in:fs.q", (41, 128)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    d, i = native.flat_search_l2(db, q, 20)
    d64, i64 = retrieval.exact_search_f64(db, q, 20)
    assert np.array_equal(i[:, 0], np.arange(100, 141))
    gap_ok = np.diff(d64, axis=1).min(axis=1) > 1e-5
    assert gap_ok.mean() > 0.9
    assert np.array_equal(i[gap_ok], i64[gap_ok])
    np.testing.assert_allclose(d, d64, atol=5e-6)
    # duplicates: lowest id first; k > n: padded with -1 / inf
    dup = np.tile(db[:3], (4, 1)).astype(np.float32)
    d, i = native.flat_search_l2(dup, db[:1], 5)
    assert i[0].tolist()[:4] == [0, 3, 6, 9]
    d, i = native.flat_search_l2(db[:3], db[:2], 5)
    assert i[0, 3:].tolist() == [-1, -1] and np.isinf(d[0, 3:]).all()
    # sharded search + merge == unsharded
    parts = [native.flat_search_l2(db[s:s + 1000], q, 20, id_base=s) for s in (0, 1000, 2000)]
    md, mi = native.merge_topk(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]))
    d, i = native.flat_search_l2(db, q, 20)
    assert np.array_equal(mi, i) and np.array_equal(md, d)


def test_eval_l2_planted_sequences():
    dummy = hash_normalish("in:ev.dummy", (500, 128)); dummy /= np.linalg.norm(dummy, axis=1, keepdims=True)
    db = hash_normalish("in:ev.db", (200, 128)); db /= np.linalg.norm(db, axis=1, keepdims=True)
    query = db + 0.04 * hash_normalish("in:ev.noise", (200, 128)); query /= np.linalg.norm(query, axis=1, keepdims=True)
    rates, raw, top1 = retrieval.eval_l2(query, db, dummy, np.arange(0, 150, 10), [1, 3, 5], k_probe=20)
    assert rates.shape == (4, 3) and raw.shape == (15, 12)
    assert (rates[0] == 100.0).all() and (top1[:, 0] == np.arange(0, 150, 10) + 500).all()


def test_eval_faiss_golden_from_reference():
    """The reference's own eval_faiss (eval.py:170-332, run by tests/golden/make_eval_golden.py with a stand-in exact
    search for the absent faiss) on the hash-filled case: the numpy restatement and the C rerank restatement must
    reproduce its hit-rate table and raw hit flags exactly."""
    from _common import eval_case
    g = golden("eval_faiss.npz")
    case = eval_case()
    lens = [int(v) for v in case["test_seq_len"].split()]
    assert np.array_equal(g["test_ids"], case["test_ids"])
    rates, raw, top1 = retrieval.eval_l2(case["query"], case["db"], case["dummy_db"], case["test_ids"], lens,
                                         k_probe=case["k_probe"])
    np.testing.assert_array_equal(rates, g["hit_rates"])
    np.testing.assert_array_equal(raw, g["raw_score"])
    assert 40.0 < g["hit_rates"][0, 0] < 60.0 and g["hit_rates"][0, -1] == 100.0      # the case is informative
    # the C restatement (fixed f32 order, batched items) gives the same flags through the same pipeline
    index_rows = np.concatenate([case["dummy_db"], case["db"]], axis=0)
    _, I = native.flat_search_l2(index_rows, case["query"], case["k_probe"])
    item_row = np.repeat(case["test_ids"], len(lens))
    item_len = np.tile(np.asarray(lens, dtype=np.int32), len(case["test_ids"]))
    pred, scores = native.seq_rerank(index_rows, case["query"], I, item_row, item_len, top=10)
    pred = pred.reshape(len(case["test_ids"]), len(lens), 10)
    gt = (case["test_ids"] + len(case["dummy_db"]))[:, None]
    flags = np.stack([pred[:, :, 0] == gt, np.abs(pred[:, :, 0] - gt) <= 1, (pred[:, :, :3] == gt[:, :, None]).any(2),
                      (pred == gt[:, :, None]).any(2)]).astype(int)
    np.testing.assert_array_equal(np.concatenate(list(flags), axis=1), g["raw_score"])
    assert np.array_equal(pred[:, :, 0], top1)
    assert (np.diff(scores, axis=1) <= 0).all()                                        # best first


def test_seq_rerank_c_vs_numpy_scores():
    """C restatement vs the float64 numpy scores of retrieval.sequence_scores on ragged items (candidates that run
    past the end of the index, ids < 0, duplicate candidates)."""
    index_rows = hash_normalish("rr:index", (300, 128)).astype(np.float32)
    q = hash_normalish("rr:q", (40, 128)).astype(np.float32)
    ids = hash_ints("rr:ids", (40, 6), 0, 299).astype(np.int64)
    ids[3, 2] = -1
    ids[5, :] = 298                                                   # sequences that leave the index
    ids[6, :3] = ids[5, 0] + 1                                        # same start id after offset compensation
    item_row = np.array([0, 4, 4, 20, 39], dtype=np.int64)
    item_len = np.array([3, 5, 1, 19, 1], dtype=np.int32)
    pred, scores = native.seq_rerank(index_rows, q, ids, item_row, item_len, top=10)
    for it in range(len(item_row)):
        r0, ql = int(item_row[it]), int(item_len[it])
        I = ids[r0:r0 + ql].copy() - np.arange(ql)[:, None]
        I[ids[r0:r0 + ql] < 0] = -1
        cand = np.unique(I[I >= 0])
        sc = retrieval.sequence_scores(q[r0:r0 + ql], index_rows, cand, ql)
        order = np.argsort(-sc, kind="stable")[:10]
        n = min(10, len(cand))
        assert np.array_equal(pred[it, :n], cand[order][:n]) and (pred[it, n:] == -1).all()
        np.testing.assert_allclose(scores[it, :n], sc[order][:n], rtol=2e-5, atol=2e-6)


def test_augment_oracle_matches_definitions():
    """oracle/csrc/augment.c against float64 numpy statements of the two transforms' published definitions
    (torch_audiomentations is not installable here: parity unpinned against the library itself): full convolution
    truncated to the input length; background scaled to the requested SNR after RMS normalisation."""
    from oracle import native as on
    rng = np.random.default_rng(3)
    B, T = 4, 3000
    x = rng.standard_normal((B, T)).astype(np.float32)
    bank = np.zeros((2, 900), np.float32)
    bank[0, :900] = rng.standard_normal(900) * np.exp(-np.arange(900) / 150.0)
    bank[1, :17] = rng.standard_normal(17)
    lens, idx = np.array([900, 17]), np.array([0, -1, 1, 0])
    y = on.ir_convolve(x, bank, lens, idx)
    for b, i in enumerate(idx):
        want = x[b] if i < 0 else np.convolve(x[b].astype(np.float64), bank[i, :lens[i]].astype(np.float64))[:T]
        np.testing.assert_allclose(y[b], want, rtol=0, atol=2e-5 * np.abs(want).max())
    assert np.array_equal(on.ir_convolve(x, bank, lens, None)[1], on.ir_convolve(x[1:2], bank, lens, np.array([0]))[0])
    noise = rng.standard_normal((2, 2000)).astype(np.float32) * 3.0
    nlen, nidx, off = np.array([2000, 777]), np.array([1, 0, -1, 1]), np.array([5, 1999, 0, 776])
    snr = np.array([10.0, 0.0, 5.0, 20.0], np.float32)
    z = on.mix_snr(x, noise, nlen, nidx, off, snr)
    for b in range(B):
        if nidx[b] < 0:
            assert np.array_equal(z[b], x[b]); continue
        n = noise[nidx[b], (off[b] + np.arange(T)) % nlen[nidx[b]]].astype(np.float64)
        added = z[b].astype(np.float64) - x[b]
        got_snr = 20 * np.log10(np.sqrt((x[b].astype(np.float64) ** 2).mean()) / np.sqrt((added ** 2).mean()))
        assert abs(got_snr - snr[b]) < 1e-3
        assert np.corrcoef(added, n)[0, 1] > 0.999999


def test_ivfpq_restatement_is_the_distance_to_the_dequantised_vector():
    """Pins oracle/ivfpq.py to the published definition: the asymmetric distance of a code equals the exact squared
    distance between the query and the vector the code reconstructs to; codes are the nearest codewords of the residual."""
    from oracle import ivfpq
    rng = np.random.RandomState(0)
    d, M, nlist = 16, 8, 5
    x = rng.randn(300, d).astype(np.float32)
    cent = x[rng.permutation(300)[:nlist]].copy()
    books = rng.randn(M, 256, d // M).astype(np.float32) * 0.5
    a, codes = ivfpq.encode(x, cent, books)
    assert a.shape == (300,) and codes.shape == (300, M) and codes.dtype == np.uint8
    rec = ivfpq.reconstruct(a, codes, cent, books)
    # no other codeword of any sub-space is closer to the residual than the chosen one
    res = (x - cent[a]).reshape(300, M, d // M)
    for m in (0, 3, 7):
        dall = ((res[:, m, None, :] - books[m][None]) ** 2).sum(-1)
        assert (dall.min(axis=1) >= ((res[:, m] - books[m][codes[:, m]]) ** 2).sum(-1) - 1e-9).all()
    q = rng.randn(7, d).astype(np.float32)
    D, I = ivfpq.search(q, a, codes, cent, books, nprobe=nlist, k=10)               # all lists probed: exhaustive
    exact = ((q[:, None, :].astype(np.float64) - rec[None]) ** 2).sum(-1)
    order = np.argsort(exact, axis=1, kind="stable")[:, :10]
    np.testing.assert_allclose(D, np.take_along_axis(exact, order, axis=1), rtol=1e-9, atol=1e-9)
    assert (I == order).mean() > 0.98                                               # ties aside
    D2, I2 = ivfpq.search(q, a, codes, cent, books, nprobe=2, k=10)                   # fewer lists: a subset, never better
    assert (D2[:, 0] >= D[:, 0] - 1e-12).all() and ((I2 >= 0).sum(1) <= 10).all()


def test_ivfpq_c_restatement_agrees_with_the_float64_definition():
    """oracle/csrc/ivfpq.c (f32, the accumulation orders the HIP kernels follow) against oracle/ivfpq.py (float64, the
    published definition): identical list ids and codes away from near-ties, identical result ids, distances within f32
    rounding; the seeded Lloyd iterations reduce the quantisation error monotonically."""
    from oracle import ivfpq, native
    rng = np.random.RandomState(0)
    x = rng.randn(3000, 16).astype(np.float32)
    init = np.arange(8) * 7
    errs = []
    for niter in (0, 1, 4):
        cent = native.kmeans(x, 1, 8, init, niter)[0]
        a = native.pq_assign(x, 1, cent[None])[:, 0]
        errs.append(float(((x - cent[a]) ** 2).sum()))
    assert errs[0] > errs[1] > errs[2]
    assert np.array_equal(native.kmeans(x, 1, 8, init, 0)[0], x[init])                  # no iteration: the seeds
    assert (a == ivfpq.assign(x, cent)).mean() > 0.999
    books = native.kmeans(x, 8, 256, np.arange(256) * 3, 3, base=cent, base_idx=a)
    codes = native.pq_assign(x, 8, books, base=cent, base_idx=a).astype(np.uint8)
    a64, codes64 = ivfpq.encode(x, cent, books)
    assert (codes == codes64).mean() > 0.999
    order = np.argsort(a, kind="stable")
    start = np.r_[0, np.cumsum(np.bincount(a, minlength=8))]
    q = x[:40] + 0.01
    probe = native.ivfpq_probe(q, cent, 3)
    D, I = native.ivfpq_search(q, cent, books, codes[order], start, order, probe, 10)
    D64, I64 = ivfpq.search(q, a, codes, cent, books, 3, 10)
    assert (I == I64).mean() > 0.99
    np.testing.assert_allclose(D, D64, rtol=1e-5, atol=1e-5)
    # an empty cluster keeps its centroid: two identical seeds, the second loses every tie of the first assignment
    init2 = np.array([0, 0, 5, 9, 11, 13, 17, 19])
    c2 = native.kmeans(x, 1, 8, init2, 1)[0]
    assert np.array_equal(c2[1], x[0]) and not np.array_equal(c2[0], x[0])
