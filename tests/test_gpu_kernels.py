"""GPU parity tests: every HIP kernel, called through the C ABI (grafp_amd.ops -> libgrafp_hip.so),
against the oracle on the same seeded inputs.  Bit-exact for index-valued results; stated tolerances for
floating point.  Run with `pytest -m gpu` on an MI355X."""
import ctypes

import numpy as np
import pytest
import torch

from _common import golden, hash_ints, hash_normalish, hash_uniform, knn_margin_mask

pytestmark = pytest.mark.gpu

CFG = dict(fs=16000, n_fft=1024, hop_len=512, win_len=1024, n_mels=64, n_frames=32, overlap=0.9, tau=0.05)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need a HIP device"
    return torch.device("cuda:0")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# =============================================================== k-NN graph (bit-exact)
@pytest.mark.parametrize("C,N,B,k", [(64, 1024, 3, 3), (128, 512, 2, 3), (256, 256, 2, 3), (512, 128, 3, 3),
                                      (24, 100, 2, 3), (3, 101, 2, 2), (40, 37, 1, 5), (8, 8, 2, 8), (16, 300, 2, 1)])
def test_knn_graph_bit_exact_vs_c_oracle(dev, C, N, B, k):
    from grafp_amd import ops
    from oracle import native
    x = hash_normalish(f"gpu:knn.{C}.{N}", (B, C, N))
    want = native.knn_graph(x, k, normalize=True)
    got = ops.knn_graph(t(x).to(dev), k, normalize=True).cpu().numpy()
    assert got.dtype == np.int64 and got.shape == (B, N, k)
    mism = np.argwhere(got != want)
    assert len(mism) == 0, f"{len(mism)} of {got.size} indices differ; first {mism[:5].tolist()}"
    # un-normalised entry (dense_knn_matrix) on integer features, including genuine ties
    xi = hash_ints(f"gpu:knn.int.{C}.{N}", (B, C, N), -3, 3).astype(np.float32)
    want = native.knn_graph(xi, k, normalize=False)
    got = ops.knn_graph(t(xi).to(dev), k, normalize=False).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("C,N", [(64, 1024), (128, 512), (256, 256), (512, 128), (24, 100)])
def test_knn_graph_matches_reference_goldens(dev, C, N):
    """Against what the reference itself produced (tests/golden/knn_graph.npz), outside near-ties."""
    from grafp_amd import ops
    g = golden("knn_graph.npz")
    xf = hash_normalish(f"in:knn.f32.{C}.{N}", (2, C, N, 1))
    ok, _ = knn_margin_mask(xf, 3, tol=1e-5)
    got = ops.knn_graph(t(xf).to(dev), 3).cpu().numpy()
    assert np.array_equal(got[ok], g[f"f32_{C}_{N}"].astype(np.int64)[ok])
    xi = hash_ints(f"in:knn.int.{C}.{N}", (2, C, N, 1), -8, 8).astype(np.float32)
    ok, _ = knn_margin_mask(xi, 3, tol=0.5, normalize=False)
    got = ops.knn_graph(t(xi).to(dev), 3, normalize=False).cpu().numpy()
    assert np.array_equal(got[ok], g[f"int_{C}_{N}"].astype(np.int64)[ok])


@pytest.mark.parametrize("C,N,B", [(64, 1024, 2), (128, 512, 2), (256, 256, 3), (512, 128, 2), (64, 128, 5)])
def test_knn_graph_prefilter_equals_f32_path(dev, C, N, B):
    """knn_pre.hip (bf16 pre-filter + exact rescoring) against knn_graph.hip (all exact f32) and the C oracle: normal
    features, clusters of near-duplicates (many survivors -> the exact fallback of a lane), exact duplicates (ties ->
    lowest index) and un-normalised integer features (loose bounds -> fallback everywhere)."""
    from grafp_amd import ops
    from oracle import native
    x = hash_normalish(f"gpu:knnpre.{C}.{N}", (B, C, N)).astype(np.float32)
    x[0, :, 40:100] = x[0, :, 40:41] + 1e-3 * hash_normalish(f"gpu:knnpre.c.{C}.{N}", (C, 60))   # a tight cluster
    x[1, :, 10:50] = x[1, :, 10:11]                                                              # exact duplicates
    for normalize, xx in ((True, x), (False, hash_ints(f"gpu:knnpre.i.{C}.{N}", (B, C, N), -4, 4).astype(np.float32))):
        xt = t(xx).to(dev)
        a = ops.knn_graph(xt, 3, normalize=normalize, prefilter=False)
        b = ops.knn_graph(xt, 3, normalize=normalize, prefilter=True)
        assert torch.equal(a, b)
        assert np.array_equal(b.cpu().numpy(), native.knn_graph(xx, 3, normalize=normalize))
        b32 = ops.knn_graph(xt, 3, normalize=normalize, index_dtype=torch.int32)
        assert torch.equal(b32.to(torch.int64), b)
    xc = t(x).to(dev).permute(1, 0, 2).contiguous()                      # (C, B, N) layout, bf16 activations
    for dt in (torch.float32, torch.bfloat16):
        a = ops.knn_graph(xc.to(dt), 3, layout="cbn", prefilter=False)
        b = ops.knn_graph(xc.to(dt), 3, layout="cbn", prefilter=True)
        assert torch.equal(a, b)
    for k in (1, 2, 4):
        xt = t(x).to(dev)
        assert torch.equal(ops.knn_graph(xt, k, prefilter=False), ops.knn_graph(xt, k, prefilter=True))


@pytest.mark.parametrize("C,N,B", [(64, 1024, 3), (128, 512, 2), (256, 256, 3), (512, 128, 2), (64, 128, 5), (32, 256, 2)])
def test_knn_graph_split_equals_f32_path(dev, C, N, B):
    """knn_split.hip (split-bf16 Gram matrix, certified order, exact recomputation of the uncertified queries) -- the
    DEFAULT path of ops.knn_graph -- against knn_graph.hip (all exact f32) and the C oracle: normal features, a tight
    cluster (near-ties: uncertified), exact duplicates (ties -> lowest index), k = 1 ... 4, both layouts, bf16 input,
    int32 / int64 indices.  Random features certify almost every query; duplicates must go through the exact path."""
    from grafp_amd import ops
    from oracle import native
    assert ops.switches.knn_split
    x = hash_normalish(f"gpu:knnsplit.{C}.{N}", (B, C, N)).astype(np.float32)
    x[0, :, 40:100] = x[0, :, 40:41] + 1e-3 * hash_normalish(f"gpu:knnsplit.c.{C}.{N}", (C, 60))   # a tight cluster
    x[1, :, 10:50] = x[1, :, 10:11]                                                                # exact duplicates
    xt = t(x).to(dev)
    for k in (3, 1, 2, 4):
        a = ops.knn_graph(xt, k, prefilter=False)                     # exact-f32 MFMA kernel
        b, unc = ops.knn_graph_split(xt, k, return_uncertified=True)
        assert torch.equal(a, b), (k, int((a != b).sum()))
        assert torch.equal(ops.knn_graph(xt, k), a)                  # the default route (split path where preferred)
        if k == 3:
            assert np.array_equal(b.cpu().numpy(), native.knn_graph(x, 3))
            # every duplicate sees >= 39 candidates at distance exactly 0: never certified
            assert 40 <= int(unc) <= 100 + 0.1 * B * N, int(unc)     # the 60-node cluster and the 40 duplicates at most
    b32 = ops.knn_graph(xt, 3, index_dtype=torch.int32)
    assert b32.dtype == torch.int32 and torch.equal(b32.to(torch.int64), ops.knn_graph(xt, 3, prefilter=False))
    xc = xt.permute(1, 0, 2).contiguous()                               # (C, B, N) layout, bf16 activations
    for dt in (torch.float32, torch.bfloat16):
        assert torch.equal(ops.knn_graph(xc.to(dt), 3, layout="cbn", prefilter=False),
                           ops.knn_graph(xc.to(dt), 3, layout="cbn"))
    # plain random features: almost everything certifies (the exact path is the exception, not the rule)
    y = t(hash_normalish(f"gpu:knnsplit.r.{C}.{N}", (B, C, N)).astype(np.float32)).to(dev)
    idx, unc = ops.knn_graph_split(y, 3, return_uncertified=True)
    assert torch.equal(idx, ops.knn_graph(y, 3, prefilter=False))
    assert int(unc) <= (0.05 if C < 512 else 0.12) * B * N, (int(unc), B * N)      # the bound grows with C (7.6 % at C = 512)


def test_knn_graph_split_worst_cases(dev):
    """All nodes identical (every distance ties: the whole clip takes the exact path, lowest indices win), one-hot
    features (exact zeros / exact ties at distance 2) and features of wildly different scale before normalisation."""
    from grafp_amd import ops
    from oracle import native
    B, C, N = 2, 64, 256
    x = np.ones((B, C, N), dtype=np.float32)
    x[1] = np.eye(C, dtype=np.float32)[:, np.arange(N) % C]
    xt = t(x).to(dev)
    idx, unc = ops.knn_graph_split(xt, 3, return_uncertified=True)
    assert int(unc) == B * N
    assert np.array_equal(idx.cpu().numpy(), native.knn_graph(x, 3))
    z = hash_normalish("gpu:knnsplit.scale", (B, C, N)).astype(np.float32) * np.logspace(-20, 15, N, dtype=np.float32)[None, None, :]
    assert np.array_equal(ops.knn_graph(t(z).to(dev), 3).cpu().numpy(), native.knn_graph(z, 3))


@pytest.mark.parametrize("C,N,B", [(64, 1024, 3), (128, 512, 3), (256, 256, 4), (512, 128, 4)])
def test_knn_graph_raw_bf16_equals_f32_path(dev, C, N, B):
    """bf16 inputs (the training path's activations) take the RAW form of knn_split.hip: the Gram matrix of the
    un-normalised bf16 features (exact operands: one MFMA per 16 channels, no planes), normalisation behind the product,
    certified tiers as before.  Against knn_graph.hip on the same bf16 values and the C oracle on their f32 widening:
    normal features, a tight cluster, exact duplicates, a zero node, k = 1 ... 4, both layouts, both index types."""
    from grafp_amd import ops
    from oracle import native
    x = hash_normalish(f"gpu:knnraw.{C}.{N}", (B, C, N)).astype(np.float32)
    x[0, :, 40:100] = x[0, :, 40:41] + 1e-2 * hash_normalish(f"gpu:knnraw.c.{C}.{N}", (C, 60))    # a tight cluster
    x[1, :, 10:50] = x[1, :, 10:11]                                                               # exact duplicates
    x[2, :, 7] = 0.0                                                                              # a zero node
    xb = t(x).to(torch.bfloat16).to(dev)
    xw = xb.float().cpu().numpy()                                      # what the oracle sees: the same bf16 values
    # the default route takes it up to C = 256 (at C = 512 the exact-f32 kernel ties at 2048 clips and wins below)
    assert ops.lib.grafp_knn_split_preferred_for(ops._DT[torch.bfloat16], C, N, 3) == (1 if C <= 256 else 0)
    for k in (3, 1, 2, 4):
        a = ops.knn_graph(xb, k, prefilter=False)                      # exact-f32 MFMA kernel
        b, unc = ops.knn_graph_split(xb, k, return_uncertified=True)
        assert torch.equal(a, b), (k, int((a != b).sum()))
        assert torch.equal(ops.knn_graph(xb, k), a)                    # the default route
        if k == 3:
            assert np.array_equal(b.cpu().numpy(), native.knn_graph(xw, 3))
            assert 40 <= int(unc) <= 100 + 0.05 * B * N, int(unc)
    b32 = ops.knn_graph(xb, 3, index_dtype=torch.int32)
    assert b32.dtype == torch.int32 and torch.equal(b32.to(torch.int64), ops.knn_graph(xb, 3, prefilter=False))
    xc = xb.permute(1, 0, 2).contiguous()                              # (C, B, N): how the encoder hands them over
    assert torch.equal(ops.knn_graph(xc, 3, layout="cbn"), ops.knn_graph(xb, 3, prefilter=False))
    # plain random features: the bound is 3 x tighter than the split form's -- almost everything certifies
    y = t(hash_normalish(f"gpu:knnraw.r.{C}.{N}", (B, C, N)).astype(np.float32)).to(torch.bfloat16).to(dev)
    idx, unc = ops.knn_graph_split(y, 3, return_uncertified=True)
    assert torch.equal(idx, ops.knn_graph(y, 3, prefilter=False))
    assert np.array_equal(idx.cpu().numpy(), native.knn_graph(y.float().cpu().numpy(), 3))
    assert int(unc) <= (0.02 if C < 512 else 0.05) * B * N, (int(unc), B * N)


def test_knn_graph_raw_bf16_worst_cases(dev):
    """bf16 inputs: all nodes identical, one-hot features, an all-zero clip, features spread over 50 decades (norms above
    2^40 send the whole clip through the exact pass: the raw Gram entries could overflow) and 2^-100-sized features."""
    from grafp_amd import ops
    from oracle import native
    B, C, N = 5, 64, 256
    x = np.ones((B, C, N), dtype=np.float32)
    x[1] = np.eye(C, dtype=np.float32)[:, np.arange(N) % C]
    x[2] = 0.0
    x[3] = hash_normalish("gpu:knnraw.scale", (C, N)).astype(np.float32) * np.logspace(-20, 30, N, dtype=np.float32)[None, :]
    x[4] = hash_normalish("gpu:knnraw.tiny", (C, N)).astype(np.float32) * np.float32(2.0 ** -100)
    xb = t(x).to(torch.bfloat16).to(dev)
    xw = xb.float().cpu().numpy()
    idx, unc = ops.knn_graph_split(xb, 3, return_uncertified=True)
    assert np.array_equal(idx.cpu().numpy(), native.knn_graph(xw, 3))
    assert int(unc) >= 4 * N                                            # clips 0-3 whole; the tiny clip may certify
    assert torch.equal(ops.knn_graph(xb, 3), ops.knn_graph(xb, 3, prefilter=False))


def test_knn_graph_full_batch_properties(dev):
    """BASELINE config-2 size (B=256, stage 0): size-independent properties + a sampled exact check."""
    from grafp_amd import ops
    from oracle import native
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(256, 64, 1024, generator=gen)
    idx = ops.knn_graph(x.to(dev), 3).cpu()
    assert idx.min() >= 0 and idx.max() < 1024
    assert bool((idx[:, :, 0] == torch.arange(1024)).all())          # self first
    assert bool((idx[:, :, 1] != idx[:, :, 2]).all())
    for b in (0, 97, 255):
        assert np.array_equal(idx[b].numpy(), native.knn_graph(x[b:b + 1].numpy(), 3)[0])


def test_knn_graph_errors(dev):
    from grafp_amd import ops
    with pytest.raises(RuntimeError, match="k="):
        ops.knn_graph(torch.zeros(1, 4, 4, device=dev), 9)
    with pytest.raises(RuntimeError, match="CPU tensor"):
        ops.knn_graph(torch.zeros(1, 4, 16), 3)


# =============================================================== MRConv gather / max-relative
@pytest.mark.parametrize("B,C,N,K", [(2, 64, 1024, 3), (3, 128, 512, 3), (2, 256, 256, 3), (2, 512, 128, 3),
                                      (2, 10, 101, 3), (1, 7, 33, 5), (2, 8, 64, 1)])
def test_max_relative_forward_backward(dev, B, C, N, K):
    from grafp_amd import ops
    from oracle import model as om
    x = t(hash_normalish(f"gpu:mr.x.{C}.{N}", (B, C, N)))
    idx = hash_ints(f"gpu:mr.idx.{C}.{N}", (B, N, K), 0, N - 1).astype(np.int64)
    idx[:, :, 0] = np.arange(N)[None, :]
    idx = t(idx)
    g = t(hash_normalish(f"gpu:mr.g.{C}.{N}", (B, 2 * C, N)))
    xr = x.clone().requires_grad_(True)
    want = om.max_relative(xr, idx)
    want.backward(g)
    xg = x.to(dev).requires_grad_(True)
    got = ops.max_relative(xg, idx.to(dev))
    got.backward(g.to(dev))
    assert torch.equal(got.detach().cpu(), want.detach())             # sub + max: exact
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,C,N,K", [(3, 64, 1024, 3), (2, 128, 512, 3), (2, 256, 256, 4), (2, 512, 128, 2), (2, 12, 2048, 1)])
def test_max_relative_backward_from_record_equals_recompute(dev, monkeypatch, B, C, N, K):
    """Training form: the forward pass records which neighbour won (2 bits per element) and the backward pass routes the
    gradients from that record.  Same outputs and the same dx, bit for bit, as the form that recomputes the arg-max from
    x -- f32 and bf16, both layouts, int32 / int64 edges, with exact ties (first maximum wins) and a NaN feature."""
    from grafp_amd import ops
    x = hash_normalish(f"gpu:mra.x.{C}.{N}", (B, C, N)).astype(np.float32)
    x = np.round(x * 4.0) / 4.0                                      # a coarse grid: many exact ties among the neighbours
    x[0, 1, 5] = np.nan
    idx = hash_ints(f"gpu:mra.idx.{C}.{N}", (B, N, K), 0, N - 1).astype(np.int64)
    idx[:, :, 0] = np.arange(N)[None, :]
    g = hash_normalish(f"gpu:mra.g.{C}.{N}", (B, 2 * C, N)).astype(np.float32)
    assert ops.lib.grafp_mrconv_arg_supported(0, C * N, N, 2 * C * N, N, N, K) == 1
    for dt in (torch.float32, torch.bfloat16):
        for layout in ("bcn", "cbn"):
            xt = t(x).to(dt).to(dev)
            gt = t(g).to(dt).to(dev)
            if layout == "cbn":
                xt, gt = xt.permute(1, 0, 2).contiguous(), gt.permute(1, 0, 2).contiguous()
            for it in (torch.int64, torch.int32):
                res = []
                for rec in (True, False):
                    monkeypatch.setattr(ops.switches, "mrconv_arg", rec)
                    xg = xt.clone().requires_grad_(True)
                    out = ops.max_relative(xg, t(idx).to(it).to(dev), layout=layout)
                    out.backward(gt)
                    res.append((out.detach(), xg.grad))
                assert torch.equal(res[0][0].nan_to_num(7.0), res[1][0].nan_to_num(7.0)), (dt, layout, it)
                assert torch.equal(res[0][1].nan_to_num(7.0), res[1][1].nan_to_num(7.0)), (dt, layout, it)


def test_max_relative_golden(dev):
    from grafp_amd import ops
    g = golden("mrconv.npz")
    B, C, N, K = 2, 8, 64, 3
    x = t(hash_normalish("in:mr.x", (B, C, N, 1)))[..., 0].to(dev).requires_grad_(True)
    idx = hash_ints("in:mr.idx", (B, N, K), 0, N - 1).astype(np.int64)
    idx[:, :, 0] = np.arange(N)[None, :]
    out = ops.max_relative(x, t(idx).to(dev))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), g["inter"][..., 0])
    out.backward(t(hash_normalish("in:mr.gi", (B, 2 * C, N, 1)))[..., 0].to(dev))
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["dx_inter"][..., 0], rtol=1e-6, atol=1e-6)


# =============================================================== NT-Xent
@pytest.mark.parametrize("B,D,tau", [(2, 128, 0.05), (8, 128, 0.05), (32, 128, 0.05), (256, 128, 0.05),
                                      (100, 128, 0.05), (33, 32, 0.5), (70, 64, 0.1),
                                      (1024, 128, 0.05)])           # BASELINE config 3: 2048 rows of negatives
def test_ntxent_vs_oracle(dev, B, D, tau):
    """Tolerance: loss 2e-5 relative, gradients 1e-4 relative + 1e-5 of the largest gradient entry (f32
    exp/log, different summation order)."""
    from grafp_amd import ops
    from oracle import model as om
    zi = hash_normalish(f"gpu:nt.zi.{B}.{D}", (B, D)); zj = zi + 2.0 * hash_normalish(f"gpu:nt.zj.{B}.{D}", (B, D))
    zi /= np.linalg.norm(zi, axis=1, keepdims=True); zj /= np.linalg.norm(zj, axis=1, keepdims=True)
    a = t(zi).clone().requires_grad_(True); b = t(zj).clone().requires_grad_(True)
    want = om.ntxent(a, b, tau); want.backward()
    ag = t(zi).to(dev).requires_grad_(True); bg = t(zj).to(dev).requires_grad_(True)
    got = ops.ntxent(ag, bg, tau); got.backward()
    assert want.item() > 1e-3                                         # a non-trivial loss
    np.testing.assert_allclose(got.item(), want.item(), rtol=2e-5)
    gmax = float(a.grad.abs().max())
    np.testing.assert_allclose(ag.grad.cpu().numpy(), a.grad.numpy(), rtol=1e-4, atol=1e-5 * gmax)
    np.testing.assert_allclose(bg.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-5 * gmax)


def test_ntxent_reference_goldens(dev):
    from grafp_amd.simclr.ntxent import ntxent_loss
    g = golden("ntxent.npz")
    for B in (2, 8, 32):
        a = t(g[f"zi_{B}"]).to(dev).requires_grad_(True); b = t(g[f"zj_{B}"]).to(dev).requires_grad_(True)
        loss = ntxent_loss(a, b, {"tau": 0.05}); loss.backward()
        np.testing.assert_allclose(loss.item(), g[f"loss_{B}"], rtol=2e-5)
        np.testing.assert_allclose(a.grad.cpu().numpy(), g[f"dzi_{B}"], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(b.grad.cpu().numpy(), g[f"dzj_{B}"], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("B,R", [(96, 3), (1024, 8)])      # (1024, 8): BASELINE config 3, 128 local rows x 2048 columns
def test_ntxent_local_rows_global_columns(dev, B, R):
    """The data-parallel form: shares of the loss add up to the global loss; local gradients are the
    corresponding slices of the global gradient (no backward collective needed)."""
    from grafp_amd import ops
    D = 128
    zi = hash_normalish(f"gpu:nt.dp.zi{B}", (B, D)); zj = zi + 1.5 * hash_normalish(f"gpu:nt.dp.zj{B}", (B, D))
    zi /= np.linalg.norm(zi, axis=1, keepdims=True); zj /= np.linalg.norm(zj, axis=1, keepdims=True)
    zi_all, zj_all = t(zi).to(dev), t(zj).to(dev)
    a = zi_all.clone().requires_grad_(True); b = zj_all.clone().requires_grad_(True)
    full = ops.ntxent(a, b, 0.05); full.backward()
    total = 0.0
    for r in range(R):
        lo, hi = r * B // R, (r + 1) * B // R
        al = zi_all[lo:hi].clone().requires_grad_(True); bl = zj_all[lo:hi].clone().requires_grad_(True)
        part = ops.ntxent(al, bl, 0.05, zi_all, zj_all, lo); part.backward()
        total += part.item()
        np.testing.assert_allclose(al.grad.cpu().numpy(), a.grad[lo:hi].cpu().numpy(), rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(bl.grad.cpu().numpy(), b.grad[lo:hi].cpu().numpy(), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(total, full.item(), rtol=1e-5)


# =============================================================== log-mel + segmentation
def test_logmel_vs_oracle(dev):
    """Tolerance: 2e-3 dB absolute (f32 FFT of different radix/order than torch.stft's)."""
    from grafp_amd import ops
    from oracle import model as om
    x = 0.1 * hash_normalish("gpu:logmel.x", (5, 16000))
    x[4] *= 0.01                                             # a quiet clip
    want = om.logmel(t(x), CFG).numpy()
    got = ops.logmel(t(x).to(dev)).cpu().numpy()
    assert got.shape == (5, 64, 32)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-3)
    z = ops.logmel(torch.zeros(1, 16000, device=dev)).cpu().numpy()
    assert np.allclose(z, -100.0)                            # 10*log10(1e-10)


def test_whole_track_segments_vs_oracle(dev):
    from grafp_amd.modules.transformations import GPUTransformNeuralfp
    from oracle import model as om
    cfg = dict(CFG, arch="grafp", dur=1.0)
    x = 0.1 * hash_normalish("gpu:logmel.track", (1, 16000 * 7 + 123))
    want = om.val_segments(t(x), CFG).numpy()
    aug = GPUTransformNeuralfp(cfg, None, None, train=False)
    Xi, Xj = aug(t(x).to(dev), None)
    assert Xi.shape == want.shape and Xi.is_contiguous()
    np.testing.assert_allclose(Xi.cpu().numpy(), want, rtol=0, atol=2e-3)
    assert Xj is Xi
    tr = GPUTransformNeuralfp(cfg, None, None, train=True)
    a, b = tr(t(x[:, :16000]).to(dev), t(x[:, 16000:32000]).to(dev))
    np.testing.assert_allclose(b.cpu().numpy(), om.logmel(t(x[:, 16000:32000]), CFG).numpy(), atol=2e-3)


# =============================================================== peak extractor
def test_peak_extract_forward_backward(dev):
    """Tolerance 1e-4 relative / 1e-5 absolute (147-term f32 dot products in a different order)."""
    from grafp_amd import ops
    from oracle import model as om
    for name, B in (("a", 2), ("b", 5)):
        spec = t(40.0 * hash_uniform(f"gpu:peak.spec.{name}", (B, 64, 32)) - 30.0)
        w = t(0.1 * hash_normalish("gpu:peak.w", (8, 3, 7, 7))).requires_grad_(True)
        bias = t(0.05 * hash_normalish("gpu:peak.b", (8,))).requires_grad_(True)
        sd = {"peak_extractor.convs.0.weight": w, "peak_extractor.convs.0.bias": bias}
        want = om.peak_extract(sd, spec)
        g = t(hash_normalish(f"gpu:peak.g.{name}", tuple(want.shape)))
        want.backward(g)
        wg = w.detach().to(dev).requires_grad_(True); bg = bias.detach().to(dev).requires_grad_(True)
        got = ops.peak_extract(spec.to(dev), wg, bg, 2)
        got.backward(g.to(dev))
        np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(wg.grad.cpu().numpy(), w.grad.numpy(), rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(bg.grad.cpu().numpy(), bias.grad.numpy(), rtol=1e-3, atol=1e-4)
        w.grad = None; bias.grad = None


@pytest.mark.parametrize("B,F,K,W", [(700, 8, 7, 32), (1300, 8, 7, 32), (37, 4, 5, 16), (600, 4, 5, 16)])
def test_peak_extract_backward_many_clips_bit_reproducible(dev, B, F, K, W):
    """More clips than workgroups (512): every workgroup accumulates several clips before its row of partial sums is
    written, and the second launch adds the rows in workgroup order -- the weight gradient is a fixed f32 expression, so
    two calls give the SAME bits (it ended in float atomics until round 5: /root/reference/peak_extractor.py:22-30 is the
    conv whose gradient this is).  Shapes: the model's (8 filters, 7x7, 32 frames: the register-blocked kernel) and
    another one (the generic kernel)."""
    from grafp_amd import ops
    from oracle import model as om
    spec = t(40.0 * hash_uniform(f"gpu:peak.many.{B}.{W}", (B, 64, W)) - 30.0)
    w = t(0.1 * hash_normalish(f"gpu:peak.many.w{F}", (F, 3, K, K))).requires_grad_(True)
    bias = t(0.05 * hash_normalish(f"gpu:peak.many.b{F}", (F,))).requires_grad_(True)
    want = om.peak_extract({"peak_extractor.convs.0.weight": w, "peak_extractor.convs.0.bias": bias}, spec)
    g = t(hash_normalish(f"gpu:peak.many.g.{B}.{F}", tuple(want.shape)))
    want.backward(g)
    grads = []
    for _ in range(2):
        wg = w.detach().to(dev).requires_grad_(True); bg = bias.detach().to(dev).requires_grad_(True)
        got = ops.peak_extract(spec.to(dev), wg, bg, 2)
        got.backward(g.to(dev))
        grads.append((wg.grad.clone(), bg.grad.clone()))
    got_c = got.detach().cpu()
    np.testing.assert_allclose(got_c.numpy(), want.detach().numpy(), rtol=1e-4, atol=1e-5)
    # An output within rounding of zero can have its ReLU decided the other way by the two forward passes (a handful out of
    # B * F * 1024 outputs); such a position moves all 3 * K * K taps of ITS filter by O(1).  Filters with a flipped mask are
    # left out of the weight comparison (and there must be few of them); every other filter is compared at f32 accuracy.
    flipped = ((got_c > 0) != (want.detach() > 0)).reshape(B, F, -1).any(dim=2).any(dim=0)
    assert int(((got_c > 0) != (want.detach() > 0)).sum()) <= 8 and int(flipped.sum()) <= F // 2, flipped
    keep = (~flipped).numpy()
    scale = float(w.grad.abs().max())
    np.testing.assert_allclose(grads[0][0].cpu().numpy()[keep], w.grad.numpy()[keep], rtol=1e-3, atol=2e-5 * scale)
    np.testing.assert_allclose(grads[0][1].cpu().numpy()[keep], bias.grad.numpy()[keep], rtol=1e-3,
                               atol=2e-5 * float(bias.grad.abs().max()))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])


def test_peak_extract_reference_golden(dev):
    from _hashfill import fill_state_dict
    from grafp_amd import ops
    g = golden("peak_extractor.npz")
    raw = fill_state_dict({"convs.0.weight": (8, 3, 7, 7), "convs.0.bias": (8,)}, "pe")
    spec = t(40.0 * hash_uniform("in:peak.spec", (2, 64, 32)) - 30.0).to(dev)
    out = ops.peak_extract(spec, t(raw["convs.0.weight"]).to(dev), t(raw["convs.0.bias"]).to(dev), 2)
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-4, atol=1e-5)


# =============================================================== brute-force search (bit-exact ids and distances)
def _search(ops, dbt, qt, k, id_base=0):
    """Both search paths (f32 scan; bf16 pre-filter + exact rescoring) must agree bit for bit; returns one."""
    sq = ops.row_sqnorm(dbt)
    d0, i0 = ops.search_l2(dbt, sq, qt, k, id_base=id_base)
    d1, i1 = ops.search_l2(dbt, sq, qt, k, id_base=id_base, db_bf16=ops.rows_to_bf16(dbt))
    assert torch.equal(i0, i1) and torch.equal(d0, d1)
    return d1, i1


def _planted(n, nq, seed, noise=0.05):
    db = hash_normalish(f"gpu:sr.db.{seed}", (n, 128)); db /= np.linalg.norm(db, axis=1, keepdims=True)
    rows = (np.arange(nq) * 7919) % n
    q = db[rows] + noise * hash_normalish(f"gpu:sr.q.{seed}", (nq, 128))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return db.astype(np.float32), q.astype(np.float32), rows


@pytest.mark.parametrize("n,nq,k", [(1000, 1, 20), (1000, 5, 1), (5000, 41, 20), (5000, 70, 20), (20000, 300, 20),
                                     (777, 33, 32), (129, 64, 7), (50000, 41, 20), (30000, 1100, 20)])
def test_search_bit_exact_vs_c_oracle(dev, n, nq, k):
    from grafp_amd import ops
    from oracle import native
    db, q, rows = _planted(n, nq, f"{n}.{nq}")
    want_d, want_i = native.flat_search_l2(db, q, k)
    dbt = t(db).to(dev)
    got_d, got_i = _search(ops, dbt, t(q).to(dev), k)
    got_d, got_i = got_d.cpu().numpy(), got_i.cpu().numpy()
    mism = np.argwhere(got_i != want_i)
    assert len(mism) == 0, f"{len(mism)} ids differ; first {mism[:5].tolist()}"
    assert np.array_equal(got_d, want_d)
    assert np.array_equal(got_i[:, 0], rows)                          # planted answers


def test_search_edge_cases(dev):
    from grafp_amd import ops
    from oracle import native
    db, q, _ = _planted(300, 9, "edge")
    dup = np.tile(db[:25], (8, 1))                                    # every row 8 times: ties -> lowest id
    dupt = t(dup).to(dev)
    d, i = _search(ops, dupt, t(db[:9]).to(dev), 20)
    wd, wi = native.flat_search_l2(dup, db[:9], 20)
    assert np.array_equal(i.cpu().numpy(), wi) and np.array_equal(d.cpu().numpy(), wd)
    small = t(db[:5]).to(dev)                                         # fewer rows than k: -1 / inf padding
    d, i = _search(ops, small, t(q).to(dev), 20)
    assert (i[:, 5:] == -1).all() and torch.isinf(d[:, 5:]).all() and (i[:, :5] >= 0).all()
    idx = ops.FlatL2Index(128)                                        # faiss-like surface, numpy in/out
    idx.add(db[:100]); idx.add(db[100:])
    D, I = idx.search(q, 20)
    wd, wi = native.flat_search_l2(db, q, 20)
    assert isinstance(I, np.ndarray) and np.array_equal(I, wi) and np.array_equal(D, wd)


def test_search_candidate_overflow_rescan(dev):
    """More exact ties than a query's candidate list holds (SR_CAP = 4096 in knn_search.hip): the select kernel
    must fall back to its exact rescan and still return the lowest ids, bit-equal to the C oracle."""
    from grafp_amd import ops
    from oracle import native
    db, q, _ = _planted(12000, 3, "overflow")
    db = db.copy()
    db[1000:7000] = db[1000]                                          # 6000 identical rows
    qs = np.stack([db[1000], q[0], db[1000] * 0.5 + q[1] * 0.5]).astype(np.float32)
    dbt = t(db).to(dev)
    d, i = _search(ops, dbt, t(qs).to(dev), 20)
    wd, wi = native.flat_search_l2(db, qs, 20)
    assert np.array_equal(i.cpu().numpy(), wi) and np.array_equal(d.cpu().numpy(), wd)
    assert (i[0].cpu().numpy() == np.arange(1000, 1020)).all()


@pytest.mark.parametrize("n,nq", [(1, 1), (31, 2), (33, 33), (64, 64), (65, 65), (127, 65), (4097, 40), (8191, 129),
                                  (70001, 7)])
def test_search_ragged_sizes(dev, n, nq):
    """Row counts around the tile (32/64/128 rows: the 64-row ring stage of the bf16 scan ends inside its last tile) and
    sample (64k rows) boundaries, query counts around the wave-shape switches (64 / 65, 128 / 129), k > n included."""
    from grafp_amd import ops
    from oracle import native
    db, q, _ = _planted(max(n, 64), max(nq, 8), f"ragged{n}")
    db, q = db[:n], q[:nq]
    dbt = t(db).to(dev)
    d, i = _search(ops, dbt, t(q).to(dev), 20)
    wd, wi = native.flat_search_l2(db, q, 20)
    assert np.array_equal(i.cpu().numpy(), wi) and np.array_equal(d.cpu().numpy(), wd)


def test_search_prefilter_worst_case_rounding(dev):
    """The bf16 pre-filter must never drop a true neighbour.  Components sit just below the midpoint between two bf16
    values (relative rounding error ~ 2^-8, the most the margin has to absorb), norms vary over 3 orders of magnitude
    (the margin scales with qq + dd), and the queries are near-duplicates of rows, so the k-th distances are small
    compared with the error the margin covers."""
    from grafp_amd import ops
    from oracle import native
    n, nq = 30000, 96
    base = hash_normalish("gpu:sr.worst", (n, 128)).astype(np.float32)
    bits = base.view(np.uint32)
    bits = (bits & np.uint32(0xFFFF0000)) | np.uint32(0x7FFF)          # mantissa tail 0x7fff: rounds down by ~2^-9 rel
    up = hash_ints("gpu:sr.worst.up", (n, 128), 0, 1).astype(bool)
    bits = np.where(up, (bits & np.uint32(0xFFFF0000)) | np.uint32(0x8001), bits)    # ... or up by the same amount
    db = bits.view(np.float32) * (10.0 ** (1.5 * hash_uniform("gpu:sr.worst.scale", (n, 1)))).astype(np.float32)
    rows = (np.arange(nq) * 311) % n
    q = (db[rows] * (1.0 + 1e-3 * hash_normalish("gpu:sr.worst.q", (nq, 128)))).astype(np.float32)
    dbt = t(db).to(dev)
    d, i = _search(ops, dbt, t(q).to(dev), 20)
    wd, wi = native.flat_search_l2(db, q, 20)
    assert np.array_equal(i.cpu().numpy(), wi) and np.array_equal(d.cpu().numpy(), wd)


def test_merge_topk_and_sharded_search(dev):
    from grafp_amd import ops
    from oracle import native
    db, q, _ = _planted(9000, 50, "shard")
    parts_d, parts_i = [], []
    for s in range(0, 9000, 3000):
        sh = t(db[s:s + 3000]).to(dev)
        d, i = _search(ops, sh, t(q).to(dev), 20, id_base=s)
        parts_d.append(d); parts_i.append(i)
    md, mi = ops.merge_topk(torch.stack(parts_d), torch.stack(parts_i))
    wd, wi = native.flat_search_l2(db, q, 20)
    assert np.array_equal(mi.cpu().numpy(), wi) and np.array_equal(md.cpu().numpy(), wd)
    od, oi = native.merge_topk(torch.stack(parts_d).cpu().numpy(), torch.stack(parts_i).cpu().numpy())
    assert np.array_equal(mi.cpu().numpy(), oi) and np.array_equal(md.cpu().numpy(), od)


def test_search_two_part_scan_vs_c_oracle(dev):
    """From 768 queries on (and n/4 >= 64k rows) the bf16 scan runs in two parts with the bound tightened in between
    (knn_search.hip, grafp_knn_search_l2_pre).  300 000 rows x 800 queries: every query against the f32 path, a sample
    against the C oracle; exact duplicates of one row planted on both sides of the part boundary (row 74 944) must come
    back lowest id first; and a query whose first-part candidate list overflows (5 000 copies of one row inside part one)
    must fall back to the exact rescan."""
    from grafp_amd import ops
    from oracle import native
    n, nq, k = 300_000, 800, 20
    db, q, rows = _planted(n, nq, "twopart")
    db = db.copy()
    dup = db[123].copy()
    dups = (74_000, 74_943, 74_944, 74_945, 75_008, 200_000, 299_999)
    for r in dups:
        db[r] = dup
    db[10_000:15_000] = db[10_000]                                    # overflow of one query's sub-lists in part one
    q = q.copy()
    q[5] = dup
    q[6] = db[10_000]
    dbt, qt = t(db).to(dev), t(q).to(dev)
    d, i = _search(ops, dbt, qt, k)                                   # asserts pre-filter path == f32 path
    d, i = d.cpu().numpy(), i.cpu().numpy()
    assert list(i[5, :8]) == [123, *dups] and (d[5, :8] == d[5, 0]).all()
    assert list(i[6]) == list(range(10_000, 10_020))
    sample = np.r_[0:48, 5, 6, nq - 16:nq]
    wd, wi = native.flat_search_l2(db, q[sample], k)
    assert np.array_equal(i[sample], wi) and np.array_equal(d[sample], wd)
    keep = ~np.isin(rows, dups) & ~((rows >= 10_000) & (rows < 15_000)) & (rows != 123)
    keep[[5, 6]] = False
    assert keep.sum() > 700 and np.array_equal(i[keep, 0], rows[keep])    # planted answers
    d41, i41 = _search(ops, dbt, qt[:41], k)                          # one-part scan of the same queries: same bits
    assert np.array_equal(i41.cpu().numpy(), i[:41]) and np.array_equal(d41.cpu().numpy(), d[:41])


@pytest.mark.parametrize("nq", [1536, 1921])
def test_search_three_query_sets_vs_c_oracle(dev, nq):
    """From 1536 queries a wave of the bf16 scan carries THREE query sets where the last 384-query group pads little
    (knn_search.hip, pre_plan: nq = 1536 -> 4 groups of 384 exactly; 1921 -> 6 groups, 383 idle slots).  200 000 rows
    (two-part scan): every query against the all-f32 path (`_search` asserts equal bits), a sample across all groups and
    query sets against the C oracle, the planted answers, and the first 41 through the one-set form: same bits."""
    from grafp_amd import ops
    from oracle import native
    n, k = 200_000, 20
    db, q, rows = _planted(n, nq, f"threesets{nq}")
    dbt, qt = t(db).to(dev), t(q).to(dev)
    d, i = _search(ops, dbt, qt, k)
    d, i = d.cpu().numpy(), i.cpu().numpy()
    sample = np.unique(np.r_[0:8, 30:34, 95:97, 127:130, 383:386, 767:770, 1151:1154, nq - 40:nq])
    wd, wi = native.flat_search_l2(db, q[sample], k)
    assert np.array_equal(i[sample], wi) and np.array_equal(d[sample], wd)
    assert np.array_equal(i[:, 0], rows)                              # planted answers, every query
    d41, i41 = _search(ops, dbt, qt[:41], k)
    assert np.array_equal(i41.cpu().numpy(), i[:41]) and np.array_equal(d41.cpu().numpy(), d[:41])


def test_search_1m_planted_top1(dev):
    """BASELINE config 4 size: 1 000 000 x 128 database resident on the GPU; planted noisy queries must
    come back top-1, and a sample of queries must match the CPU oracle exactly."""
    from grafp_amd import ops
    from oracle import native
    gen = torch.Generator(device="cpu").manual_seed(2)
    db = torch.nn.functional.normalize(torch.randn(1_000_000, 128, generator=gen), dim=1)
    rows = torch.randint(0, 1_000_000, (4096,), generator=gen)
    q = torch.nn.functional.normalize(db[rows] + 0.03 * torch.randn(4096, 128, generator=gen), dim=1)
    dbt = db.to(dev)
    sq = ops.row_sqnorm(dbt)
    dbh = ops.rows_to_bf16(dbt)
    d, i = ops.search_l2(dbt, sq, q.to(dev), 20, db_bf16=dbh)
    d0, i0 = ops.search_l2(dbt, sq, q.to(dev), 20)
    assert torch.equal(i, i0) and torch.equal(d, d0)                  # pre-filter path == f32 scan, 4096 queries
    assert torch.equal(i[:, 0].cpu(), rows)
    assert bool((d[:, 1:] >= d[:, :-1]).all())                        # sorted ascending
    for nq in (1, 41):
        dd, ii = ops.search_l2(dbt, sq, q[:nq].to(dev), 20, db_bf16=dbh)
        wd, wi = native.flat_search_l2(db.numpy(), q[:nq].numpy(), 20)
        assert np.array_equal(ii.cpu().numpy(), wi) and np.array_equal(dd.cpu().numpy(), wd)
        assert torch.equal(ii.cpu(), i[:nq].cpu())                    # batch-size independent


def test_search_10m_sharded_equals_unsharded(dev):
    """BASELINE config 5 size: 10 000 000 x 128 fingerprints (5.1 GB + the bf16 copy) resident on ONE MI355X.
    Size-independent properties: planted noisy queries come back top-1; the database cut into the 8 contiguous shards
    the 8-GPU run uses (searched one after the other here, ids offset by the shard start) and merged equals the
    unsharded search bit for bit -- ids and distances; both search paths agree."""
    from grafp_amd import ops
    from grafp_amd.dist import shard_range
    n, nq, k = 10_000_000, 2048, 20
    gen = torch.Generator(device=dev).manual_seed(5)
    db = torch.empty((n, 128), dtype=torch.float32, device=dev)
    for lo in range(0, n, 1_000_000):
        db[lo:lo + 1_000_000] = torch.nn.functional.normalize(
            torch.randn(1_000_000, 128, generator=gen, device=dev), dim=1)
    rows = torch.randint(0, n, (nq,), generator=gen, device=dev)
    q = torch.nn.functional.normalize(db[rows] + 0.03 * torch.randn(nq, 128, generator=gen, device=dev), dim=1)
    sq = ops.row_sqnorm(db)
    dbh = ops.rows_to_bf16(db)
    d, i = ops.search_l2(db, sq, q, k, db_bf16=dbh)
    assert torch.equal(i[:, 0], rows)
    assert bool((d[:, 1:] >= d[:, :-1]).all())
    d0, i0 = ops.search_l2(db, sq, q[:256], k)                       # all-f32 path on a slice of the queries
    assert torch.equal(i0, i[:256]) and torch.equal(d0, d[:256])
    parts_d, parts_i = [], []
    for r in range(8):
        lo, hi = shard_range(n, r, 8)
        pd, pi = ops.search_l2(db[lo:hi], sq[lo:hi], q, k, id_base=lo, db_bf16=dbh[lo:hi])
        assert int(pi.min()) >= lo and int(pi.max()) < hi
        parts_d.append(pd); parts_i.append(pi)
    md, mi = ops.merge_topk(torch.stack(parts_d), torch.stack(parts_i))
    assert torch.equal(mi, i) and torch.equal(md, d)


# =============================================================== (C,B,N) layout + bf16 variants
@pytest.mark.parametrize("B,C,N,k", [(3, 64, 1024, 3), (2, 512, 128, 3), (2, 10, 101, 4)])
def test_knn_graph_cbn_layout_and_bf16(dev, B, C, N, k):
    """The strided entry reads the GEMM-friendly (C,B,N) layout and bf16 activations; results are the same
    indices as the (B,C,N) f32 path on the widened values (bit-exact vs the C oracle)."""
    from grafp_amd import ops
    from oracle import native
    x = hash_normalish(f"gpu:knn.cbn.{C}.{N}", (B, C, N))
    want = native.knn_graph(x, k)
    xt = t(x).to(dev)
    got = ops.knn_graph(xt.permute(1, 0, 2).contiguous(), k, layout="cbn").cpu().numpy()
    assert np.array_equal(got, want)
    xb = xt.to(torch.bfloat16)
    want16 = native.knn_graph(xb.float().cpu().numpy(), k)
    assert np.array_equal(ops.knn_graph(xb, k).cpu().numpy(), want16)
    assert np.array_equal(ops.knn_graph(xb.permute(1, 0, 2).contiguous(), k, layout="cbn").cpu().numpy(), want16)


@pytest.mark.parametrize("B,C,N,K", [(2, 64, 1024, 3), (3, 512, 128, 3), (2, 10, 101, 3)])
def test_max_relative_cbn_layout_and_bf16(dev, B, C, N, K):
    from grafp_amd import ops
    from oracle import model as om
    x = t(hash_normalish(f"gpu:mr.cbn.x.{C}.{N}", (B, C, N)))
    idx = hash_ints(f"gpu:mr.cbn.idx.{C}.{N}", (B, N, K), 0, N - 1).astype(np.int64)
    idx[:, :, 0] = np.arange(N)[None, :]
    idx = t(idx)
    g = t(hash_normalish(f"gpu:mr.cbn.g.{C}.{N}", (B, 2 * C, N)))
    xr = x.clone().requires_grad_(True)
    want = om.max_relative(xr, idx); want.backward(g)
    xg = x.to(dev).permute(1, 0, 2).contiguous().requires_grad_(True)
    got = ops.max_relative(xg, idx.to(dev), layout="cbn")
    assert got.shape == (2 * C, B, N)
    got.backward(g.to(dev).permute(1, 0, 2).contiguous())
    assert torch.equal(got.detach().permute(1, 0, 2).cpu(), want.detach())
    np.testing.assert_allclose(xg.grad.permute(1, 0, 2).cpu().numpy(), xr.grad.numpy(), rtol=1e-5, atol=1e-6)
    # bf16 in/out: f32 arithmetic on the widened values, one rounding on store
    xb = x.to(torch.bfloat16)
    wantb = om.max_relative(xb.float(), idx).to(torch.bfloat16)
    gotb = ops.max_relative(xb.to(dev).permute(1, 0, 2).contiguous(), idx.to(dev), layout="cbn")
    assert gotb.dtype == torch.bfloat16 and torch.equal(gotb.permute(1, 0, 2).cpu(), wantb)


def _bn_ref(x, gamma, beta, rm, rv, training, pb, res, act, slope, eps=1e-5, mom=0.1):
    import torch.nn.functional as F
    C = x.shape[0]
    y = x.reshape(1, C, -1).double()
    if pb is not None:
        y = y + pb.double().reshape(1, C, 1)
    y = F.batch_norm(y, rm, rv, gamma.double(), beta.double(), training, mom, eps).reshape(x.shape)
    y = F.relu(y) if act == 1 else (F.leaky_relu(y, slope) if act == 2 else y)
    return y if res is None else y + res.double()


@pytest.mark.parametrize("C,M,act,use_pb,use_res,training", [
    (64, 4 * 1024, 1, True, False, True), (256, 3 * 256, 0, True, True, True), (2048, 2 * 128, 1, False, False, True),
    (8, 5 * 1024, 2, False, False, True), (16, 1001, 1, True, True, True), (64, 2048, 1, True, True, False),
    (3, 7, 0, False, True, True)])
def test_bn_act_forward_backward_f32(dev, C, M, act, use_pb, use_res, training):
    """Fused [bias]+BatchNorm+act+residual vs torch (float64 reference): forward 2e-5, grads 1e-4 of their max,
    running statistics 1e-5 -- including rows with |mean| >> std (shifted-sum statistics)."""
    from grafp_amd import ops
    x = t(hash_normalish(f"gpu:bn.x.{C}.{M}", (C, M))) * 2.0 + 30.0 * t(hash_uniform(f"gpu:bn.mu.{C}", (C, 1)))
    gamma = 1.0 + 0.2 * t(hash_uniform(f"gpu:bn.g.{C}", (C,))); beta = 0.3 * t(hash_uniform(f"gpu:bn.b.{C}", (C,)))
    pb = 0.5 * t(hash_uniform(f"gpu:bn.pb.{C}", (C,))) if use_pb else None
    res = t(hash_normalish(f"gpu:bn.r.{C}.{M}", (C, M))) if use_res else None
    rm0 = 0.1 * t(hash_uniform(f"gpu:bn.rm.{C}", (C,))); rv0 = 1.0 + 0.5 * t(hash_uniform(f"gpu:bn.rv.{C}", (C,))).abs()
    gz = t(hash_normalish(f"gpu:bn.gz.{C}.{M}", (C, M)))
    # reference (float64 autograd)
    xr = x.double().requires_grad_(True); gr = gamma.double().requires_grad_(True); br = beta.double().requires_grad_(True)
    rr = res.double().requires_grad_(True) if use_res else None
    pr = pb.double().requires_grad_(True) if use_pb else None
    rm_ref, rv_ref = rm0.double().clone(), rv0.double().clone()
    want = _bn_ref(xr, gr, br, rm_ref, rv_ref, training, pr, rr, act, 0.2)
    want.backward(gz.double())
    # HIP
    xg = x.to(dev).requires_grad_(True); gg = gamma.to(dev).requires_grad_(True); bg = beta.to(dev).requires_grad_(True)
    rg = res.to(dev).requires_grad_(True) if use_res else None
    pg = pb.to(dev).requires_grad_(True) if use_pb else None
    rm, rv = rm0.to(dev).clone(), rv0.to(dev).clone()
    got = ops.bn_act(xg, gg, bg, rm, rv, training, 0.1, 1e-5, pg, rg, act, 0.2)
    got.backward(gz.to(dev))
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=2e-5, atol=2e-5)
    for a, b in ((xg.grad, xr.grad), (gg.grad, gr.grad), (bg.grad, br.grad)):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=1e-4, atol=1e-4 * float(b.abs().max()) + 1e-7)
    if use_res:
        assert torch.equal(rg.grad.cpu(), gz)
    if use_pb:
        np.testing.assert_allclose(pg.grad.cpu().numpy(), pr.grad.numpy(), rtol=1e-4,
                                   atol=2e-4 * float(gz.abs().sum(1).max()) if training else 1e-4 * float(pr.grad.abs().max()))
    np.testing.assert_allclose(rm.cpu().numpy(), rm_ref.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rv.cpu().numpy(), rv_ref.numpy(), rtol=1e-4, atol=1e-5)


def test_bn_act_bf16(dev):
    """bf16 activations in/out, f32 statistics: equals the f32 computation on the widened inputs up to the final
    bf16 rounding (1 ulp = 2^-8 relative)."""
    from grafp_amd import ops
    C, M = 128, 4096
    x = t(hash_normalish("gpu:bn16.x", (C, M))).to(torch.bfloat16)
    res = t(hash_normalish("gpu:bn16.r", (C, M))).to(torch.bfloat16)
    gamma = 1.0 + 0.2 * t(hash_uniform("gpu:bn16.g", (C,))); beta = 0.3 * t(hash_uniform("gpu:bn16.b", (C,)))
    want = _bn_ref(x.float(), gamma, beta, None, None, True, None, res.float(), 1, 0.0).float()
    xg = x.to(dev).requires_grad_(True)
    got = ops.bn_act(xg, gamma.to(dev), beta.to(dev), None, None, True, residual=res.to(dev), act=1)
    assert got.dtype == torch.bfloat16
    np.testing.assert_allclose(got.detach().float().cpu().numpy(), want.numpy(), rtol=2 ** -7, atol=2 ** -7)
    got.float().sum().backward()
    assert xg.grad.dtype == torch.bfloat16 and torch.isfinite(xg.grad.float()).all()


@pytest.mark.parametrize("C,M,G,dt", [(64, 2 * 131072, 2, "bf16"), (24, 3 * 40000, 3, "f32"), (256, 65536, 2, "bf16"),
                                      (8, 2 * 16384 + 16, 1, "bf16"), (5, 8192 * 300, 1, "f32")])
def test_bn_single_pass_equals_two_pass(dev, monkeypatch, C, M, G, dt):
    """The single-pass training kernels (chunk held in registers across the row rendezvous; rows of many chunks,
    ragged last chunk, 3 groups, f32 and bf16) against the two-pass kernels: outputs/gradients equal up to the
    different partial-sum chunking (1e-6 relative on the statistics), and the rendezvous buffer is all-ones again
    after every call.  The last case exceeds 256 chunks per row and must fall back by itself."""
    from grafp_amd import ops
    dtype = torch.float32 if dt == "f32" else torch.bfloat16
    gen = torch.Generator(device=dev).manual_seed(C * 7 + G)
    x = (torch.randn(C, M, device=dev, generator=gen) * 1.5 + 4.0 * torch.randn(C, 1, device=dev, generator=gen)).to(dtype)
    res = torch.randn(C, M, device=dev, generator=gen).to(dtype)
    gz = torch.randn(C, M, device=dev, generator=gen).to(dtype)
    gamma = 1.0 + 0.2 * torch.randn(C, device=dev, generator=gen); beta = 0.3 * torch.randn(C, device=dev, generator=gen)
    pb = 0.5 * torch.randn(C, device=dev, generator=gen)

    def run(two_pass):
        monkeypatch.setattr(ops.switches, "bn_two_pass", bool(two_pass))
        xg = x.clone().requires_grad_(True); gg = gamma.clone().requires_grad_(True); bg = beta.clone().requires_grad_(True)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        out = ops.bn_act(xg, gg, bg, rm, rv, True, 0.1, 1e-5, pb, res, 2, 0.2, G)
        out.backward(gz)
        return [v.detach().float() for v in (out, xg.grad, gg.grad, bg.grad, rm, rv)]
    one, two = run(False), run(True)
    torch.cuda.synchronize()
    for buf in ops._BN_SYNC.values():
        assert bool((buf == -1).all()), "rendezvous buffer not re-armed"
    ulp = 2 ** -7 if dt == "bf16" else 2e-6
    for a, b, name in zip(one, two, ("out", "dx", "dgamma", "dbeta", "running_mean", "running_var")):
        scale = float(b.abs().max()) + 1e-12
        err = float((a - b).abs().max()) / scale
        assert err <= (ulp if name in ("out", "dx") else 2e-5), (name, err)


@pytest.mark.parametrize("cout,cin,groups,M", [(64, 64, 1, 8192), (256, 64, 1, 4096), (64, 128, 1, 3000), (128, 128, 4, 8192),
                                                (512, 128, 1, 2048), (40, 24, 1, 777), (96, 192, 4, 1000)])
def test_conv1x1_wgrad_bf16(dev, cout, cin, groups, M):
    """Split-K weight gradient vs a float64 product of the same bf16 operands: 2e-3 of the largest entry (f32
    accumulation of bf16 products in a different order)."""
    from grafp_amd import ops
    x = t(hash_normalish(f"gpu:wg.x.{cin}.{M}", (cin, M))).to(torch.bfloat16)
    w = t(0.1 * hash_normalish(f"gpu:wg.w.{cout}.{cin}", (cout, cin // groups)))
    g = t(hash_normalish(f"gpu:wg.g.{cout}.{M}", (cout, M))).to(torch.bfloat16)
    xg = x.to(dev).requires_grad_(True); wg = w.to(dev).requires_grad_(True)
    y = ops.conv1x1_rows(xg, wg, groups)
    y.backward(g.to(dev))
    xd, gd = x.double(), g.double()
    want = torch.cat([gd.reshape(groups, cout // groups, M)[i] @ xd.reshape(groups, cin // groups, M)[i].t()
                      for i in range(groups)], dim=0)
    assert wg.grad.shape == (cout, cin // groups) and wg.grad.dtype == torch.float32
    np.testing.assert_allclose(wg.grad.cpu().numpy(), want.numpy(), rtol=1e-3, atol=2e-3 * float(want.abs().max()))
    dense = torch.block_diag(*w.to(torch.bfloat16).double().reshape(groups, cout // groups, cin // groups).unbind(0))
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), (dense @ xd).numpy(), rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(xg.grad.float().cpu().numpy(), (dense.t() @ gd).numpy(), rtol=2e-2, atol=3e-2)


@pytest.mark.parametrize("cout,cin,groups,M", [(64, 64, 1, 8192), (256, 64, 1, 40000), (128, 128, 4, 8192), (512, 128, 1, 2048),
                                                (40, 24, 1, 777), (96, 192, 4, 1001), (64, 256, 1, 262144)])
def test_conv1x1_wgrad_f32_split_bf16(dev, monkeypatch, cout, cin, groups, M):
    """Weight gradient of f32 operands through the split-bf16 (hi/lo, three MFMAs) streaming kernel vs a float64
    product: 3e-5 of the largest entry (2^-16 per product, averaged over M terms); the library f32 GEMM on the same
    data as a cross-check of the bar (it lands at ~1e-6)."""
    from grafp_amd import ops
    gen = torch.Generator(device=dev).manual_seed(cout * 31 + cin)
    x = torch.randn(cin, M, device=dev, generator=gen) + 0.5
    g = torch.randn(cout, M, device=dev, generator=gen) * 0.3
    w = torch.randn(cout, cin // groups, device=dev, generator=gen) * 0.1

    def run():
        xg = x.clone().requires_grad_(True); wg = w.clone().requires_grad_(True)
        ops.conv1x1_rows(xg, wg, groups).backward(g)
        return wg.grad
    got = run()
    monkeypatch.setattr(ops.switches, "wgrad_f32_library", True)
    lib_ = run()
    xd, gd = x.double(), g.double()
    want = torch.cat([gd.reshape(groups, cout // groups, M)[i] @ xd.reshape(groups, cin // groups, M)[i].t()
                      for i in range(groups)], dim=0)
    scale = float(want.abs().max())
    assert got.shape == (cout, cin // groups) and got.dtype == torch.float32
    assert float((got.double() - want).abs().max()) <= 3e-5 * scale
    assert float((lib_.double() - want).abs().max()) <= 3e-5 * scale


@pytest.mark.parametrize("C,M,G,dt", [(64, 8192, 2, "f32"), (256, 1536, 2, "f32"), (16, 3000, 3, "f32"), (128, 4096, 2, "bf16")])
def test_bn_act_groups_equal_sequential_calls(dev, C, M, G, dt):
    """groups=G (views stacked along the columns) == G separate calls on the column segments: same outputs, same
    input gradients, parameter gradients = the sum, running statistics advanced once per view in order."""
    from grafp_amd import ops
    dtype = torch.float32 if dt == "f32" else torch.bfloat16
    x = (t(hash_normalish(f"gpu:bng.x.{C}.{M}", (C, M))) * 1.5 + 3.0 * t(hash_uniform(f"gpu:bng.mu.{C}", (C, 1)))).to(dtype)
    res = t(hash_normalish(f"gpu:bng.r.{C}.{M}", (C, M))).to(dtype)
    gz = t(hash_normalish(f"gpu:bng.gz.{C}.{M}", (C, M))).to(dtype)
    gamma = 1.0 + 0.2 * t(hash_uniform(f"gpu:bng.g.{C}", (C,))); beta = 0.3 * t(hash_uniform(f"gpu:bng.b.{C}", (C,)))
    pb = 0.5 * t(hash_uniform(f"gpu:bng.pb.{C}", (C,)))
    def run(groups):
        xg = x.to(dev).requires_grad_(True); gg = gamma.to(dev).requires_grad_(True); bg = beta.to(dev).requires_grad_(True)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        if groups > 1:
            out = ops.bn_act(xg, gg, bg, rm, rv, True, 0.1, 1e-5, pb.to(dev), res.to(dev), 1, 0.0, groups)
        else:
            segs = [ops.bn_act(xs.contiguous(), gg, bg, rm, rv, True, 0.1, 1e-5, pb.to(dev), rs.contiguous(), 1, 0.0)
                    for xs, rs in zip(xg.chunk(G, dim=1), res.to(dev).chunk(G, dim=1))]
            out = torch.cat(segs, dim=1)
        out.backward(gz.to(dev))
        return out.detach().float().cpu(), xg.grad.float().cpu(), gg.grad.cpu(), bg.grad.cpu(), rm.cpu(), rv.cpu()
    a, b = run(G), run(1)
    tol = dict(rtol=1e-5, atol=1e-5) if dt == "f32" else dict(rtol=2 ** -7, atol=2 ** -7)
    np.testing.assert_allclose(a[0].numpy(), b[0].numpy(), **tol)
    np.testing.assert_allclose(a[1].numpy(), b[1].numpy(), rtol=tol["rtol"], atol=tol["atol"] * float(b[1].abs().max()) + 1e-7)
    for i in (2, 3):
        np.testing.assert_allclose(a[i].numpy(), b[i].numpy(), rtol=2e-3 if dt == "bf16" else 1e-4, atol=1e-3 * float(b[i].abs().max()))
    np.testing.assert_allclose(a[4].numpy(), b[4].numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(a[5].numpy(), b[5].numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n,nq_rows,k,lens", [(300, 40, 6, (3, 5, 1, 19, 1)), (5000, 200, 20, (1, 11, 21, 41)),
                                              (1000, 64, 32, (64, 1, 33)), (50, 10, 3, (2, 10))])
def test_seq_rerank_bit_exact_vs_c_oracle(dev, n, nq_rows, k, lens):
    """ops.seq_rerank == oracle/csrc/seq_rerank.c: ids AND scores, bit for bit, on items with ids < 0, candidates
    that run past the end of the index, duplicate candidates, 1..64 segments and up to 2048 candidates per item."""
    from grafp_amd import ops
    from oracle import native
    index_rows = hash_normalish(f"gpu:rr.index{n}", (n, 128)).astype(np.float32)
    q = hash_normalish(f"gpu:rr.q{n}", (nq_rows, 128)).astype(np.float32)
    ids = hash_ints(f"gpu:rr.ids{n}", (nq_rows, k), 0, n - 1).astype(np.int64)
    ids[1, k // 2] = -1
    ids[2, :] = n - 2
    ids[3, : k // 2] = ids[2, 0] + 1
    ids[nq_rows // 2:, 0] = np.arange(nq_rows - nq_rows // 2) + 7      # a planted run: equal start id after compensation
    rows, ln = [], []
    for i in range(24):
        L = lens[i % len(lens)]
        rows.append((i * 5) % (nq_rows - L + 1)); ln.append(L)
    item_row, item_len = np.asarray(rows, dtype=np.int64), np.asarray(ln, dtype=np.int32)
    want_i, want_s = native.seq_rerank(index_rows, q, ids, item_row, item_len, top=10)
    got_i, got_s = ops.seq_rerank(t(index_rows).to(dev), t(q).to(dev), torch.from_numpy(ids).to(dev),
                                  torch.from_numpy(item_row).to(dev), torch.from_numpy(item_len).to(dev), top=10)
    assert np.array_equal(got_i.cpu().numpy(), want_i)
    assert np.array_equal(got_s.cpu().numpy(), want_s)


@pytest.mark.parametrize("shape", [(5, 3, 64), (8, 2, 101), (2, 4, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stride2_taps_vs_pad_slice_cat(dev, shape, dtype):
    """ops.stride2_taps == F.pad + three strided slices (what the stride-2 3x1 convolution of Downsample reads),
    forward exactly and backward exactly in f32 / to bf16 rounding."""
    import torch.nn.functional as F
    from grafp_amd import ops
    x = t(hash_normalish(f"gpu:taps.{shape}", shape)).to(dev).to(dtype)
    N = shape[-1]
    n_out = (N - 1) // 2 + 1
    xr = x.clone().float().requires_grad_(True)
    xp = F.pad(xr, (1, 1))
    want = torch.stack([xp[..., t0:t0 + 2 * n_out - 1:2] for t0 in range(3)], dim=0)
    xg = x.clone().requires_grad_(True)
    got = ops.stride2_taps(xg)
    assert got.shape == want.shape and got.dtype == dtype
    assert torch.equal(got.float(), want.detach().to(dtype).float())
    g = t(hash_normalish(f"gpu:taps.g.{shape}", tuple(want.shape))).to(dev)
    want.backward(g.to(dtype).float())
    got.backward(g.to(dtype))
    tol = 0.0 if dtype == torch.float32 else 1e-2
    assert torch.allclose(xg.grad.float(), xr.grad, rtol=tol, atol=tol)


def test_c_abi_entries_are_graph_capturable(dev):
    """The C entries only enqueue kernels on the stream they are given (no allocation, no synchronisation): a search
    and a k-NN graph build captured into a HIP graph replay correctly on new input contents."""
    from grafp_amd import ops
    db, q, _ = _planted(20000, 41, "graph")
    dbt, qt = t(db).to(dev), t(q).to(dev)
    sq, dbh = ops.row_sqnorm(dbt), ops.rows_to_bf16(dbt)
    x = t(hash_normalish("gpu:graph.knn", (4, 64, 256))).to(dev)
    want_d, want_i = ops.search_l2(dbt, sq, qt, 20, db_bf16=dbh)
    want_g = ops.knn_graph(x, 3)
    q_static, x_static = torch.zeros_like(qt), torch.zeros_like(x)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # warm-up on the capture stream (allocator, module loading)
        ops.search_l2(dbt, sq, q_static, 20, db_bf16=dbh)
        ops.knn_graph(x_static, 3)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_d, out_i = ops.search_l2(dbt, sq, q_static, 20, db_bf16=dbh)
        out_g = ops.knn_graph(x_static, 3)
    q_static.copy_(qt)
    x_static.copy_(x)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_i, want_i) and torch.equal(out_d, want_d) and torch.equal(out_g, want_g)


@pytest.mark.parametrize("B,T,L", [(3, 16000, 16000), (2, 5000, 700), (1, 40000, 3001), (4, 2048, 9), (2, 777, 2000)])
def test_ir_convolve_bit_exact_vs_oracle(dev, B, T, L):
    """ApplyImpulseResponse on the device: every output is the oracle's fmaf chain (taps ascending) -> equal bits;
    skipped signals (index -1) are copies; responses longer than the signal, ragged lengths, multi-tile signals."""
    from grafp_amd import ops
    from oracle import native as on
    rng = np.random.default_rng(B * 1000 + L)
    x = rng.standard_normal((B, T)).astype(np.float32)
    bank = np.zeros((3, L), np.float32)
    lens = np.array([L, max(1, L // 3), 1], np.int32)
    for i in range(3):
        bank[i, :lens[i]] = rng.standard_normal(lens[i]) * np.exp(-np.arange(lens[i]) / (0.2 * L + 1))
    idx = np.array([(0, -1, 1, 2)[b % 4] for b in range(B)], np.int32)
    want = on.ir_convolve(x, bank, lens, idx)
    got = ops.ir_convolve(torch.from_numpy(x).to(dev), torch.from_numpy(bank).to(dev), torch.from_numpy(lens),
                          torch.from_numpy(idx)).cpu().numpy()
    assert np.array_equal(got, want)          # (== treats +0 and -0 alike: zero-padded taps may flip a zero's sign)
    ref = np.convolve(x[0].astype(np.float64), bank[idx[0], :lens[idx[0]]].astype(np.float64))[:T]
    np.testing.assert_allclose(got[0], ref, rtol=0, atol=3e-5 * np.abs(ref).max())


@pytest.mark.parametrize("B,T", [(5, 16000), (1, 100000), (3, 4097)])
def test_mix_snr_vs_oracle(dev, B, T):
    """AddBackgroundNoise on the device vs the oracle (double sums): 2e-6 of the signal's peak; the realised SNR
    is the requested one; skipped signals are copies."""
    from grafp_amd import ops
    from oracle import native as on
    rng = np.random.default_rng(T)
    x = rng.standard_normal((B, T)).astype(np.float32) * 0.1
    noise = rng.standard_normal((3, 30000)).astype(np.float32)
    nlen = np.array([30000, 12345, 100], np.int32)
    nidx = np.array([(0, 1, -1, 2, 1)[b % 5] for b in range(B)], np.int32)
    off = np.array([(7, 12344, 0, 99, 5000)[b % 5] for b in range(B)], np.int32)
    snr = np.array([(0.0, 5.0, 10.0, 20.0, 3.3)[b % 5] for b in range(B)], np.float32)
    want = on.mix_snr(x, noise, nlen, nidx, off, snr)
    got = ops.mix_snr(torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev), torch.from_numpy(nlen),
                      torch.from_numpy(nidx), torch.from_numpy(off), torch.from_numpy(snr)).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6 * np.abs(want).max())
    for b in range(B):
        if nidx[b] < 0:
            assert np.array_equal(got[b], x[b])
        else:
            added = got[b].astype(np.float64) - x[b]
            got_snr = 20 * np.log10(np.sqrt((x[b].astype(np.float64) ** 2).mean()) / np.sqrt((added ** 2).mean()))
            assert abs(got_snr - snr[b]) < 2e-3


def test_device_augmentation_in_the_transform(dev):
    """GPUTransformNeuralfp with recordings: the train branch augments the second view on the device (both
    transforms, per-clip draws, reproducible from the seed), probabilities 0 leave the view untouched, the
    validation branch augments the whole track before segmentation."""
    from grafp_amd.modules.transformations import GPUTransformNeuralfp
    from grafp_amd.util import load_config
    from grafp_amd import ops
    cfg = dict(load_config())
    rng = np.random.default_rng(0)
    irs = (rng.standard_normal((4, 4000)) * np.exp(-np.arange(4000) / 600.0)).astype(np.float32)
    noise = rng.standard_normal((3, 50000)).astype(np.float32)
    x = torch.from_numpy(rng.standard_normal((6, 16000)).astype(np.float32) * 0.1).to(dev)
    t1 = GPUTransformNeuralfp(dict(cfg, aug_seed=11), irs, noise, train=True).to(dev)
    t2 = GPUTransformNeuralfp(dict(cfg, aug_seed=11), irs, noise, train=True).to(dev)
    a, b = t1.train_transform(x), t2.train_transform(x)
    assert torch.equal(a, b) and not torch.equal(a, x) and torch.isfinite(a).all()
    Xi, Xj = t1(x, x)
    assert Xi.shape == Xj.shape == (6, cfg["n_mels"], cfg["n_frames"]) and not torch.equal(Xi, Xj)
    assert torch.equal(Xi, ops.logmel(x, cfg["fs"], cfg["n_fft"], cfg["win_len"], cfg["hop_len"], cfg["n_mels"]))
    off = GPUTransformNeuralfp(dict(cfg, ir_prob=0.0, noise_prob=0.0), irs, noise, train=True).to(dev)
    assert torch.equal(off.train_transform(x), x)
    val = GPUTransformNeuralfp(dict(cfg, aug_seed=3), irs, noise, train=False).to(dev)
    track = x.reshape(-1)
    S_i, S_j = val(track.unsqueeze(0), track.unsqueeze(0))
    assert S_i.shape == S_j.shape and S_i.shape[0] > 1 and not torch.equal(S_i, S_j)


def test_augmentation_ragged_banks_equal_padded(dev):
    """Recordings stored back to back (flat buffer + starts + lengths, what load_bank produces) give the same bits as
    the zero-padded (n, Lmax) rows, for both transforms, and match the oracle's ragged form."""
    from grafp_amd import ops
    from oracle import native as on
    rng = np.random.default_rng(7)
    lens = np.array([1500, 37, 9000], np.int32)
    recs = [rng.standard_normal(n).astype(np.float32) * 0.2 for n in lens]
    flat = np.concatenate(recs); starts = np.concatenate([[0], np.cumsum(lens[:-1])]).astype(np.int64)
    padded = np.zeros((3, 9000), np.float32)
    for i, r in enumerate(recs):
        padded[i, :lens[i]] = r
    x = rng.standard_normal((4, 12000)).astype(np.float32)
    idx = np.array([2, 0, -1, 1], np.int32); off = np.array([8999, 3, 0, 36], np.int32)
    snr = np.array([3.0, 10.0, 0.0, 20.0], np.float32)
    T = torch.from_numpy
    xd, fd, pd = T(x).to(dev), T(flat).to(dev), T(padded).to(dev)
    a = ops.ir_convolve(xd, fd, T(lens), T(idx), T(starts)); b = ops.ir_convolve(xd, pd, T(lens), T(idx))
    assert torch.equal(a, b) and np.array_equal(a.cpu().numpy(), on.ir_convolve(x, flat, lens, idx, starts))
    c = ops.mix_snr(xd, fd, T(lens), T(idx), T(off), T(snr), T(starts)); d = ops.mix_snr(xd, pd, T(lens), T(idx), T(off), T(snr))
    assert torch.equal(c, d)
    np.testing.assert_allclose(c.cpu().numpy(), on.mix_snr(x, flat, lens, idx, off, snr, starts), rtol=0, atol=2e-6 * np.abs(x).max())


def test_bn_single_pass_never_depends_on_absent_row_mates(dev, monkeypatch):
    """The rendezvous of bn_fwd1 / bn_bwd1 is bounded and falls back to recomputing the missing partial sums (it used
    to trap): (1) with a spin limit of 0 every workgroup fills in whatever is not yet published -- results must be the
    same BITS as the normal run; (2) the same while a second stream holds 15/16 of the chip's wave slots for tens of
    milliseconds (stand-in for collective kernels occupying CUs during backward), with a short spin limit."""
    from grafp_amd import ops
    from grafp_amd._lib import check, lib
    C, M = 64, 2 * 256 * 1024
    gen = torch.Generator(device=dev).manual_seed(5)
    x = (torch.randn(C, M, device=dev, generator=gen) * 2 + 1).to(torch.bfloat16)
    dz = torch.randn(C, M, device=dev, generator=gen).to(torch.bfloat16)
    gamma = torch.rand(C, device=dev, generator=gen) + 0.5
    beta = torch.randn(C, device=dev, generator=gen)

    def run():
        xg = x.clone().requires_grad_(True)
        g = gamma.clone().requires_grad_(True)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        z = ops.bn_act(xg, g, beta, rm, rv, True, act=ops.ACT_RELU, groups=2)
        z.backward(dz)
        torch.cuda.synchronize()
        return z.detach(), xg.grad, g.grad, rm

    want = run()
    monkeypatch.setattr(ops.switches, "bn_spin_limit", 0)          # the spin_limit argument of grafp_bn_*_1pass
    got = run()
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    monkeypatch.setattr(ops.switches, "bn_spin_limit", 16)
    side = torch.cuda.Stream()
    for blocks in (256, 480):
        check(lib.grafp_debug_occupy(blocks, 1024, 60_000_000, ctypes.c_void_p(side.cuda_stream)), "occupy")
        got = run()
        side.synchronize()
        for a, b in zip(got, want):
            assert torch.equal(a, b)


# =============================================================== IVF-PQ parity index
def test_ivfpq_index_vs_oracle_and_exact(dev):
    """grafp_amd.ivfpq.IVFPQIndex -- k-means, encoding, probe and the fused scan + top-k all in csrc/ivfpq.hip -- against
    oracle/csrc/ivfpq.c BIT FOR BIT (trained quantisers, list ids, codes, probed lists, result ids and distances), against
    the numpy (float64) restatement of the published definition within rounding, and against the exact index
    statistically: recall of the true nearest neighbour among its top-20."""
    from grafp_amd.ivfpq import IVFPQIndex, kmeans_init_rows
    from grafp_amd.ops import FlatL2Index
    from oracle import ivfpq as oq
    from oracle import native
    rng = np.random.RandomState(3)
    n, d = 6000, 128
    base = rng.randn(40, d).astype(np.float32)
    x = (base[rng.randint(0, 40, n)] + 0.35 * rng.randn(n, d)).astype(np.float32)       # clustered, like fingerprints
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    idx = IVFPQIndex(d, nlist=16, M=64, niter=6)
    idx.train(x)
    idx2 = IVFPQIndex(d, nlist=16, M=64, niter=6)
    idx2.train(x)
    assert torch.equal(idx.centroids, idx2.centroids) and torch.equal(idx.codebooks, idx2.codebooks)   # seeded
    # training == the C restatement of the same seeded Lloyd iterations, bit for bit
    cent_o = native.kmeans(x, 1, 16, kmeans_init_rows(n, 16, idx.seed).numpy(), 6)[0]
    assert np.array_equal(idx.centroids.cpu().numpy(), cent_o)
    a_tr = native.pq_assign(x, 1, cent_o[None])[:, 0]
    books_o = native.kmeans(x, 64, 256, kmeans_init_rows(n, 256, idx.seed + 2).numpy(), 6, base=cent_o, base_idx=a_tr)
    assert np.array_equal(idx.codebooks.cpu().numpy(), books_o)
    idx.add(x[:2500]); idx.add(x[2500:])
    idx.nprobe = 5
    cent, books = cent_o, books_o
    a_g, codes_g = torch.cat(idx._assign).cpu().numpy(), torch.cat(idx._codes).cpu().numpy()
    assert np.array_equal(a_g, a_tr)                                                    # coarse assignment: C oracle
    assert np.array_equal(codes_g, native.pq_assign(x, 64, books, base=cent, base_idx=a_tr).astype(np.uint8))
    a_o, codes_o = oq.encode(x, cent, books)                                            # float64 definition
    assert (a_g == a_o).mean() > 0.999 and (codes_g == codes_o).mean() > 0.999          # f32 vs f64 near-ties only
    q = (x[::60] + 0.05 * rng.randn(100, d)).astype(np.float32)
    D, I = idx.search(q, 20)
    order = np.argsort(a_g, kind="stable")
    start = np.r_[0, np.cumsum(np.bincount(a_g, minlength=16))]
    probe_o = native.ivfpq_probe(q, cent, 5)
    Dc, Ic = native.ivfpq_search(q, cent, books, codes_g[order], start, order, probe_o, 20)
    assert np.array_equal(I, Ic) and np.array_equal(D, Dc)                               # search: C oracle, bit for bit
    Do, Io = oq.search(q, a_g, codes_g, cent, books, 5, 20)                               # float64 definition
    np.testing.assert_allclose(D, Do, rtol=2e-4, atol=2e-5)
    assert (I == Io).mean() > 0.99
    # duplicates: equal estimates come back lowest id first; fewer candidates than k: (+inf, -1) padding
    dup = IVFPQIndex(d, nlist=4, M=64, niter=2)
    dup.train(x[:600])
    dup.add(np.tile(x[:7], (5, 1)))
    dup.nprobe = 4
    Dd, Id = dup.search(x[:3], 32)
    for r in range(3):
        assert list(Id[r, :5]) == [r, r + 7, r + 14, r + 21, r + 28] and (Dd[r, :5] == Dd[r, 0]).all()
    few = IVFPQIndex(d, nlist=4, M=64, niter=2)
    few.train(x[:600])
    few.add(x[:3])
    few.nprobe = 4
    Df, If = few.search(x[:2], 10)
    assert (If[:, 3:] == -1).all() and np.isinf(Df[:, 3:]).all() and sorted(If[0, :3]) == [0, 1, 2]
    # k beyond the fused kernel's register list (the reference's k_probe is a free argument, eval.py:177): the dense scan
    # + selection route returns the C oracle's lists too, and its first 20 columns are the fused kernel's
    D50, I50 = idx.search(q, 50)
    Dc50, Ic50 = native.ivfpq_search(q, cent, books, codes_g[order], start, order, probe_o, 50)
    assert np.array_equal(I50, Ic50) and np.array_equal(D50, Dc50)
    assert np.array_equal(I50[:, :20], I) and np.array_equal(D50[:, :20], D)
    Df40, If40 = few.search(x[:2], 40)                                                   # fewer candidates than k, dense route
    assert (If40[:, 3:] == -1).all() and np.isinf(Df40[:, 3:]).all() and np.array_equal(If40[:, :3], If[:, :3])
    Dd40, Id40 = dup.search(x[:3], 35)                                                   # ties: lowest id first there too
    assert np.array_equal(Id40[:, :32], Id) and np.array_equal(Dd40[:, :32], Dd) and (Id40[:, 35 - 1] >= -1).all()
    with pytest.raises(ValueError):
        idx.search(q, 0)
    # torch tensors in, tensors out
    Dt, It = idx.search(torch.from_numpy(q[:3]).to(dev), 4)
    assert Dt.is_cuda and torch.equal(It.cpu(), torch.from_numpy(I[:3, :4]))
    # against the exact index: the planted neighbour is found
    ex = FlatL2Index(d)
    ex.add(x)
    _, Ie = ex.search(q, 1)
    assert (I == Ie[:, :1]).any(axis=1).mean() > 0.9
    idx.nprobe = 16                                                                       # all lists: PQ error only
    _, Iall = idx.search(q, 20)
    assert (Iall == Ie[:, :1]).any(axis=1).mean() > 0.97
    assert idx.rows().shape == (n, d) and idx.ntotal == n


@pytest.mark.parametrize("d,M,nlist,n", [(64, 64, 8, 1500), (32, 8, 5, 900), (128, 16, 3, 100), (16, 16, 70, 400),
                                         (128, 64, 512, 3000)])
def test_ivfpq_other_shapes_vs_c_oracle(dev, d, M, nlist, n):
    """Sub-space widths 1 / 4 / 8 (the table and scan templates), M not a multiple of 16 (byte-wise code reads), fewer
    training rows than codewords (repeated seeds, empty clusters keep their centroid), more lists than a wave has lanes,
    and 512 lists at d = 128 (eval.py's n_centroids argument: the coarse centroids pass through LDS in tiles, 256 KB do not
    fit at once): quantisers, codes, probes and search results bit-equal to oracle/csrc/ivfpq.c."""
    from grafp_amd.ivfpq import IVFPQIndex, kmeans_init_rows
    from oracle import native
    rng = np.random.RandomState(d + M)
    x = rng.randn(n, d).astype(np.float32)
    idx = IVFPQIndex(d, nlist=nlist, M=M, niter=3)
    idx.train(x)
    cent = native.kmeans(x, 1, nlist, kmeans_init_rows(n, nlist, idx.seed).numpy(), 3)[0]
    a = native.pq_assign(x, 1, cent[None])[:, 0]
    books = native.kmeans(x, M, 256, kmeans_init_rows(n, 256, idx.seed + 2).numpy(), 3, base=cent, base_idx=a)
    assert np.array_equal(idx.centroids.cpu().numpy(), cent) and np.array_equal(idx.codebooks.cpu().numpy(), books)
    idx.add(x)
    codes = native.pq_assign(x, M, books, base=cent, base_idx=a).astype(np.uint8)
    assert np.array_equal(torch.cat(idx._codes).cpu().numpy(), codes)
    nprobe = min(nlist, 4)
    idx.nprobe = nprobe
    q = (x[:23] + 0.02 * rng.randn(23, d)).astype(np.float32)
    D, I = idx.search(q, 9)
    order = np.argsort(a, kind="stable")
    start = np.r_[0, np.cumsum(np.bincount(a, minlength=nlist))]
    probe = native.ivfpq_probe(q, cent, nprobe)
    Dc, Ic = native.ivfpq_search(q, cent, books, codes[order], start, order, probe, 9)
    assert np.array_equal(I, Ic) and np.array_equal(D, Dc)


def test_ivfpq_dense_scan_agrees_with_the_fused_search(dev):
    """grafp_ivfpq_scan_f32 (every estimate of the probed lists out, the round-1 form) followed by a host-side selection
    gives the fused kernel's results: the two share the table and the order of the sub-space sum."""
    import ctypes
    from grafp_amd._lib import check, lib
    from grafp_amd.ivfpq import IVFPQIndex
    rng = np.random.RandomState(5)
    n, d = 3000, 128
    x = rng.randn(n, d).astype(np.float32)
    idx = IVFPQIndex(d, nlist=8, M=64, niter=3)
    idx.train(x)
    idx.add(x)
    idx.nprobe = 3
    q = torch.from_numpy(x[:17] + 0.01).to(dev)
    D, I = idx.search(q, 10)
    codes, ids, start, counts = idx._materialise()
    probe = torch.empty((17, 3), dtype=torch.int32, device=dev)
    vp = ctypes.c_void_p
    st = vp(torch.cuda.current_stream().cuda_stream)
    check(lib.grafp_ivfpq_probe_f32(vp(q.data_ptr()), 17, d, vp(idx.centroids.data_ptr()), 8, 3, vp(probe.data_ptr()), st),
          "probe")
    lens = counts[probe.long()]
    ostart = (torch.cumsum(lens, 1) - lens).contiguous()
    stride = int(lens.sum(1).max().item())
    dist = torch.full((17, stride), float("inf"), device=dev)
    pos = torch.full((17, stride), -1, dtype=torch.int32, device=dev)
    check(lib.grafp_ivfpq_scan_f32(vp(q.data_ptr()), 17, d, vp(idx.centroids.data_ptr()), 8, vp(idx.codebooks.data_ptr()), 64,
                                   vp(codes.data_ptr()), vp(start.data_ptr()), vp(probe.data_ptr()), 3, vp(ostart.data_ptr()),
                                   stride, vp(dist.data_ptr()), vp(pos.data_ptr()), st), "scan")
    dist, pos, ids = dist.cpu().numpy(), pos.cpu().numpy(), ids.cpu().numpy()
    for r in range(17):
        ok = pos[r] >= 0
        cand_i, cand_d = ids[pos[r][ok]], dist[r][ok]
        sel = np.lexsort((cand_i, cand_d))[:10]
        assert np.array_equal(cand_i[sel], I[r].cpu().numpy()) and np.array_equal(cand_d[sel], D[r].cpu().numpy())


def test_eval_faiss_with_the_ivfpq_index(dev, tmp_path, monkeypatch):
    """eval_faiss(index_type='ivfpq') -- the default of the reference's test_fp.py -- runs the protocol's index (64
    lists, 64 x 8-bit codes, nprobe 20) and its hit-rate table stays close to the exact index's on the golden case."""
    from _common import eval_case, golden, write_eval_case
    from grafp_amd.eval import eval_faiss
    g = golden("eval_faiss.npz")
    case = eval_case()
    write_eval_case(str(tmp_path), case)
    np.save(tmp_path / "ids.npy", case["test_ids"])
    kw = dict(test_ids=str(tmp_path / "ids.npy"), test_seq_len=case["test_seq_len"], k_probe=case["k_probe"])
    approx = eval_faiss(str(tmp_path), index_type="ivfpq", n_centroids=8, **kw)
    assert approx.shape == g["hit_rates"].shape
    assert np.abs(approx - g["hit_rates"]).max() <= 10.0 and approx[3].min() >= g["hit_rates"][3].min() - 10.0   # 40 ids: 2.5 pt each
    import grafp_amd.eval as geval
    monkeypatch.setattr(geval, "SERVE_IVFPQ_EXACTLY", True)
    np.testing.assert_array_equal(eval_faiss(str(tmp_path), index_type="ivfpq", **kw), g["hit_rates"])


# =============================================================== optimizer update (csrc/adam.hip)
def test_adam_multi_tensor_vs_torch(dev):
    """grafp_amd.optim.Adam (one hand-written multi-tensor kernel per 64 tensors; /root/reference/train.py:79,174) against
    torch.optim.Adam (the single-tensor f32 implementation) from the same state over six steps: parameters, both moments
    and the step counters; 70 tensors (two launches) with sizes around the 4096-element workgroup, an UNALIGNED gradient
    view, a learning rate that lives on the device and changes between steps, a parameter that gets no gradient in some
    steps (skipped, its counter stands still), and state_dict round trips in both directions."""
    from grafp_amd.optim import Adam
    g = torch.Generator().manual_seed(11)
    sizes = [1, 3, 4, 64, 127, 4095, 4096, 4097, 8192 + 5, 64 * 64, 3 * 7 * 7 * 8, 300_000] + [257 + 13 * i for i in range(58)]
    init = [torch.randn(n, generator=g) for n in sizes]
    mine = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    ref = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    frozen = torch.nn.Parameter(torch.ones(5, device=dev), requires_grad=False)
    lr = torch.tensor(3e-3, device=dev)
    a = Adam(mine + [frozen], lr=lr)
    b = torch.optim.Adam(ref + [frozen], lr=3e-3, foreach=False, fused=False)
    flat = torch.empty(sum(sizes) + 1, device=dev)
    for step in range(6):
        off = 1                                                      # every gradient of `mine` is a 4-byte-aligned view
        for i, (p, q) in enumerate(zip(mine, ref)):
            gr = (torch.randn(p.numel(), generator=g) * (10.0 ** ((i % 5) - 2))).to(dev)
            if i == 7 and step in (2, 3):
                p.grad = q.grad = None
                continue
            flat[off:off + p.numel()] = gr
            p.grad = flat[off:off + p.numel()].view_as(p)
            q.grad = gr.clone()
            off += p.numel()
        if step == 4:
            lr.fill_(1e-3)
            for grp in b.param_groups:
                grp["lr"] = 1e-3
        a.step()
        b.step()
        for i, (p, q) in enumerate(zip(mine, ref)):
            scale = float(q.detach().abs().max()) + 1e-12
            assert float((p.detach() - q.detach()).abs().max()) <= 2e-6 * scale + 1e-9, (step, i)
            sa, sb = a.state[p], b.state[q]
            assert float(sa["step"]) == float(sb["step"]), (step, i)
            for k in ("exp_avg", "exp_avg_sq"):
                sk = float(sb[k].abs().max()) + 1e-30                      # (entries near a cancellation: per-tensor scale)
                assert float((sa[k] - sb[k]).abs().max()) <= 2e-6 * sk, (step, i, k)
    assert float(a.state[mine[7]]["step"]) == 4.0 and float(a.state[mine[0]]["step"]) == 6.0
    assert frozen not in a.state or "step" not in a.state[frozen]
    # state_dict: torch's class loads ours and continues identically; ours loads torch's
    c = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in mine] + [frozen], lr=1e-3, foreach=False, fused=False)
    import copy
    sd = copy.deepcopy(a.state_dict())           # (load_state_dict keeps the tensors it is given when no cast is needed)
    sd["param_groups"] = [dict(grp, lr=1e-3) for grp in sd["param_groups"]]
    c.load_state_dict(sd)
    d = Adam([torch.nn.Parameter(q.detach().clone()) for q in ref] + [frozen], lr=lr)
    d.load_state_dict(copy.deepcopy(b.state_dict()))
    for grp in d.param_groups:
        grp["lr"] = lr
    for opt in (a, c, d, b):
        for i, p in enumerate(opt.param_groups[0]["params"][:-1]):
            p.grad = torch.full_like(p, 0.01 * (i + 1))
        opt.step()
    for pa, pc, pd, pb in zip(*(o.param_groups[0]["params"][:-1] for o in (a, c, d, b))):
        scale = float(pb.detach().abs().max()) + 1e-12
        assert float((pa.detach() - pc.detach()).abs().max()) <= 2e-6 * scale + 1e-9
        assert float((pd.detach() - pb.detach()).abs().max()) <= 2e-6 * scale + 1e-9
    assert float(d.state[d.param_groups[0]["params"][0]]["step"]) == 7.0
    with pytest.raises(NotImplementedError):
        Adam(mine, lr=1e-3, weight_decay=0.1)
    with pytest.raises(RuntimeError):
        Adam([torch.nn.Parameter(torch.zeros(4))], lr=1e-3).step()          # a CPU parameter: no fallback
