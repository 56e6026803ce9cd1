"""CPU tests of the host-side logic that needs no kernel: module surface, state-dict schema, drop-in aliases,
dense helpers vs the oracle, rerank logic, on-disk formats, configuration."""
import os

import numpy as np
import pytest
import torch

from _common import ROOT, filled_state_dict, hash_normalish, manifest_shapes


def _model(B=4):
    from grafp_amd.train import build_model
    from grafp_amd.util import load_config
    cfg = load_config()
    cfg["bsz_train"] = B
    return cfg, build_model(cfg)


def test_state_dict_schema_matches_reference_manifest():
    cfg, model = _model()
    shapes = manifest_shapes()
    sd = model.state_dict()
    assert list(sd.keys()) == list(shapes.keys())                      # same 443 keys, same order
    assert all(tuple(v.shape) == shapes[k] for k, v in sd.items())
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 18367264
    assert sum(p.numel() for p in model.parameters() if not p.requires_grad) == 2253312   # frozen relative_pos
    assert len(model.encoder.backbone) == 15 and hasattr(model.encoder, "stem") and hasattr(model.encoder, "proj")
    # a reference-style checkpoint (with or without DataParallel's 'module.' prefix) loads
    from grafp_amd.util import strip_module_prefix
    filled = filled_state_dict()
    full = dict(sd)
    full.update(filled)
    model.load_state_dict(full)
    model.load_state_dict(strip_module_prefix({"module." + k: v for k, v in full.items()}))


def test_constructor_surface():
    import inspect
    from grafp_amd.encoder.graph_encoder import GraphEncoder
    from grafp_amd.modules.transformations import GPUTransformNeuralfp
    from grafp_amd.simclr.simclr import SimCLR
    p = inspect.signature(GraphEncoder.__init__).parameters
    assert [k for k in p][1:] == ["cfg", "k", "conv", "act", "norm", "bias", "dropout", "dilation", "epsilon", "drop_path",
                                  "size", "emb_dims", "in_channels"]
    assert p["k"].default == 3 and p["size"].default == "t" and p["in_channels"].default == 3
    assert [k for k in inspect.signature(SimCLR.__init__).parameters][1:] == ["cfg", "encoder"]
    assert [k for k in inspect.signature(GPUTransformNeuralfp.__init__).parameters][1:] == [
        "cfg", "ir_dir", "noise_dir", "train", "cpu", "abl"]
    from grafp_amd.eval import eval_faiss
    ps = inspect.signature(eval_faiss).parameters
    assert list(ps) == ["emb_dir", "emb_dummy_dir", "index_type", "nogpu", "max_train", "test_ids", "test_seq_len",
                        "k_probe", "n_centroids"]
    assert ps["index_type"].default == "ivfpq" and ps["k_probe"].default == 20


def test_dropin_aliases():
    import sys
    import grafp_amd.dropin as d
    saved = {k: sys.modules.get(k) for k in d._ALIASES}
    try:
        d.install(overwrite=True)
        from encoder.graph_encoder import GraphEncoder          # noqa: F401
        from eval import eval_faiss, get_index, load_memmap_data  # noqa: F401
        from generate import create_db                          # noqa: F401
        from modules.transformations import GPUTransformNeuralfp  # noqa: F401
        from simclr.ntxent import ntxent_loss                   # noqa: F401
        from simclr.simclr import SimCLR                        # noqa: F401
        from test_fp import create_dummy_db, create_fp_db       # noqa: F401
        import grafp_amd.encoder.graph_encoder as real
        assert GraphEncoder is real.GraphEncoder
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_dense_helpers_match_conv_library_semantics():
    """conv1x1 / conv3_stride2 (single-GEMM formulation on (C,B,N)) == F.conv2d on (B,C,N,1)."""
    from torch import nn
    from grafp_amd.encoder._dense import conv1x1, conv3_stride2, from_cbn, to_cbn
    torch.manual_seed(0)
    x = torch.randn(3, 16, 40)
    assert to_cbn(x).shape == (16, 3, 40) and torch.equal(from_cbn(to_cbn(x), x), x)
    assert from_cbn(to_cbn(x.unsqueeze(-1)), x.unsqueeze(-1)).shape == (3, 16, 40, 1)
    for groups in (1, 4):
        conv = nn.Conv2d(16, 24, 1, groups=groups, bias=False)
        got = from_cbn(conv1x1(conv, to_cbn(x)), x)
        np.testing.assert_allclose(got.detach().numpy(), conv(x.unsqueeze(-1)).squeeze(-1).detach().numpy(),
                                   rtol=1e-5, atol=1e-5)
    from _common import CpuOps
    for n in (40, 41):
        xs = torch.randn(2, 16, n)
        down = nn.Conv2d(16, 32, 3, stride=2, padding=1, bias=False)
        with CpuOps():                              # the tap gather is a HIP kernel; its CPU stand-in is test-only
            got = from_cbn(conv3_stride2(down, to_cbn(xs)), xs)
        np.testing.assert_allclose(got.detach().numpy(), down(xs.unsqueeze(-1)).squeeze(-1).detach().numpy(),
                                   rtol=1e-5, atol=1e-5)


def test_dense_part_of_the_model_matches_the_oracle_on_cpu():
    """With the HIP ops swapped for torch/oracle equivalents (TEST-ONLY, _common.CpuOps) the module mirror
    reproduces the oracle's forward AND backward: pins the GEMM / fused-BN / residual wiring, the (C,B,N) layout
    handling and the state-dict mapping without a GPU."""
    from _common import CpuOps, simclr_inputs
    from oracle import model as om
    cfg, model = _model()
    filled = filled_state_dict()
    sd = model.state_dict(); sd.update(filled); model.load_state_dict(sd)
    xi, xj = simclr_inputs()
    with CpuOps():
        model.train()
        h_i, h_j, z_i, z_j = model(xi, xj)
        loss = om.ntxent(z_i, z_j, 0.05)
        loss.backward()
    osd = {k: v.clone() for k, v in filled.items()}
    for k, v in osd.items():
        if v.is_floating_point() and k.rsplit(".", 1)[-1] not in ("running_mean", "running_var"):
            v.requires_grad_(True)
    o = om.simclr_forward(osd, xi, xj, True)
    oloss = om.ntxent(o[2], o[3], 0.05)
    oloss.backward()
    np.testing.assert_allclose(z_i.detach().numpy(), o[2].detach().numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(z_j.detach().numpy(), o[3].detach().numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(h_i.detach().numpy(), o[0].detach().numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(loss.item(), oloss.item(), rtol=1e-5)
    sdm = model.state_dict()
    for k in ("encoder.stem.1.running_mean", "encoder.backbone.7.1.fc1.1.running_var", "encoder.backbone.3.0.fc2.1.running_mean"):
        np.testing.assert_allclose(sdm[k].numpy(), osd[k].detach().numpy(), rtol=1e-4, atol=1e-6)
    assert int(sdm["encoder.stem.1.num_batches_tracked"]) == 2
    P = dict(model.named_parameters())
    gn = max(float(osd[k].grad.norm()) for k in P if P[k].requires_grad)
    rel = [float((P[k].grad - osd[k].grad).norm()) / max(float(osd[k].grad.norm()), 1e-4 * gn)
           for k in P if P[k].requires_grad]
    assert max(rel) < 5e-2 and np.median(rel) < 1e-2, (max(rel), np.median(rel))


def test_sequence_rerank_matches_oracle():
    from grafp_amd.eval import sequence_rerank
    from oracle import retrieval
    rng = np.random.default_rng(0)
    recon = rng.standard_normal((400, 128)).astype(np.float32)
    for sl in (1, 3, 11):
        q = recon[100:100 + sl] + 0.1 * rng.standard_normal((sl, 128)).astype(np.float32)
        I = rng.integers(0, 400, size=(sl, 20))
        I[:, 0] = 100 + np.arange(sl)
        I[0, 5] = -1
        I[-1, 3] = 399                                                   # candidate sequence runs off the end
        pred = sequence_rerank(q, I.copy(), recon, sl)
        J = I - np.arange(sl)[:, None]
        cand = np.unique(J[J >= 0])
        scores = retrieval.sequence_scores(q, recon, cand, sl)
        want = cand[np.argsort(-scores, kind="stable")[:10]]
        assert pred[0] == 100 and np.array_equal(pred, want)


def test_memmap_format_roundtrip(tmp_path):
    from grafp_amd.eval import load_memmap_data
    from grafp_amd.fpdb import _write_memmap
    arr = hash_normalish("host:mm", (37, 128))
    arr[3, 5] = np.nan
    _write_memmap(str(tmp_path / "db"), arr)
    assert os.path.getsize(tmp_path / "db.mm") == 37 * 128 * 4
    assert tuple(np.load(tmp_path / "db_shape.npy")) == (37, 128)
    data, shape = load_memmap_data(str(tmp_path), "db", display=False)
    assert tuple(shape) == (37, 128) and data[3, 5] == 0.0               # NaN -> 0 (eval.py:165)
    ref = arr.copy(); ref[3, 5] = 0.0
    np.testing.assert_array_equal(np.asarray(data), ref)
    assert tuple(load_memmap_data(str(tmp_path), "db", shape_only=True)) == (37, 128)


def test_config_and_util():
    from grafp_amd import util
    cfg = util.load_config()
    ref = dict(fs=16000, n_fft=1024, win_len=1024, hop_len=512, n_mels=64, n_frames=32, peak_stride=2, n_filters=8,
               d=128, h=1024, u=32, tau=0.05, bsz_train=256, overlap=0.9, blur_kernel=[7, 7], arch="grafp")
    assert all(cfg[k] == v for k, v in ref.items())
    assert [util.query_len_from_seconds(s, 0.9, 1.0) for s in (1, 2, 3, 5)] == [1, 10, 20, 40] or \
        [util.query_len_from_seconds(s, 0.9, 1.0) for s in (1, 2, 3, 5)] == [1, 11, 21, 41]
    assert util.override(3, None) == 3 and util.override(3, 5) == 5


def test_augmentation_banks_and_cpu_branch(tmp_path):
    """Recordings are decoded once into zero-padded banks (.npy and PCM .wav, directory / list / array sources);
    a wrong sample rate is refused; without a GPU the transforms themselves refuse to run (no CPU fallback); the
    DataLoader-worker branch (cpu=True) stays an identity crop."""
    import wave
    from grafp_amd.modules.transformations import GPUTransformNeuralfp, load_bank
    from grafp_amd.util import load_config
    cfg = load_config()
    a = (np.sin(np.arange(300) * 0.1) * 0.5).astype(np.float32)
    np.save(tmp_path / "a.npy", a)
    pcm = (np.clip(np.cos(np.arange(500) * 0.05), -1, 1) * 32767).astype("<i2")
    for name, fs in (("b.wav", cfg["fs"]), ("bad.wav", 8000)):
        with wave.open(str(tmp_path / name), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(fs); w.writeframes(pcm.tobytes())
    bank, starts, lens = load_bank([str(tmp_path / "a.npy"), str(tmp_path / "b.wav")], cfg["fs"])
    assert bank.shape == (800,) and lens.tolist() == [300, 500] and starts.tolist() == [0, 300]     # ragged: no padding
    assert np.array_equal(bank[:300].numpy(), a)
    np.testing.assert_allclose(bank[300:].numpy(), pcm.astype(np.float32) / 32768.0)
    with pytest.raises(ValueError):
        load_bank(str(tmp_path / "bad.wav"), cfg["fs"])
    (tmp_path / "d").mkdir(); np.save(tmp_path / "d" / "x.npy", a)
    assert load_bank(str(tmp_path / "d"), cfg["fs"])[2].tolist() == [300]
    b3 = load_bank(np.ones((3, 7), np.float32), cfg["fs"], max_len=5)
    assert b3[0].shape == (15,) and b3[1].tolist() == [0, 5, 10]
    t = GPUTransformNeuralfp(cfg, str(tmp_path / "d"), np.ones((2, 100), np.float32), train=True)
    assert t.ir_bank.shape == (300,) and t.noise_len.tolist() == [100, 100] and t.noise_start.tolist() == [0, 100]
    with pytest.raises(RuntimeError):
        t.train_transform(torch.zeros(2, 16000))           # CPU tensors: the HIP path refuses, nothing falls back
    t = GPUTransformNeuralfp(cfg, None, None, cpu=True)
    a, b = t(torch.zeros(16000), torch.arange(16010.0))
    assert b.shape == (16000,)


_IMPORT_REFERENCE_SCRIPTS = r'''
import importlib.util, os, sys, types
sys.dont_write_bytecode = True
REF, ROOT = "/root/reference", sys.argv[1]
sys.path.insert(0, ROOT)

def stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m; return m
# third-party packages of the reference's requirements.txt that this image lacks: inert stand-ins (decode / logging only)
ta = stub("torchaudio", set_audio_backend=lambda *a, **k: None, load=None)
stub("torchaudio.transforms"); ta.transforms = sys.modules["torchaudio.transforms"]
import torch.utils
stub("torch.utils.tensorboard", SummaryWriter=type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None}))
stub("soundfile"); stub("prettytable", PrettyTable=type("PrettyTable", (), {})); stub("librosa")
assert "faiss" not in sys.modules

sys.path.insert(0, REF)                        # a checkout of the reference: its scripts, util.py, modules/data.py
import grafp_amd.dropin as dropin
names = dropin.install()
assert "faiss" in names and "util" not in names and "modules" not in names, names

def load(name, fname):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, fname))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod); return mod

import grafp_amd.eval, grafp_amd.fpdb, grafp_amd.faiss_standin
import grafp_amd.encoder.graph_encoder as ge, grafp_amd.simclr.simclr as sc, grafp_amd.simclr.ntxent as nx
import grafp_amd.modules.transformations as tf

t = load("ref_test_fp", "test_fp.py")          # module level: imports, argparse definitions; main() is not run
assert t.faiss is grafp_amd.faiss_standin and t.faiss.IndexFlatL2 is grafp_amd.ops.FlatL2Index
assert t.eval_faiss is grafp_amd.eval.eval_faiss and t.get_index is grafp_amd.eval.get_index
assert t.load_memmap_data is grafp_amd.eval.load_memmap_data
assert t.GraphEncoder is ge.GraphEncoder and t.SimCLR is sc.SimCLR and t.GPUTransformNeuralfp is tf.GPUTransformNeuralfp
assert t.load_config.__module__ == "util" and sys.modules["util"].__file__.startswith(REF)      # the reference's own
assert t.NeuralfpDataset.__module__ == "modules.data" and sys.modules["modules.data"].__file__.startswith(REF)
assert sys.modules["modules.transformations"] is tf and sys.modules["modules"].transformations is tf

tr = load("ref_train", "train.py")
assert tr.create_fp_db is grafp_amd.fpdb.create_fp_db and tr.create_dummy_db is grafp_amd.fpdb.create_dummy_db
assert tr.eval_faiss is grafp_amd.eval.eval_faiss and tr.ntxent_loss is nx.ntxent_loss
assert tr.SimCLR is sc.SimCLR and tr.GraphEncoder is ge.GraphEncoder and tr.GPUTransformNeuralfp is tf.GPUTransformNeuralfp

g = load("ref_generate", "generate.py")
assert g.GraphEncoder is ge.GraphEncoder and g.SimCLR is sc.SimCLR and g.GPUTransformNeuralfp is tf.GPUTransformNeuralfp

# the reference's OWN eval.py against the stand-in: its module-level GPU plumbing (eval.py:42-46) runs on it
e = load("ref_eval", "eval.py")
assert e.faiss is grafp_amd.faiss_standin
import faiss
opts = faiss.GpuClonerOptions(); opts.useFloat16 = True
assert faiss.index_cpu_to_gpu(faiss.StandardGpuResources(), 0, "index", opts) == "index"
print("reference scripts import through the alias route")
'''


def test_reference_scripts_import_through_the_alias_route(tmp_path):
    """with grafp_amd.dropin.install() -- which now also registers a `faiss` stand-in -- the reference's
    train.py, test_fp.py, generate.py (and its own eval.py) IMPORT unchanged from a checkout of the reference, and the
    names they bind are this package's: GraphEncoder, SimCLR, ntxent_loss, GPUTransformNeuralfp, eval_faiss, get_index,
    create_fp_db, create_dummy_db, faiss.IndexFlatL2.  util.py and modules/data.py stay the reference's own host-side
    files.  torchaudio / tensorboard / soundfile / prettytable / librosa (absent from this image) are inert stubs of the
    TEST.  Container-only: nothing of the reference travels to the GPU box."""
    import subprocess
    import sys
    if not os.path.isdir("/root/reference"):
        pytest.skip("the reference checkout is only mounted in the build container")
    script = tmp_path / "import_ref.py"
    script.write_text(_IMPORT_REFERENCE_SCRIPTS)
    res = subprocess.run([sys.executable, str(script), ROOT], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         timeout=600, cwd=str(tmp_path))
    assert res.returncode == 0 and "import through the alias route" in res.stdout, res.stdout[-3000:]


def test_optimizer_refuses_what_the_path_does_not_use():
    """grafp_amd.optim.Adam is torch.optim.Adam with step() on the multi-tensor kernel for what train.py:174 uses (the
    defaults); the options outside the path raise at construction instead of silently running another update, and the
    object keeps torch's state_dict schema (param_groups keys) so that a checkpoint of either class loads into the other."""
    import torch
    from grafp_amd.optim import Adam
    w = torch.nn.Parameter(torch.zeros(4))
    for bad in ({"weight_decay": 1e-2}, {"amsgrad": True}, {"maximize": True}):
        with pytest.raises(NotImplementedError):
            Adam([w], lr=1e-3, **bad)
    opt = Adam([w], lr=3e-4)
    ref = torch.optim.Adam([torch.nn.Parameter(torch.zeros(4))], lr=3e-4)
    assert isinstance(opt, torch.optim.Adam)
    keys = set(ref.state_dict()["param_groups"][0])
    assert set(opt.state_dict()["param_groups"][0]) == keys
    g, h = opt.param_groups[0], ref.param_groups[0]
    assert (g["lr"], g["betas"], g["eps"], g["weight_decay"], g["amsgrad"]) == (h["lr"], h["betas"], h["eps"], 0, False)
    w.grad = torch.ones(4)
    with pytest.raises(RuntimeError):                   # a CPU parameter: there is no CPU fallback behind step()
        opt.step()
