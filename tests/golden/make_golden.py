#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules (TEST INFRASTRUCTURE).

Runs only where /root/reference is mounted (the build container).  The reference is imported
verbatim from there with inert stubs for imports that are dead on the model path (timm DropPath is
never constructed because every drop_path is 0; torchvision/torchmetrics/librosa are imported but
unused: SURVEY.md section 8c / Appendix A).  Nothing from the reference is copied: the fixtures hold
only inputs-by-name (regenerated from tests/_hashfill.py) and expected OUTPUTS.

    python tests/golden/make_golden.py            # rewrites every fixture
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _hashfill import fill_state_dict, hash_ints, hash_normalish, hash_uniform  # noqa: E402

REF = "/root/reference"
sys.dont_write_bytecode = True


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m


class _NeverDropPath(nn.Module):
    def __init__(self, *a, **k):
        raise RuntimeError("DropPath must never be constructed (all drop_path are 0)")


def import_reference():
    _stub("timm"); _stub("timm.models")
    _stub("timm.models.layers", DropPath=_NeverDropPath, to_2tuple=None, trunc_normal_=None)
    _stub("torchvision"); _stub("torchvision.transforms")
    _stub("torchvision.transforms.functional", gaussian_blur=None)
    _stub("torchmetrics"); _stub("torchmetrics.functional", pairwise_cosine_similarity=None)
    _stub("librosa")
    sys.path.insert(0, REF)
    from encoder.gcn_lib import torch_edge, torch_nn, torch_vertex
    from encoder.graph_encoder import FFN, GraphEncoder
    from peak_extractor import GPUPeakExtractorv2
    from simclr.ntxent import ntxent_loss
    from simclr.simclr import SimCLR
    return dict(torch_edge=torch_edge, torch_nn=torch_nn, torch_vertex=torch_vertex, FFN=FFN,
                GraphEncoder=GraphEncoder, GPUPeakExtractorv2=GPUPeakExtractorv2,
                ntxent_loss=ntxent_loss, SimCLR=SimCLR)


def ref_cfg(bsz):
    cfg = yaml.safe_load(open(os.path.join(REF, "config/grafp.yaml")))
    cfg["bsz_train"] = bsz
    return cfg


def load_filled(module, prefix="w"):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    filled = fill_state_dict(shapes, prefix)
    sd = module.state_dict()
    for k, v in filled.items():
        sd[k] = torch.from_numpy(np.asarray(v)).reshape(sd[k].shape).to(sd[k].dtype)
    module.load_state_dict(sd)
    return shapes


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: " + ", ".join(f"{k}{tuple(np.shape(v))}" for k, v in arrs.items()))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    R = import_reference()
    te, tn, tv = R["torch_edge"], R["torch_nn"], R["torch_vertex"]

    # ---- 1. peak extractor (peak_extractor.py:56-82) ------------------------------------------
    cfg = ref_cfg(2)
    pe = R["GPUPeakExtractorv2"](cfg)
    load_filled(pe, "pe")
    spec = 40.0 * hash_uniform("in:peak.spec", (2, 64, 32)) - 30.0       # dB-like range
    with torch.no_grad():
        out = pe(t(spec))
    # batch != bsz_train exercises the except-branch ramps (peak_extractor.py:70-76)
    spec3 = 40.0 * hash_uniform("in:peak.spec3", (3, 64, 32)) - 30.0
    with torch.no_grad():
        out3 = pe(t(spec3))
    save("peak_extractor.npz", out=out.numpy(), out3=out3.numpy())

    # ---- 2. k-NN graph (torch_edge.py:70-103, 270-284) ------------------------------------------
    knn = {}
    for (C, N) in [(64, 1024), (128, 512), (256, 256), (512, 128), (24, 100)]:
        # (a) integer-valued features straight into dense_knn_matrix: every f32 product/sum is exact
        xi = hash_ints(f"in:knn.int.{C}.{N}", (2, C, N, 1), -8, 8).astype(np.float32)
        e = te.dense_knn_matrix(t(xi), k=3)
        assert e.shape == (2, 2, N, 3)
        assert bool((e[1] == torch.arange(N).view(1, N, 1)).all())        # centre index is arange
        knn[f"int_{C}_{N}"] = e[0].numpy().astype(np.int32)
        # (b) float features through DenseDilatedKnnGraph (normalise + knn + dilation slice)
        xf = hash_normalish(f"in:knn.f32.{C}.{N}", (2, C, N, 1))
        g = te.DenseDilatedKnnGraph(3, 1, False, 0.2)
        ef = g(t(xf))
        knn[f"f32_{C}_{N}"] = ef[0].numpy().astype(np.int32)
    # k = 5 variant
    xf = hash_normalish("in:knn.f32.k5", (2, 32, 200, 1))
    knn["f32_k5"] = te.DenseDilatedKnnGraph(5, 1, False, 0.2)(t(xf))[0].numpy().astype(np.int32)
    save("knn_graph.npz", **knn)

    # ---- 3. gather + MRConv (torch_nn.py:79-98, torch_vertex.py:19-34) --------------------------
    B, C, N, K = 2, 8, 64, 3
    x = hash_normalish("in:mr.x", (B, C, N, 1))
    idx = hash_ints("in:mr.idx", (B, N, K), 0, N - 1).astype(np.int64)
    idx[:, :, 0] = np.arange(N)[None, :]                                  # self first, as in the model
    center = np.broadcast_to(np.arange(N, dtype=np.int64)[None, :, None], (B, N, K)).copy()
    sel = tn.batched_index_select(t(x), t(idx))
    mr = tv.MRConv2d(C, 2 * C, "relu", "batch", True)
    load_filled(mr, "mr")
    mr.train()
    xt = t(x).clone().requires_grad_(True)
    edge = torch.stack((t(idx), t(center)), dim=0)
    y = mr(xt, edge)
    gy = t(hash_normalish("in:mr.gy", tuple(y.shape)))
    y.backward(gy)
    # the parameter-free part (what the HIP kernel computes): max-relative + interleave
    xt2 = t(x).clone().requires_grad_(True)
    x_i = tn.batched_index_select(xt2, t(center)); x_j = tn.batched_index_select(xt2, t(idx))
    rel, _ = torch.max(x_j - x_i, -1, keepdim=True)
    inter = torch.cat([xt2.unsqueeze(2), rel.unsqueeze(2)], dim=2).reshape(B, 2 * C, N, 1)
    gi = t(hash_normalish("in:mr.gi", tuple(inter.shape)))
    inter.backward(gi)
    save("mrconv.npz", sel=sel.numpy(), y=y.detach().numpy(), dx=xt.grad.numpy(),
         dw=mr.nn[0].weight.grad.numpy(), inter=inter.detach().numpy(), dx_inter=xt2.grad.numpy())

    # ---- 4. one Grapher + FFN block, train and eval (torch_vertex.py:146-194, graph_encoder.py:45-67)
    C, N = 16, 64
    blk = nn.Sequential(tv.Grapher(C, 3, 1, "mr", "relu", "batch", True, False, 0.2, 1, n=N,
                                   drop_path=0.0, relative_pos=True),
                        R["FFN"](C, 4 * C, C, act="relu", drop_path=0.0))
    shapes = load_filled(blk, "blk")
    xb = hash_normalish("in:blk.x", (3, C, N, 1))
    blk.train()
    y_tr = blk(t(xb)).detach().numpy()
    rm = blk[0].fc1[1].running_mean.numpy().copy()                       # after ONE train forward
    blk.eval()
    with torch.no_grad():
        y_ev = blk(t(xb)).numpy()
    save("block.npz", y_train=y_tr, y_eval=y_ev, fc1_running_mean=rm,
         keys=np.array(sorted(shapes.keys())))

    # ---- 5. full SimCLR forward, B=4 (simclr.py:29-47, graph_encoder.py:167-191) ----------------
    cfg = ref_cfg(4)
    model = R["SimCLR"](cfg, encoder=R["GraphEncoder"](cfg=cfg, in_channels=cfg["n_filters"], k=3))
    shapes = load_filled(model, "w")
    with open(os.path.join(HERE, "state_dict_manifest.txt"), "w") as f:
        for k, v in model.state_dict().items():
            f.write(f"{k} {list(v.shape)} {str(v.dtype).replace('torch.', '')}\n")
    xi = 40.0 * hash_uniform("in:simclr.xi", (4, 64, 32)) - 30.0
    xj = xi + 3.0 * hash_normalish("in:simclr.xj", (4, 64, 32))
    model.train()
    with torch.no_grad():
        h_i, h_j, z_i, z_j = model(t(xi), t(xj))
    bn_mean_after = model.encoder.stem[1].running_mean.numpy().copy()    # two train forwards (i, j)
    model.eval()
    with torch.no_grad():
        eh_i, eh_j, ez_i, ez_j = model(t(xi), t(xj))
    save("simclr_forward.npz", h_i=h_i.numpy(), h_j=h_j.numpy(), z_i=z_i.numpy(), z_j=z_j.numpy(),
         eval_z_i=ez_i.numpy(), eval_z_j=ez_j.numpy(), eval_h_i=eh_i.numpy(),
         stem_running_mean=bn_mean_after)

    # ---- 6. NT-Xent value + grads (ntxent.py:4-29) ---------------------------------------------
    nt = {}
    for Bn in (2, 8, 32):
        zi = hash_normalish(f"in:nt.zi.{Bn}", (Bn, 128)); zj = hash_normalish(f"in:nt.zj.{Bn}", (Bn, 128))
        zi /= np.linalg.norm(zi, axis=1, keepdims=True); zj /= np.linalg.norm(zj, axis=1, keepdims=True)
        a = t(zi).clone().requires_grad_(True); b = t(zj).clone().requires_grad_(True)
        loss = R["ntxent_loss"](a, b, {"tau": 0.05})
        loss.backward()
        nt[f"loss_{Bn}"] = loss.detach().numpy(); nt[f"dzi_{Bn}"] = a.grad.numpy(); nt[f"dzj_{Bn}"] = b.grad.numpy()
        nt[f"zi_{Bn}"] = zi; nt[f"zj_{Bn}"] = zj                          # normalised inputs are stored
    # un-normalised inputs + another tau: the loss must not assume unit rows
    zi = 0.3 * hash_normalish("in:nt.raw.zi", (6, 16)); zj = 0.3 * hash_normalish("in:nt.raw.zj", (6, 16))
    a = t(zi).clone().requires_grad_(True); b = t(zj).clone().requires_grad_(True)
    loss = R["ntxent_loss"](a, b, {"tau": 0.5}); loss.backward()
    nt["loss_raw"] = loss.detach().numpy(); nt["dzi_raw"] = a.grad.numpy(); nt["dzj_raw"] = b.grad.numpy()
    save("ntxent.npz", **nt)

    # ---- 7. one full training step (train.py:60-74): loss, grad norms, post-Adam checksums ------
    cfg = ref_cfg(4)
    model = R["SimCLR"](cfg, encoder=R["GraphEncoder"](cfg=cfg, in_channels=cfg["n_filters"], k=3))
    load_filled(model, "w")
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=8e-5)
    opt.zero_grad()
    _, _, z_i, z_j = model(t(xi), t(xj))
    loss = R["ntxent_loss"](z_i, z_j, cfg)
    loss.backward()
    probe = ["peak_extractor.convs.0.weight", "encoder.stem.0.weight",
             "encoder.backbone.0.0.fc1.0.weight", "encoder.backbone.0.0.graph_conv.gconv.nn.0.weight",
             "encoder.backbone.7.1.fc2.0.weight", "encoder.backbone.12.conv.0.weight",
             "encoder.proj.weight", "projector.2.weight"]
    params = dict(model.named_parameters())
    gnorm = np.array([params[k].grad.double().norm().item() for k in probe])
    gfull = {("grad:" + k): params[k].grad.numpy().copy() for k in probe[:2]}
    opt.step()
    psum = np.array([params[k].detach().double().sum().item() for k in probe])
    save("train_step.npz", loss=loss.detach().numpy(), probe=np.array(probe), grad_norm=gnorm,
         param_sum_after=psum, **gfull)


if __name__ == "__main__":
    main()
