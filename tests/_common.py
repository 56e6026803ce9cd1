"""Shared helpers for the test-suite (inputs by name, state dicts from the committed manifest)."""
import os

import numpy as np
import torch

from _hashfill import fill_state_dict, hash_ints, hash_normalish, hash_uniform  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def golden(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def manifest_shapes():
    shapes = {}
    with open(os.path.join(GOLD, "state_dict_manifest.txt")) as f:
        for line in f:
            name, rest = line.strip().split(" ", 1)
            dims = rest[rest.index("[") + 1:rest.index("]")]
            shapes[name] = tuple(int(s) for s in dims.split(",") if s.strip())
    return shapes


def filled_state_dict(shapes=None, prefix="w", requires_grad=False):
    """Hash-filled torch state dict (CPU, f32) for `shapes` (default: the full 443-key manifest)."""
    shapes = manifest_shapes() if shapes is None else shapes
    sd = {}
    for k, v in fill_state_dict(shapes, prefix).items():
        t = torch.from_numpy(np.array(v)).reshape(shapes[k])
        if requires_grad and t.is_floating_point() and k.rsplit(".", 1)[-1] not in (
                "running_mean", "running_var"):
            t.requires_grad_(True)
        sd[k] = t
    return sd


def simclr_inputs():
    xi = 40.0 * hash_uniform("in:simclr.xi", (4, 64, 32)) - 30.0
    xj = xi + 3.0 * hash_normalish("in:simclr.xj", (4, 64, 32))
    return torch.from_numpy(xi), torch.from_numpy(xj)


def knn_margin_mask(x, idx_k, tol=1e-5, normalize=True):
    """Nodes whose k-th / (k+1)-th neighbour distance gap (float64) exceeds tol, and whose top-k are
    mutually separated by > tol: only there is the neighbour ORDER/SET well defined in f32."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 4:
        x = x[..., 0]
    if normalize:
        x = x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)
    g = np.einsum("bci,bcj->bij", x, x)
    sq = np.einsum("bci,bci->bi", x, x)
    d = sq[:, :, None] - 2 * g + sq[:, None, :]
    ds = np.sort(d, axis=-1)
    k = idx_k
    gaps = ds[..., 1:k + 1] - ds[..., 0:k]                 # gaps between consecutive of the first k+1
    return (gaps > tol).all(-1), d


class RecordedGraphs:
    """Context manager: records every k-NN graph grafp_amd.ops.knn_graph builds (TEST-ONLY monkeypatch), and
    offers an `idx_fn` that replays them, in order, inside the oracle -- so a CPU/GPU comparison holds the edges
    equal.  Upstream f32 rounding differs between CPU and GPU GEMMs, hence a few NEAR-TIE neighbours flip; the
    k-NN decision itself is verified bit-exactly elsewhere on identical inputs."""

    def __enter__(self):
        from grafp_amd import ops
        self._ops, self._orig, self.graphs = ops, ops.knn_graph, []

        def rec(x, k, normalize=True, layout="bcn", index_dtype=torch.int64):
            idx = self._orig(x, k, normalize, layout, index_dtype)
            self.graphs.append(idx.cpu().long())
            return idx
        ops.knn_graph = rec
        return self

    def __exit__(self, *exc):
        self._ops.knn_graph = self._orig

    def sequential(self, views=2):
        """The model stacks its views along the batch (one graph per block over all views); the oracle runs the
        views one after the other: re-order [block][view-stacked] -> view-major list of per-view graphs."""
        if views == 1:
            return list(self.graphs)
        return [g.chunk(views, dim=0)[v].contiguous() for v in range(views) for g in self.graphs]

    def replay_fn(self, views=2):
        it = iter(self.sequential(views))
        self.flips = 0

        def idx_fn(x, k):
            from oracle import native
            idx = next(it)
            own = native.knn_graph(x.detach().numpy(), k)
            self.flips += int((own != idx.numpy()).any(-1).sum())
            return idx
        return idx_fn


class CpuOps:
    """TEST-ONLY: swap the HIP ops for torch/oracle equivalents so the module mirror's WIRING (GEMM formulation,
    layouts, state-dict mapping, BatchNorm bookkeeping) can be checked on a CPU-only machine.  The product never
    does this: its ops refuse CPU tensors."""

    def __enter__(self):
        import torch.nn.functional as F
        from grafp_amd import ops
        from oracle import model as om
        self._ops = ops
        self._saved = (ops.knn_graph, ops.max_relative, ops.peak_extract, ops.bn_act, ops.stride2_taps)

        def bcn(x, layout):
            return x if layout == "bcn" else x.permute(1, 0, 2)

        def knn(x, k, normalize=True, layout="bcn", index_dtype=torch.int64, prefilter=None):
            if x.dim() == 4:
                x = x.squeeze(-1)
            return om.knn_graph_torch(bcn(x, layout).float(), k).to(index_dtype)

        def maxrel(x, idx, layout="bcn"):
            out = om.max_relative(bcn(x, layout), idx.long())
            return out if layout == "bcn" else out.permute(1, 0, 2).contiguous()

        def peak(spec, w, b, s):
            return om.peak_extract({"peak_extractor.convs.0.weight": w, "peak_extractor.convs.0.bias": b}, spec, s)

        def bn_act(x, gamma, beta, rm, rv, training, momentum=0.1, eps=1e-5, pre_bias=None, residual=None, act=0,
                   slope=0.0, groups=1):
            C = x.shape[0]
            y = x.reshape(1, C, -1)
            if pre_bias is not None:
                y = y + pre_bias.reshape(1, C, 1)
            y = torch.cat([F.batch_norm(seg, rm, rv, gamma, beta, training, momentum, eps)
                           for seg in y.chunk(groups, dim=2)], dim=2).reshape(x.shape)
            y = F.relu(y) if act == 1 else (F.leaky_relu(y, slope) if act == 2 else y)
            return y if residual is None else y + residual
        def taps(x):       # pad + three strided slices: what Conv2d(3, stride 2, padding 1) reads along N
            n_out = (x.shape[-1] - 1) // 2 + 1
            xp = F.pad(x, (1, 1))
            return torch.stack([xp[..., t0:t0 + 2 * n_out - 1:2] for t0 in range(3)], dim=0)
        ops.knn_graph, ops.max_relative, ops.peak_extract, ops.bn_act, ops.stride2_taps = knn, maxrel, peak, bn_act, taps
        return self

    def __exit__(self, *exc):
        o = self._ops
        o.knn_graph, o.max_relative, o.peak_extract, o.bn_act, o.stride2_taps = self._saved


def reference_graphs(sd, xi, xj, train):
    """The k-NN graphs the REFERENCE builds for this input (oracle forward with the torch restatement of
    dense_knn_matrix, which reproduces the reference's edges exactly on this host -- tests/test_oracle.py),
    in call order.  `sd` is cloned: running statistics are not disturbed."""
    from oracle import model as om
    graphs = []

    def rec(x, k):
        idx = om.knn_graph_torch(x, k)
        graphs.append(idx)
        return idx
    with torch.no_grad():
        om.simclr_forward({k: v.detach().clone() for k, v in sd.items()}, xi, xj, train, idx_fn=rec)
    return graphs


class ReplayGraphs:
    """TEST-ONLY: make grafp_amd.ops.knn_graph return pre-computed graphs (moved to the input's device).
    `views` > 1: `graphs` is the oracle's view-major list (all blocks of view 0, then view 1, ...); the model stacks
    the views along the batch, so block b gets cat(view0[b], view1[b], ...)."""

    def __init__(self, graphs, views=1):
        if views > 1:
            nb = len(graphs) // views
            graphs = [torch.cat([graphs[v * nb + b] for v in range(views)], dim=0) for b in range(nb)]
        self.graphs = graphs

    def __enter__(self):
        from grafp_amd import ops
        self._ops, self._orig, it = ops, ops.knn_graph, iter(self.graphs)
        ops.knn_graph = lambda x, k, normalize=True, layout="bcn", index_dtype=torch.int64: next(it).to(
            x.device, dtype=index_dtype)
        return self

    def __exit__(self, *exc):
        self._ops.knn_graph = self._orig


# ---- synthetic {query, db, dummy_db} set for the eval_faiss goldens (tests/golden/make_eval_golden.py) ------------
def eval_case():
    """dummy_db (400,128), db (150,128), query = db + row-dependent noise (so that some items miss), all rows
    L2-normalised; 40 test ids; lengths 1 3 5 9; k_probe 20.  Everything comes from the closed-form hash filler."""
    def unit(a):
        return (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
    dummy = unit(hash_normalish("eval:dummy", (400, 128)))
    db = hash_normalish("eval:db", (150, 128))
    # neighbouring segments of a track overlap: make consecutive rows correlated, as real fingerprints are
    db = unit(db + 0.6 * np.roll(db, 1, axis=0))
    sigma = np.linspace(0.3, 2.2, 150, dtype=np.float32)[:, None]      # clean -> hopeless
    query = unit(db + sigma * hash_normalish("eval:noise", (150, 128)) / np.sqrt(128.0).astype(np.float32) * 4.0)
    test_ids = (np.arange(40) * 3 + 1).astype(np.int64)               # < 150 - 9
    return {"dummy_db": dummy, "db": db, "query": query, "test_ids": test_ids, "test_seq_len": "1 3 5 9",
            "k_probe": 20}


def write_eval_case(root, case):
    """The on-disk layout eval.py:126-168 reads: <name>.mm float32 memmap + <name>_shape.npy."""
    import os
    for name in ("query", "db", "dummy_db"):
        arr = np.ascontiguousarray(case[name], dtype=np.float32)
        mm = np.memmap(os.path.join(root, name + ".mm"), dtype="float32", mode="w+", shape=arr.shape)
        mm[:] = arr
        mm.flush()
        del mm
        np.save(os.path.join(root, name + "_shape.npy"), np.asarray(arr.shape))
