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

        def rec(x, k, normalize=True):
            idx = self._orig(x, k, normalize)
            self.graphs.append(idx.cpu())
            return idx
        ops.knn_graph = rec
        return self

    def __exit__(self, *exc):
        self._ops.knn_graph = self._orig

    def replay_fn(self):
        it = iter(self.graphs)
        self.flips = 0

        def idx_fn(x, k):
            from oracle import native
            idx = next(it)
            own = native.knn_graph(x.detach().numpy(), k)
            self.flips += int((own != idx.numpy()).any(-1).sum())
            return idx
        return idx_fn
