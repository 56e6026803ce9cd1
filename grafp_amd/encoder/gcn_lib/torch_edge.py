"""Dynamic k-NN graph construction (mirror of the live part of encoder/gcn_lib/torch_edge.py).

The reference normalises, forms the (B,N,N) distance matrix with a batched matmul and calls topk
(:7-18, :70-103, :270-284).  Here all of it is one HIP op (ops.knn_graph: exact-f32 MFMA Gram tiles +
register top-k).  Unreachable variants of the reference (part_pairwise_distance for n > 10000, the xy_* /
*_plg / *_new forms used only with r > 1) are not provided.
"""
import torch
from torch import nn

from ... import ops


def _edge_index(nn_idx):
    B, N, k = nn_idx.shape
    center = torch.arange(N, device=nn_idx.device, dtype=nn_idx.dtype).view(1, N, 1).expand(B, N, k)
    return torch.stack((nn_idx, center), dim=0)


def dense_knn_matrix(x, k=16, relative_pos=None):
    """x (B,C,N,1) -> int64 (2,B,N,k): [neighbour idx, centre idx] (torch_edge.py:70-103)."""
    if relative_pos is not None:
        raise NotImplementedError("relative_pos is never passed on the GraFPrint path (torch_vertex.py:190)")
    with torch.no_grad():
        return _edge_index(ops.knn_graph(x, k, normalize=False))


class DenseDilated(nn.Module):
    """Pick every `dilation`-th neighbour (torch_edge.py:245-255)."""

    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.k, self.dilation, self.stochastic, self.epsilon = k, dilation, stochastic, epsilon

    def forward(self, edge_index):
        if self.stochastic and self.training and torch.rand(1) < self.epsilon:
            pick = torch.randperm(self.k * self.dilation)[:self.k]
            return edge_index[:, :, :, pick]
        return edge_index[:, :, :, ::self.dilation]


class DenseDilatedKnnGraph(nn.Module):
    """L2-normalise over channels, k*dilation nearest neighbours, dilate (torch_edge.py:270-284)."""

    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.k, self.dilation, self.stochastic, self.epsilon = k, dilation, stochastic, epsilon
        self._dilated = DenseDilated(k, dilation, stochastic, epsilon)

    def neighbours(self, x, layout="bcn", index_dtype=torch.int64):
        """(B,C,N[,1]) ['bcn'] or (C,B,N) ['cbn'] -> int64 (B,N,k): the neighbour half of the edge index
        (centres are arange)."""
        with torch.no_grad():
            idx = ops.knn_graph(x, self.k * self.dilation, normalize=True, layout=layout, index_dtype=index_dtype)
        if self.dilation > 1 or self.stochastic:
            idx = self._dilated(idx.unsqueeze(0)).squeeze(0).contiguous()
        return idx

    def forward(self, x, y=None, relative_pos=None):
        if y is not None or relative_pos is not None:
            raise NotImplementedError("r > 1 / relative_pos graphs are unreachable in GraFPrint (r = 1)")
        return _edge_index(self.neighbours(x))
