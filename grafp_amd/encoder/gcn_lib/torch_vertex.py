"""Graph convolution blocks (mirror of encoder/gcn_lib/torch_vertex.py for the live 'mr' conv).

MRConv2d's gather -> subtract -> max -> interleave (torch_vertex.py:21-32) is ops.max_relative (one HIP
kernel forward, one backward); the grouped 1x1 conv + BN + ReLU that follows is a library GEMM.  Edge /
GraphSAGE / GIN convs are unreachable in the reference (conv is hard-coded to 'mr',
encoder/graph_encoder.py:123) and are not provided.
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ... import ops
from .._dense import conv_bn_act, from_cbn, shortcut_token, to_cbn
from .pos_embed import get_2d_relative_pos_embed
from .torch_edge import DenseDilatedKnnGraph
from .torch_nn import BasicConv


class MRConv2d(nn.Module):
    """Max-Relative graph conv: nn( interleave(x, max_j (x_j - x_i)) )."""

    def __init__(self, in_channels, out_channels, act="relu", norm=None, bias=True):
        super().__init__()
        self.nn = BasicConv([in_channels * 2, out_channels], act, norm, bias)

    def aggregate_cbn(self, x, nn_idx, groups=1, consumer=None):
        """x (C,B,N), nn_idx (B,N,k) -> (Cout,B,N); with `consumer` (the one layer that reads the result) ->
        (output, DeferredNorm or None): see BasicConv.forward_cbn_deferred."""
        cat = ops.max_relative(x, nn_idx, layout="cbn")
        if consumer is not None:
            return self.nn.forward_cbn_deferred(cat, groups, consumer)
        return self.nn.forward_cbn(cat, groups)

    def forward(self, x, edge_index, y=None):
        if y is not None:
            raise NotImplementedError("r > 1 (separate y) is unreachable in GraFPrint")
        return from_cbn(self.aggregate_cbn(to_cbn(x), edge_index[0]), x)


class GraphConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, conv="edge", act="relu", norm=None, bias=True):
        super().__init__()
        if conv != "mr":
            raise NotImplementedError(f"conv:{conv} is not supported (GraFPrint hard-codes 'mr')")
        self.gconv = MRConv2d(in_channels, out_channels, act, norm, bias)

    def forward(self, x, edge_index, y=None):
        return self.gconv(x, edge_index, y)


class DyGraphConv2d(GraphConv2d):
    """Rebuilds the k-NN graph from the current features on every call (torch_vertex.py:114-139)."""

    def __init__(self, in_channels, out_channels, kernel_size=9, dilation=1, conv="edge", act="relu",
                 norm=None, bias=True, stochastic=False, epsilon=0.0, r=1):
        super().__init__(in_channels, out_channels, conv, act, norm, bias)
        if r != 1:
            raise NotImplementedError("r > 1 is unreachable in GraFPrint")
        self.k, self.d, self.r = kernel_size, dilation, r
        self.dilated_knn_graph = DenseDilatedKnnGraph(kernel_size, dilation, stochastic, epsilon)

    def forward_cbn(self, x, groups=1, consumer=None):
        # edges stay in the compact int32 format between the two graph kernels (the int64 (2,B,N,k) edge_index of the
        # reference is only materialised by the public forward())
        idx = self.dilated_knn_graph.neighbours(x, layout="cbn", index_dtype=torch.int32)
        return self.gconv.aggregate_cbn(x, idx, groups, consumer)

    def forward(self, x, relative_pos=None):
        shape = x.shape
        nodes = x.reshape(shape[0], shape[1], -1)
        out = from_cbn(self.forward_cbn(to_cbn(nodes)), nodes)
        return out.reshape(shape[0], -1, *shape[2:])


class Grapher(nn.Module):
    """fc1 -> dynamic graph conv -> fc2, plus the residual (torch_vertex.py:146-194)."""

    def __init__(self, in_channels, kernel_size=9, dilation=1, conv="edge", act="relu", norm=None, bias=True,
                 stochastic=False, epsilon=0.0, r=1, n=196, drop_path=0.0, relative_pos=False):
        super().__init__()
        if drop_path > 0.0:
            raise NotImplementedError("drop_path > 0 never occurs in GraFPrint (idx is never incremented)")
        self.channels, self.n, self.r = in_channels, n, r
        self.fc1 = nn.Sequential(nn.Conv2d(in_channels, in_channels, 1), nn.BatchNorm2d(in_channels))
        self.graph_conv = DyGraphConv2d(in_channels, in_channels * 2, kernel_size, dilation, conv, act, norm, bias,
                                        stochastic, epsilon, r)
        self.fc2 = nn.Sequential(nn.Conv2d(in_channels * 2, in_channels, 1), nn.BatchNorm2d(in_channels))
        self.fc1[0]._shortcut_first = True     # its data gradient also carries the shortcut's (ops.ShortcutToken)
        self.drop_path = nn.Identity()
        self.relative_pos = None
        if relative_pos:   # frozen, never read by forward: kept for state-dict compatibility only
            table = torch.from_numpy(np.float32(get_2d_relative_pos_embed(in_channels, int(n ** 0.5))))
            table = F.interpolate(table[None, None], size=(n, n // (r * r)), mode="bicubic", align_corners=False)
            self.relative_pos = nn.Parameter(-table.squeeze(1), requires_grad=False)

    def forward_cbn(self, x, groups=1):
        """x (C,B,N) -> (C,B,N): 3 GEMMs, 3 fused BN kernels, the k-NN build and the max-relative gather."""
        tok = shortcut_token(x, self.fc1[0], self.fc2[0], groups)     # the shortcut's gradient rides fc1's data gradient
        y = conv_bn_act(self.fc1[0], self.fc1[1], x, groups=groups, token=tok, token_role=1)
        # stages 0-1: the grouped conv's BatchNorm + ReLU ride fc2's operand staging (no pass of their own)
        y, d = self.graph_conv.forward_cbn(y, groups, consumer=(self.fc2[0], self.fc2[1]))
        return conv_bn_act(self.fc2[0], self.fc2[1], y, residual=x, groups=groups, token=tok, token_role=2, defer=d,
                           defer_role=2)

    def forward(self, x):
        return from_cbn(self.forward_cbn(to_cbn(x)), x)
