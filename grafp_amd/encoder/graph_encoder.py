"""GraphEncoder backbone (mirror of encoder/graph_encoder.py:71-191).

Same constructor, attributes (`stem`, `backbone`, `proj`) and state-dict keys as the reference; the
forward keeps nodes as (B,C,N), builds every block's graph with the HIP k-NN kernel and runs the 1x1
convolutions as plain GEMMs (see _dense.py).  12 Grapher+FFN blocks, k-NN graph rebuilt in each; N
halves at each of the 3 Downsample modules (1024 -> 512 -> 256 -> 128 nodes).
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from ._dense import (bn_act, conv1x1, conv_bn_act, deferred_counters, deferred_norm, from_cbn, shortcut_token,
                     stride2_operands, to_cbn)
from .gcn_lib.torch_nn import act_layer
from .gcn_lib.torch_vertex import Grapher

_SIZES = {
    "t": ([2, 2, 6, 2], [64, 128, 256, 512]),
    "s": ([2, 2, 6, 2], [80, 160, 400, 640]),
    "m": ([2, 2, 16, 2], [96, 192, 384, 768]),
}


class Downsample(nn.Module):
    """3x3 stride-2 conv + BN over the (N,1) node grid: halves N, changes width."""

    def __init__(self, in_dim=3, out_dim=768):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_dim, out_dim, 3, stride=2, padding=1), nn.BatchNorm2d(out_dim))

    def forward_cbn(self, x, groups=1):
        taps, w = stride2_operands(self.conv[0], x)
        return conv_bn_act(self.conv[0], self.conv[1], taps, groups=groups, weight=w)

    def forward(self, x):
        return from_cbn(self.forward_cbn(to_cbn(x)), x)


class ChannelConv(nn.Module):
    """Present in the reference (graph_encoder.py:31-43) but never instantiated."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_dim, out_dim, kernel_size=1, bias=False), nn.BatchNorm2d(out_dim))

    def forward_cbn(self, x, groups=1):
        return conv_bn_act(self.conv[0], self.conv[1], x, groups=groups)

    def forward(self, x):
        return from_cbn(self.forward_cbn(to_cbn(x)), x)


class FFN(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act="relu", drop_path=0.0):
        super().__init__()
        if drop_path > 0.0:
            raise NotImplementedError("drop_path > 0 never occurs in GraFPrint")
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.drop_path = nn.Identity()
        self.act = act_layer(act)
        self.fc1 = nn.Sequential(nn.Conv2d(in_features, hidden_features, 1, bias=False), nn.BatchNorm2d(hidden_features))
        self.fc2 = nn.Sequential(nn.Conv2d(hidden_features, out_features, 1, bias=False), nn.BatchNorm2d(out_features))
        self.fc1[0]._shortcut_first = True     # its data gradient also carries the shortcut's (ops.ShortcutToken)

    def forward_cbn(self, x, groups=1):
        """x (C,B,N) -> (C,B,N): 2 GEMMs + 2 fused BN kernels (ReLU and the shortcut add are inside them)."""
        tok = shortcut_token(x, self.fc1[0], self.fc2[0], groups)     # the shortcut's gradient rides fc1's data gradient
        if isinstance(self.act, torch.nn.ReLU):
            # stages 0-1: the hidden activation (4C rows) is never normalised by a pass of its own -- fc2 does it on load
            d = deferred_norm(x, self.fc1[0], self.fc1[1], self.fc2[0], self.fc2[1], groups)
            h = conv_bn_act(self.fc1[0], self.fc1[1], x, act=ops.ACT_RELU, groups=groups, token=tok, token_role=1,
                            defer=d, defer_role=1)
        else:
            d = None
            h = self.act(conv_bn_act(self.fc1[0], self.fc1[1], x, groups=groups, token=tok, token_role=1))
        return conv_bn_act(self.fc2[0], self.fc2[1], h, residual=x, groups=groups, token=tok, token_role=2, defer=d,
                           defer_role=2)

    def forward(self, x):
        return from_cbn(self.forward_cbn(to_cbn(x)), x)


class GraphEncoder(nn.Module):
    def __init__(self, cfg, k=3, conv="mr", act="relu", norm="batch", bias=True, dropout=0.0, dilation=True,
                 epsilon=0.2, drop_path=0.1, size="t", emb_dims=1024, in_channels=3):
        super().__init__()
        self.blocks, self.channels = _SIZES.get(size, ([2, 2, 18, 2], [128, 256, 512, 1024]))
        self.k, self.act, self.norm, self.bias = int(k), act, norm, bias
        self.drop_path, self.emb_dims, self.epsilon = drop_path, emb_dims, epsilon
        self.dilation, self.dropout = dilation, dropout
        self.num_blocks = sum(self.blocks)
        self.conv = "mr"                      # the reference ignores its `conv` argument (:123)
        self._lowp = None                     # cached low-precision copies of the 1x1 conv weights (ops.lowp_weights)
        n_nodes = cfg["n_mels"] * cfg["n_frames"] // cfg["peak_stride"]

        self.stem = nn.Sequential(nn.Conv2d(in_channels, self.channels[0], kernel_size=1, bias=False),
                                  nn.BatchNorm2d(self.channels[0]), nn.LeakyReLU(negative_slope=0.2))
        # The reference never advances its block counter (:138-151): every block gets k neighbours,
        # dilation 1 and drop-path 0, and its bookkeeping N shrinks 4x per stage (relative_pos shapes).
        mods = []
        for stage, (depth, width) in enumerate(zip(self.blocks, self.channels)):
            if stage > 0:
                mods.append(Downsample(self.channels[stage - 1], width))
                n_nodes //= 4
            for _ in range(depth):
                mods.append(nn.Sequential(
                    Grapher(width, self.k, 1, self.conv, act, norm, bias, False, epsilon, 1, n=n_nodes,
                            drop_path=0.0, relative_pos=True),
                    FFN(width, width * 4, width, act=act, drop_path=0.0)))
        self.backbone = nn.Sequential(*mods)
        self.proj = nn.Conv2d(self.channels[-1], 1024, 1, bias=True)

    def model_init(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                m.weight.requires_grad = True
                if m.bias is not None:
                    m.bias.data.zero_()
                    m.bias.requires_grad = True

    def forward(self, x, views=1):
        """x (B, C_in, N) node features -> (B, 1024).  Internally every activation is a (C, B, N) matrix.
        views > 1: the batch holds that many equally sized views stacked along B; BatchNorm keeps separate batch
        statistics per view (the reference runs the views one after the other), everything else is per clip."""
        x = to_cbn(x)
        g = int(views)
        if torch.is_autocast_enabled() and x.is_cuda:      # all 1x1 conv weights to the autocast dtype in one launch
            if self._lowp is None:
                # (self.proj is not in the list: the readout runs in f32 on the pooled means)
                self._lowp = ops.lowp_weights([m for m in self.modules() if isinstance(m, nn.Conv2d)
                                               and m.kernel_size == (1, 1) and m is not self.proj])
            self._lowp.refresh(torch.get_autocast_dtype("cuda"))
        elif self._lowp is not None:
            self._lowp.clear()
        with deferred_counters():
            x = conv_bn_act(self.stem[0], self.stem[1], x, act=ops.ACT_LEAKY, slope=self.stem[2].negative_slope,
                            groups=g)
            for mod in self.backbone:
                if isinstance(mod, Downsample):
                    x = mod.forward_cbn(x, g)
                else:
                    x = mod[1].forward_cbn(mod[0].forward_cbn(x, g), g)
        # readout: mean over nodes commutes with the (linear) 1x1 projection -- project the (C,B) means instead
        # of the (C,B,N) activations (graph_encoder.py:187-188 projects first: same result, 128x the work)
        # Always f32, also under bf16 autocast: B*512 numbers per step, and it keeps the embedding one rounding-free
        # function of the last block's (bf16-stored) node features.
        with torch.autocast(x.device.type, enabled=False):
            pooled = x.float().mean(dim=2)                                          # (C, B)
            w = self.proj.weight.reshape(self.proj.out_channels, -1)
            h = torch.mm(w.float(), pooled) + self.proj.bias.float().reshape(-1, 1)
        return h.t().contiguous()
