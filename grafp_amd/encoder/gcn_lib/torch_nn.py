"""Basic layers of the graph encoder (mirror of encoder/gcn_lib/torch_nn.py: act_layer, norm_layer,
BasicConv, batched_index_select, MLP)."""
import torch
from torch import nn

from ... import ops
from .._dense import bn_act, conv1x1, conv_bn_act, deferred_norm, from_cbn, to_cbn

_ACTS = {
    "relu": lambda inplace, slope, n: nn.ReLU(inplace),
    "leakyrelu": lambda inplace, slope, n: nn.LeakyReLU(slope, inplace),
    "prelu": lambda inplace, slope, n: nn.PReLU(num_parameters=n, init=slope),
    "gelu": lambda inplace, slope, n: nn.GELU(),
    "hswish": lambda inplace, slope, n: nn.Hardswish(inplace),
}


def act_layer(act, inplace=False, neg_slope=0.2, n_prelu=1):
    try:
        return _ACTS[act.lower()](inplace, neg_slope, n_prelu)
    except KeyError:
        raise NotImplementedError("activation layer [%s] is not found" % act)


def norm_layer(norm, nc):
    norm = norm.lower()
    if norm == "batch":
        return nn.BatchNorm2d(nc, affine=True)
    if norm == "instance":
        return nn.InstanceNorm2d(nc, affine=False)
    raise NotImplementedError("normalization layer [%s] is not found" % norm)


class MLP(nn.Sequential):
    def __init__(self, channels, act="relu", norm=None, bias=True):
        layers = []
        for cin, cout in zip(channels[:-1], channels[1:]):
            layers.append(nn.Linear(cin, cout, bias))
            if act is not None and act.lower() != "none":
                layers.append(act_layer(act))
            if norm is not None and norm.lower() != "none":
                layers.append(norm_layer(norm, channels[-1]))
        super().__init__(*layers)


class BasicConv(nn.Sequential):
    """[Conv2d(1x1, groups=4) -> norm -> act] per consecutive channel pair (torch_nn.py:52-76): same child
    indices / parameter names / init (kaiming-normal weights, zero bias); forward works on (B,C,N)."""

    def __init__(self, channels, act="relu", norm=None, bias=True, drop=0.0):
        layers = []
        for cin, cout in zip(channels[:-1], channels[1:]):
            layers.append(nn.Conv2d(cin, cout, 1, bias=bias, groups=4))
            if norm is not None and norm.lower() != "none":
                layers.append(norm_layer(norm, channels[-1]))
            if act is not None and act.lower() != "none":
                layers.append(act_layer(act))
            if drop > 0:
                layers.append(nn.Dropout2d(drop))
        super().__init__(*layers)
        self.reset_parameters()

    def reset_parameters(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d)) and m.weight is not None:
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def forward_cbn_deferred(self, x, groups, consumer):
        """As forward_cbn for a [conv, BatchNorm, ReLU] stack whose output feeds only `consumer` = (conv, bn), another
        fused layer: -> (output, ops.DeferredNorm or None).  With a token the output is the RAW convolution output and
        the consumer applies BatchNorm + ReLU while it stages its operand (_dense.deferred_norm decides)."""
        mods = list(self)
        if (len(mods) == 3 and isinstance(mods[0], nn.Conv2d) and isinstance(mods[1], nn.BatchNorm2d)
                and isinstance(mods[2], nn.ReLU)):
            d = deferred_norm(x, mods[0], mods[1], consumer[0], consumer[1], groups)
            if d is not None:
                return conv_bn_act(mods[0], mods[1], x, act=ops.ACT_RELU, groups=groups, defer=d, defer_role=1), d
        return self.forward_cbn(x, groups), None

    def forward_cbn(self, x, groups=1):
        """x (Cin,B,N) -> (Cout,B,N).  [conv, BatchNorm, ReLU] triples run as GEMM + one fused kernel."""
        mods, i = list(self), 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Conv2d):
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                if isinstance(nxt, nn.BatchNorm2d):
                    act_mod = mods[i + 2] if i + 2 < len(mods) else None
                    if isinstance(act_mod, nn.ReLU):
                        x, i = conv_bn_act(m, nxt, x, act=ops.ACT_RELU, groups=groups), i + 3
                    elif isinstance(act_mod, nn.LeakyReLU):
                        x, i = conv_bn_act(m, nxt, x, act=ops.ACT_LEAKY, slope=act_mod.negative_slope, groups=groups), i + 3
                    else:
                        x, i = conv_bn_act(m, nxt, x, groups=groups), i + 2
                    continue
                y = conv1x1(m, x)
                x = y if m.bias is None else y + m.bias.reshape(-1, 1, 1).to(y.dtype)
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.Dropout2d)):
                x = bn_act(m, x, groups=groups) if isinstance(m, nn.BatchNorm2d) else to_cbn(m(from_cbn(x, x.new_empty(0, 0, 0, 0))))
            else:
                x = m(x)
            i += 1
        return x

    def forward(self, x):
        return from_cbn(self.forward_cbn(to_cbn(x)), x)


def batched_index_select(x, idx):
    """x (B,C,N,1), idx (B,N,k) -> (B,C,N,k) neighbour features (torch_nn.py:79-98).  Kept for API
    completeness; the live path never materialises this tensor (ops.max_relative fuses it away)."""
    B, C, N = x.shape[:3]
    k = idx.shape[-1]
    flat = idx.reshape(B, 1, -1).expand(B, C, idx.shape[1] * k)
    return torch.gather(x.reshape(B, C, N), 2, flat).reshape(B, C, idx.shape[1], k)
