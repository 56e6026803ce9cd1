"""Dense building blocks on the (B, C, N) node layout.

The reference keeps activations as (B,C,N,1) images and runs 1x1 `Conv2d`s (through the conv library);
with N contiguous a 1x1 convolution is exactly the batched GEMM  W (Cout x Cin) @ X_b (Cin x N), so the
parameter-holding nn.Conv2d / nn.BatchNorm2d modules are kept (state-dict schema)
but applied functionally as plain library GEMMs (hipBLASLt/rocBLAS via torch.matmul) -- no conv
library, no layout change, and bf16 autocast applies to them directly.
"""
import torch
import torch.nn.functional as F


def pointwise(conv, x):
    """1x1 Conv2d (any `groups`) applied to x (B,Cin,N) -> (B,Cout,N)."""
    B, cin, N = x.shape
    cout, g = conv.out_channels, conv.groups
    w = conv.weight.reshape(cout, cin // g)
    if g == 1:
        y = torch.matmul(w, x)
    else:
        y = torch.matmul(w.reshape(g, cout // g, cin // g), x.reshape(B, g, cin // g, N)).reshape(B, cout, N)
    if conv.bias is not None:
        y = y + conv.bias.reshape(1, cout, 1).to(y.dtype)
    return y


def strided3(conv, x):
    """Conv2d(k=3, stride=2, pad=1) applied to the (N,1) node grid (graph_encoder.py:21-24).  Only kernel
    column 1 ever overlaps data (columns 0 and 2 see the zero padding of the width-1 axis), so the op is
    a 3-tap stride-2 convolution along N: gather the three taps and run one GEMM with K = 3*Cin."""
    B, cin, N = x.shape
    cout = conv.out_channels
    n_out = (N - 1) // 2 + 1
    xp = F.pad(x, (1, 1))
    taps = torch.cat([xp[:, :, t:t + 2 * n_out - 1:2] for t in range(3)], dim=1)      # (B, 3Cin, n_out)
    w = conv.weight[:, :, :, 1].permute(0, 2, 1).reshape(cout, 3 * cin)                # [o][t*Cin + c]
    y = torch.matmul(w, taps)
    if conv.bias is not None:
        y = y + conv.bias.reshape(1, cout, 1).to(y.dtype)
    return y


def batchnorm(bn, x):
    """nn.BatchNorm2d semantics (batch statistics + running-stat update in train mode) on (B,C,N)."""
    use_batch = bn.training or not bn.track_running_stats
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, use_batch,
                        0.0 if bn.momentum is None else bn.momentum, bn.eps)
