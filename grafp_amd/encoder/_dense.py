"""Dense building blocks on the MI355X-first activation layout (C, B, N) -- "CBN".

The reference keeps activations as (B,C,N,1) images and runs 1x1 `Conv2d`s through the conv library.  Here an
activation is a (C, M = B*N) matrix whose ROWS are channels (N contiguous inside each clip, clips side by side):

  * a 1x1 convolution is ONE plain GEMM  W (Cout x Cin) @ X (Cin x M)  with M = B*N = 262 144 at B=256 -- not B
    small batched GEMMs; its weight gradient is ONE GEMM with K = M (no (B,Cout,Cin) intermediate + reduction),
    its input gradient one more; bf16 autocast applies to them directly;
  * BatchNorm statistics are reductions over one contiguous row: fused with the conv bias, the activation and the
    residual add in a single HIP kernel pair (ops.bn_act);
  * the graph kernels (ops.knn_graph / ops.max_relative) address clip b, channel c at  c*B*N + b*N  -- no layout
    or dtype copies anywhere in the block.

The parameter-holding nn.Conv2d / nn.BatchNorm2d modules are kept (state-dict schema) and
applied functionally.  `to_cbn` / `from_cbn` convert at the module boundary when a block is called on the
reference's (B,C,N[,1]) layout directly.
"""
import torch
import torch.nn.functional as F

from .. import ops


def to_cbn(x):
    """(B,C,N) or (B,C,N,1) -> (C,B,N) contiguous."""
    if x.dim() == 4:
        x = x.squeeze(-1)
    return x.permute(1, 0, 2).contiguous()


def from_cbn(y, like):
    """(C,B,N) -> the layout of `like` ((B,C,N) or (B,C,N,1))."""
    y = y.permute(1, 0, 2).contiguous()
    return y.unsqueeze(-1) if like.dim() == 4 else y


def conv1x1(conv, x):
    """1x1 Conv2d weights (any `groups`, bias NOT applied) on x (Cin,B,N) -> (Cout,B,N): one GEMM (a 4-batch GEMM
    for the grouped conv of the max-relative block)."""
    cin, B, N = x.shape
    cout, g = conv.out_channels, conv.groups
    return ops.conv1x1_rows(x.reshape(cin, B * N), conv.weight, g, getattr(conv, "_w_lowp", None)).reshape(cout, B, N)


# num_batches_tracked of every BatchNorm touched inside a `deferred_counters()` block advance with ONE multi-tensor
# add at the end of the block instead of one tiny launch per layer (64 per forward pass of the encoder)
_PENDING_COUNTERS = None


class deferred_counters:
    def __enter__(self):
        global _PENDING_COUNTERS
        self._outer = _PENDING_COUNTERS
        _PENDING_COUNTERS = []
        return self

    def __exit__(self, *exc):
        global _PENDING_COUNTERS
        pending, _PENDING_COUNTERS = _PENDING_COUNTERS, self._outer
        if pending and exc[0] is None:
            by_step = {}
            for t, g in pending:
                by_step.setdefault(g, []).append(t)
            for g, ts in by_step.items():
                torch._foreach_add_(ts, g)
        return False


def conv3_stride2(conv, x):
    """Conv2d(k=3, stride=2, pad=1) on the (N,1) node grid (graph_encoder.py:21-24), bias NOT applied.  Only kernel
    column 1 ever overlaps data (columns 0 and 2 see the zero padding of the width-1 axis), so the op is a 3-tap
    stride-2 convolution along N: gather the three taps, then one GEMM with K = 3*Cin (the conv library computes
    3x the flops on zeros)."""
    cin, B, N = x.shape
    cout = conv.out_channels
    n_out = (N - 1) // 2 + 1
    taps, w = stride2_operands(conv, x)
    return ops.conv1x1_rows(taps.reshape(3 * cin, B * n_out), w).reshape(cout, B, n_out)


def stride2_operands(conv, x):
    """The (3*Cin, B, n_out) tap gather of x and the (Cout, 3*Cin) weight matrix whose product is conv3_stride2."""
    cin, B, N = x.shape
    taps = ops.stride2_taps(x)                                                          # (3, Cin, B, n_out)
    w = conv.weight[:, :, :, 1].permute(0, 2, 1).reshape(conv.out_channels, 3 * cin)    # [o][t*Cin + c]
    return taps.reshape(3 * cin, B, taps.shape[-1]), w


def _bn_training(bn, groups):
    """nn.BatchNorm2d bookkeeping shared by the fused and the two-step forms: returns (use batch statistics, momentum)
    and advances num_batches_tracked (once per view)."""
    training = bn.training or not bn.track_running_stats
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if _PENDING_COUNTERS is not None:
            _PENDING_COUNTERS.append((bn.num_batches_tracked, groups))
        else:
            bn.num_batches_tracked.add_(groups)
    return training, (0.0 if bn.momentum is None else bn.momentum)


def deferred_norm(x, prod_conv, prod_bn, cons_conv, cons_bn, groups=1):
    """An ops.DeferredNorm for  x -> [prod_conv, prod_bn, act] -> [cons_conv, cons_bn ...]  when the first layer's output
    feeds only the second, both take the fused bf16 path in training mode and the measured rule says it pays."""
    if not (prod_bn.training and cons_bn.training and cons_conv.groups == 1):
        return None
    cin, B, N = x.shape
    xa = x
    if torch.is_autocast_enabled() and x.is_cuda and x.dtype == torch.float32:
        xa = x.to(torch.get_autocast_dtype("cuda"))
    if not ops.conv_bn_act_supported(xa.reshape(cin, B * N), prod_conv.out_channels, prod_conv.groups, groups):
        return None
    if not ops.conv_bn_act_shape_supported(cons_conv.in_channels, B * N, cons_conv.out_channels, 1, groups):
        return None
    if not ops.defer_norm_pays(cons_conv.out_channels, cons_conv.in_channels, B * N):
        return None
    return ops.DeferredNorm()


def conv_bn_act(conv, bn, x, residual=None, act=ops.ACT_NONE, slope=0.0, groups=1, weight=None, use_bias=True,
                token=None, token_role=0, defer=None, defer_role=0):
    """[1x1 Conv2d -> BatchNorm2d -> activation -> + shortcut] on x (Cin, B, N) -> (Cout, B, N).  bf16 activations on
    the GPU take the fused path (ops.conv_bn_act: hand-written GEMM with the batch statistics in its epilogue, one
    normalise pass); everything else the GEMM + fused BatchNorm kernel pair.  `weight`: a 2-D (Cout, K) weight derived
    from conv.weight (Downsample's tap matrix) instead of the conv's own; groups = views stacked along B."""
    cin, B, N = x.shape
    cout = conv.out_channels
    cg = conv.groups if weight is None else 1
    if torch.is_autocast_enabled() and x.is_cuda and x.dtype == torch.float32:
        x = x.to(torch.get_autocast_dtype("cuda"))
    x2 = x.reshape(cin, B * N)
    bias = conv.bias if use_bias else None
    if ops.conv_bn_act_supported(x2, cout, cg, groups):
        training, momentum = _bn_training(bn, groups)
        w = conv.weight if weight is None else weight
        res = None if residual is None else residual.reshape(cout, B * N)
        z = ops.conv_bn_act(x2, w, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, momentum, bn.eps,
                            bias, res, act, slope, cg, groups, getattr(conv, "_w_lowp", None) if weight is None else None,
                            token, token_role, getattr(conv, "_w_t", None) if weight is None else None,
                            getattr(conv, "_w_aug", None) if weight is None else None, defer, defer_role,
                            getattr(conv, "_lowp_owner", None) if weight is None else None)
        return z.reshape(cout, B, N)
    if defer is not None:
        raise RuntimeError("conv_bn_act: a DeferredNorm was handed to a layer outside the fused bf16 path")
    if weight is None:
        y = conv1x1(conv, x)
    else:
        y = ops.conv1x1_rows(x2, weight).reshape(cout, B, N)
    return bn_act(bn, y, pre_bias=bias, residual=residual, act=act, slope=slope, groups=groups)


def shortcut_token(x, first_conv, last_conv, groups=1):
    """A ShortcutToken for the residual block  x -> first_conv ... last_conv(+ x)  when both ends take the fused bf16
    path in training mode (otherwise None: autograd adds the two gradients of x as usual)."""
    if not (torch.is_grad_enabled() and x.requires_grad and first_conv.groups == 1):
        return None
    cin, B, N = x.shape
    xa = x
    if torch.is_autocast_enabled() and x.is_cuda and x.dtype == torch.float32:
        xa = x.to(torch.get_autocast_dtype("cuda"))
    x2 = xa.reshape(cin, B * N)
    if not (ops.conv_bn_act_supported(x2, first_conv.out_channels, 1, groups) and ops.shortcut_token_supported(x2)):
        return None
    if not ops.conv_bn_act_shape_supported(last_conv.in_channels, B * N, last_conv.out_channels, last_conv.groups, groups):
        return None
    return ops.ShortcutToken()


def bn_act(bn, y, pre_bias=None, residual=None, act=ops.ACT_NONE, slope=0.0, groups=1):
    """nn.BatchNorm2d semantics (batch statistics + running-stat update in train mode) over the rows of y (C,B,N),
    fused with the preceding conv's bias, the activation and the residual add.  groups = number of views stacked
    along B: each view keeps its OWN batch statistics and the running statistics advance once per view, exactly
    as when the views pass through the module one after the other (simclr/simclr.py:35,43)."""
    training, momentum = _bn_training(bn, groups)
    return ops.bn_act(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, momentum, bn.eps, pre_bias,
                      residual, act, slope, groups)
