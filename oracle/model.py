"""Functional torch-CPU restatement of the GraFPrint model path.  TEST INFRASTRUCTURE (see __init__).

Everything is a pure function of a flat state dict `sd` (name -> tensor, the reference's 443-key
schema) so that it shares no code with grafp_amd's nn.Module mirror.  Citations are into
/root/reference.
"""
import math

import torch
import torch.nn.functional as F

# ------------------------------------------------------------------------------------------------
# K1 / K1b  log-mel (modules/transformations.py:50-57, 78, 83, 89-90; torchaudio 2.3.0 definitions)
# ------------------------------------------------------------------------------------------------


def mel_filterbank(n_freqs=513, n_mels=64, sample_rate=16000, f_min=0.0, f_max=None):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') -> (n_freqs, n_mels) f32."""
    f_max = float(sample_rate // 2) if f_max is None else f_max
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


def logmel(x, cfg):
    """(..., T) waveform -> (..., n_mels, 1 + T // hop) dB.  MelSpectrogram(power=2, hann periodic,
    center, reflect, onesided, htk, norm=None) followed by AmplitudeToDB(stype='power', top_db=None)."""
    n_fft, hop, win = cfg["n_fft"], cfg["hop_len"], cfg["win_len"]
    shape = x.shape
    x2 = x.reshape(-1, shape[-1]).float()
    spec = torch.stft(x2, n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win),
                      center=True, pad_mode="reflect", normalized=False, onesided=True,
                      return_complex=True)
    power = spec.abs().pow(2.0)                                           # (B, 513, frames)
    fb = mel_filterbank(n_fft // 2 + 1, cfg["n_mels"], cfg["fs"])
    mel = torch.matmul(power.transpose(-1, -2), fb).transpose(-1, -2)      # (B, n_mels, frames)
    db = 10.0 * torch.log10(torch.clamp(mel, min=1e-10))                  # ref=1 -> no offset
    return db.reshape(*shape[:-1], cfg["n_mels"], db.shape[-1])


def val_segments(x, cfg):
    """Whole-track branch (transformations.py:89-90): (1,T) or (T,) -> (n_seg, n_mels, n_frames)."""
    X = logmel(x.reshape(-1), cfg).transpose(1, 0)                        # (frames, n_mels)
    step = int(cfg["n_frames"] * (1 - cfg["overlap"]))
    return X.unfold(0, size=cfg["n_frames"], step=step)                   # (n_seg, n_mels, n_frames)


# ------------------------------------------------------------------------------------------------
# K2  peak extractor (peak_extractor.py:56-82)
# ------------------------------------------------------------------------------------------------


def peak_extract(sd, spec, stride=2, prefix="peak_extractor."):
    B, H, W = spec.shape
    lo = torch.amin(spec, dim=(1, 2), keepdim=True)
    hi = torch.amax(spec, dim=(1, 2), keepdim=True)
    s = (spec - lo) / (hi - lo)
    t_ramp = torch.linspace(0, 1, steps=W).view(1, 1, W).expand(B, H, W)
    f_ramp = torch.linspace(0, 1, steps=H).view(1, H, 1).expand(B, H, W)
    inp = torch.stack((t_ramp, f_ramp, s), dim=1)                          # (B,3,H,W)
    w = sd[prefix + "convs.0.weight"]
    y = F.conv2d(inp, w, sd[prefix + "convs.0.bias"], stride=(stride, 1),
                 padding=(w.shape[2] // 2, w.shape[3] // 2))
    return F.relu(y).reshape(B, w.shape[0], -1)


# ------------------------------------------------------------------------------------------------
# K3-K5  k-NN graph (torch_edge.py:7-18, 70-103, 270-284) -- torch restatement (float semantics of
# the reference; tie order and accumulation order are whatever torch does).  The bit-exact,
# fully-specified version is csrc/knn_graph.c.
# ------------------------------------------------------------------------------------------------


def knn_graph_torch(x, k):
    """x (B,C,N) -> nn_idx int64 (B,N,k); centre index is implicit (arange)."""
    with torch.no_grad():
        xn = F.normalize(x.detach().float(), p=2.0, dim=1)                # over channels
        p = xn.transpose(2, 1)                                            # (B,N,C)
        inner = -2 * torch.matmul(p, p.transpose(2, 1))
        sq = torch.sum(p * p, dim=-1, keepdim=True)
        dist = sq + inner + sq.transpose(2, 1)
        return torch.topk(-dist, k=k)[1]


# ------------------------------------------------------------------------------------------------
# K6-K8  gather + max-relative + grouped conv (torch_nn.py:79-98, torch_vertex.py:19-34)
# ------------------------------------------------------------------------------------------------


def gather_nodes(x, idx):
    """x (B,C,N), idx (B,N,K) -> (B,C,N,K): out[b,c,n,k] = x[b,c,idx[b,n,k]]."""
    B, C, N = x.shape
    K = idx.shape[-1]
    flat = idx.reshape(B, 1, N * K).expand(B, C, N * K)
    return torch.gather(x, 2, flat).reshape(B, C, N, K)


def max_relative(x, idx):
    """(B,C,N),(B,N,K) -> (B,2C,N): channel 2c = x[c], channel 2c+1 = max_k(x[c,idx]-x[c])."""
    B, C, N = x.shape
    rel = (gather_nodes(x, idx) - x.unsqueeze(-1)).max(dim=-1)[0]
    return torch.stack((x, rel), dim=2).reshape(B, 2 * C, N)


def _bn(sd, name, x, train, momentum=0.1, eps=1e-5):
    if train and (name + ".num_batches_tracked") in sd:
        sd[name + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], train, momentum, eps)


def round_bf16(x):
    """Round-to-nearest-even to bfloat16, returned as f32: the storage rounding of the bf16 mode.  Under autograd the
    GRADIENT passing through is rounded to bf16 as well (the backward of the two casts), which is where the HIP bf16
    path stores its gradients: dY behind the BatchNorm backward, dX behind the data-gradient product."""
    return x.to(torch.bfloat16).to(torch.float32)


class _RoundStraightThrough(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def round_bf16_weight(w):
    """The bf16 operand copy of an f32 parameter: rounded forward, gradient passed through unrounded (the HIP path
    accumulates and returns weight gradients in f32: grafp_conv1x1_wgrad_bf16)."""
    return _RoundStraightThrough.apply(w)


round_bf16.weight = round_bf16_weight


def _conv1x1(sd, name, x, groups=1, q=None, **conv_kw):
    """q = None: the reference's f32 convolution.  q = round_bf16: the bf16 mode of the HIP path restated -- operands
    rounded to bf16, products accumulated in f32, the bias-free output rounded to bf16 for storage, the conv bias
    added in f32 by the normalisation that follows (grafp_amd/encoder/_dense.py: `pre_bias`)."""
    if q is None:
        return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), groups=groups, **conv_kw)
    qw = getattr(q, "weight", q)                       # weights: rounded operand copy, f32 gradient (see round_bf16_weight)
    y = q(F.conv2d(q(x), qw(sd[name + ".weight"]), None, groups=groups, **conv_kw))
    b = sd.get(name + ".bias")
    return y if b is None else y + b.reshape(1, -1, 1, 1)


def _q(q, x):
    return x if q is None else q(x)


def _notap(name, t):
    return t


def grapher(sd, p, x, k, train, idx_fn=knn_graph_torch, q=None, tap=None):
    """torch_vertex.py:183-194.  x (B,C,N,1).  q: storage rounding of the bf16 mode (see _conv1x1); tap(name, tensor)
    sees (and may replace) every stored activation -- the layer-by-layer comparisons of tests/test_gpu_bf16.py."""
    tap = tap or _notap
    short = x
    x = tap(p + "fc1", _q(q, _bn(sd, p + "fc1.1", _conv1x1(sd, p + "fc1.0", x, q=q), train)))
    idx = idx_fn(x.squeeze(-1), k)
    m = tap(p + "mr", _q(q, max_relative(x.squeeze(-1), idx).unsqueeze(-1)))
    g = p + "graph_conv.gconv.nn."
    m = tap(g + "0", _q(q, F.relu(_bn(sd, g + "1", _conv1x1(sd, g + "0", m, groups=4, q=q), train))))
    x = _bn(sd, p + "fc2.1", _conv1x1(sd, p + "fc2.0", m, q=q), train)
    return tap(p + "fc2", _q(q, x + short))


def ffn(sd, p, x, train, q=None, tap=None):
    """graph_encoder.py:60-67."""
    tap = tap or _notap
    h = tap(p + "fc1", _q(q, F.relu(_bn(sd, p + "fc1.1", _conv1x1(sd, p + "fc1.0", x, q=q), train))))
    return tap(p + "fc2", _q(q, _bn(sd, p + "fc2.1", _conv1x1(sd, p + "fc2.0", h, q=q), train) + x))


def downsample(sd, p, x, train, q=None, tap=None):
    """graph_encoder.py:21-28: 3x3 stride-2 pad-1 conv on the (N,1) grid, then BN."""
    tap = tap or _notap
    y = _conv1x1(sd, p + "conv.0", x, q=q, stride=2, padding=1)
    return tap(p + "conv", _q(q, _bn(sd, p + "conv.1", y, train)))


def graph_encoder(sd, x, train, k=3, prefix="encoder.", idx_fn=knn_graph_torch, q=None, tap=None):
    """graph_encoder.py:167-191.  x (B,C_in,N) -> (B,1024).  With q the node features are stored in bf16 after every
    layer; the readout (mean over nodes and the projection, which commute) stays in f32 as in the HIP path."""
    tap_ = tap or _notap
    x = x.unsqueeze(-1)
    x = tap_(prefix + "stem", _q(q, F.leaky_relu(_bn(sd, prefix + "stem.1", _conv1x1(sd, prefix + "stem.0", x, q=q),
                                                      train), 0.2)))
    i = 0
    while True:
        p = f"{prefix}backbone.{i}."
        if (p + "conv.0.weight") in sd:
            x = downsample(sd, p, x, train, q, tap)
        elif (p + "0.fc1.0.weight") in sd:
            x = grapher(sd, p + "0.", x, k, train, idx_fn, q, tap)
            x = ffn(sd, p + "1.", x, train, q, tap)
        else:
            break
        i += 1
    x = F.conv2d(x, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"])
    return torch.mean(x, dim=2).squeeze(-1).squeeze(-1)


def simclr_forward(sd, x_i, x_j, train, k=3, stride=2, idx_fn=knn_graph_torch, q=None, tap=None):
    """simclr.py:29-47: views run sequentially through the same modules."""
    def one(spec):
        h = graph_encoder(sd, peak_extract(sd, spec, stride), train, k, idx_fn=idx_fn, q=q, tap=tap)
        z = F.linear(F.elu(F.linear(h, sd["projector.0.weight"], sd["projector.0.bias"])),
                     sd["projector.2.weight"], sd["projector.2.bias"])
        return h, F.normalize(z, p=2)
    h_i, z_i = one(x_i)
    h_j, z_j = one(x_j)
    return h_i, h_j, z_i, z_j


# ------------------------------------------------------------------------------------------------
# K12  NT-Xent (simclr/ntxent.py:17-29)
# ------------------------------------------------------------------------------------------------


def ntxent_loop(z_i, z_j, tau):
    """Row-by-row restatement (small cases only)."""
    z = torch.stack((z_i, z_j), dim=1).reshape(2 * z_i.shape[0], z_i.shape[1])
    a = torch.matmul(z, z.T) / tau
    rows = []
    for r in range(z.shape[0]):
        others = torch.cat([a[r, :r], a[r, r + 1:]])
        rows.append(F.log_softmax(others, dim=0)[r if r % 2 == 0 else r - 1])
    return torch.stack(rows).sum() / -z.shape[0]


def ntxent(z_i, z_j, tau):
    """Closed form of the loop: cross-entropy of S (diag = -inf) against the partner row r^1."""
    M = 2 * z_i.shape[0]
    z = torch.stack((z_i, z_j), dim=1).reshape(M, z_i.shape[1])
    s = torch.matmul(z, z.T) / tau
    s = s.masked_fill(torch.eye(M, dtype=torch.bool, device=s.device), float("-inf"))
    return F.cross_entropy(s, torch.arange(M, device=s.device) ^ 1)


# ------------------------------------------------------------------------------------------------
# One training step (train.py:60-74) on a state dict
# ------------------------------------------------------------------------------------------------

_NON_PARAM = ("running_mean", "running_var", "num_batches_tracked", "relative_pos")


def trainable(sd):
    return {k: v for k, v in sd.items() if k.rsplit(".", 1)[-1] not in _NON_PARAM}


def make_state_dict(manifest_lines, filler):
    """Build a state dict from 'name [shape] dtype' manifest lines and filler(name, shape)->ndarray|None."""
    sd = {}
    for line in manifest_lines:
        name, rest = line.split(" ", 1)
        shape = tuple(int(s) for s in rest[rest.index("[") + 1:rest.index("]")].split(",") if s.strip())
        arr = filler(name, shape)
        if arr is not None:
            sd[name] = torch.as_tensor(arr).reshape(shape).clone()
    return sd


def train_step(sd, opt, x_i, x_j, tau, k=3, stride=2):
    """zero_grad -> forward (train-mode BN) -> NT-Xent -> backward -> optimizer step."""
    opt.zero_grad()
    _, _, z_i, z_j = simclr_forward(sd, x_i, x_j, True, k, stride)
    loss = ntxent(z_i, z_j, tau)
    loss.backward()
    opt.step()
    return loss.detach()
