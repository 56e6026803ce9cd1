"""Torch-facing wrappers over the C ABI (include/grafp_hip.h).

PyTorch is used here for device memory, the current HIP stream and autograd plumbing only; every
computation below runs in libgrafp_hip.so.  There is no CPU path: tensors that are not on a HIP device
raise immediately.
"""
import contextlib
import ctypes
import math
import os

import numpy as np
import torch
from torch.amp import custom_bwd, custom_fwd

from ._lib import check, lib

_vp = ctypes.c_void_p


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "grafp_amd ops run only on a HIP device (MI355X): got a CPU tensor and there is no CPU "
                "fallback by design (the CPU restatement lives in oracle/ and is test infrastructure).")


def _p(t):
    return _vp(t.data_ptr()) if t is not None else None


# the raw handle of torch's current stream, straight from the C bindings: torch.cuda.current_stream() builds a Stream
# object and validates the device on every call (~10 us; 160 launches of a 128-pair step made that 1.6 ms of host time
# per pass -- the eager step is host-bound at that size: tools/host_profile.py)
_raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice


def _stream():
    return _vp(_raw_stream(_cur_device()))


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


class _Switches:
    """A/B switches of the host glue for tests and measurement tools (tests/, tools/): plain attributes read at call
    time and set by the test or tool itself.  The product path reads NO environment variable for its behaviour."""
    bn_two_pass = False           # True: the two-kernel BatchNorm (statistics, then apply) instead of the single pass
    grouped_bmm_min = 128         # f32 mode: channels per group from which a grouped conv runs as a batched library GEMM
    wgrad_f32_library = False     # True: f32 weight gradients by the library GEMM instead of the split-bf16 kernel
    fused_conv_bn = True          # False: library GEMM + fused BatchNorm kernel pair instead of conv1x1_gemm (bf16 mode)
    shortcut_fusion = True        # False: autograd's accumulate kernel instead of the [W^T | I] data-gradient product
    bn_spin_limit = -1            # polls of the single-pass BatchNorm rendezvous (-1: the library default; 0: never wait)
    knn_split = True              # False: every k-NN graph by the exact-f32 MFMA kernel (knn_graph.hip) -- same indices
    fused_eval_affine = True      # False: eval-mode conv+BN+act as GEMM + normalise pass (same bits) instead of one kernel
    defer_norm = True             # False: every BatchNorm + activation by its own pass (no normalise-on-load at stages 0-1)
    mrconv_arg = True             # False: max-relative backward recomputes the arg-max from x instead of reading the record
    f32_split_gemm = False        # True: the f32 mode's matrix-bound products (>= 256 operand rows) as split-bf16 MFMAs --
    #                               2^-16 instead of 2^-24 per product: faster, but NOT inside the f32 mode's 1e-4 parity bars


switches = _Switches()


# ------------------------------------------------------------------------------------------------
# Optional per-kernel timing with HIP events on the launch stream (bench.py's roofline block)
# ------------------------------------------------------------------------------------------------
_TIMED = {}     # name -> list of (start_event, end_event, meta)


@contextlib.contextmanager
def time_kernels(*names):
    """Record HIP events around every launch of the named ops inside the block.
    Yields a dict name -> list[(start, end, meta)]; call `elapsed_ms(events)` after a synchronize."""
    for n in names:
        _TIMED[n] = []
    try:
        yield _TIMED
    finally:
        for n in names:
            _TIMED.pop(n, None)


@contextlib.contextmanager
def _timed(name, meta=None):
    rec = _TIMED.get(name)
    if rec is None:
        yield
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    yield
    e.record()
    rec.append((s, e, meta))


def elapsed_ms(events):
    return [s.elapsed_time(e) for s, e, _ in events]


# ------------------------------------------------------------------------------------------------
# K1 / K1b  log-mel
# ------------------------------------------------------------------------------------------------
_MEL_PLANS = {}


def mel_filterbank(n_freqs, n_mels, sample_rate, f_min=0.0, f_max=None):
    """HTK triangular filterbank, norm=None, in f32 exactly as torchaudio==2.3.0 builds it
    (`melscale_fbanks`; the reference gets it through MelSpectrogram, modules/transformations.py:51)."""
    f_max = float(sample_rate // 2) if f_max is None else float(f_max)
    freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    mel_lo = 2595.0 * math.log10(1.0 + f_min / 700.0)
    mel_hi = 2595.0 * math.log10(1.0 + f_max / 700.0)
    pts = 700.0 * (10.0 ** (torch.linspace(mel_lo, mel_hi, n_mels + 2) / 2595.0) - 1.0)
    width = pts[1:] - pts[:-1]
    slope = pts.unsqueeze(0) - freqs.unsqueeze(1)
    rising = slope[:, 2:] / width[1:]
    falling = (-1.0 * slope[:, :-2]) / width[:-1]
    return torch.clamp(torch.minimum(falling, rising), min=0.0)


class _MelPlan:
    def __init__(self, device, fs, n_fft, win_len, n_mels):
        win = torch.hann_window(win_len)                       # periodic, as torch.stft's default use
        if win_len < n_fft:
            left = (n_fft - win_len) // 2
            win = torch.nn.functional.pad(win, (left, n_fft - win_len - left))
        j = torch.arange(n_fft // 2, dtype=torch.float64)
        ang = 2.0 * math.pi * j / n_fft
        tw = torch.stack((torch.cos(ang), -torch.sin(ang)), dim=1).to(torch.float32)
        fb = mel_filterbank(n_fft // 2 + 1, n_mels, fs)
        nz = fb > 0
        any_nz = nz.any(dim=0)
        lo = torch.where(any_nz, nz.to(torch.int32).argmax(dim=0), torch.ones(n_mels, dtype=torch.int64))
        hi = torch.where(any_nz, fb.shape[0] - 1 - nz.flip(0).to(torch.int32).argmax(dim=0),
                         torch.zeros(n_mels, dtype=torch.int64))
        self.window = win.to(device).contiguous()
        self.twiddle = tw.to(device).contiguous()
        self.fb = fb.to(device).contiguous()
        self.band_lo = lo.to(torch.int32).to(device).contiguous()
        self.band_hi = hi.to(torch.int32).to(device).contiguous()


def _mel_plan(device, fs, n_fft, win_len, n_mels):
    key = (str(device), fs, n_fft, win_len, n_mels)
    if key not in _MEL_PLANS:
        _MEL_PLANS[key] = _MelPlan(device, fs, n_fft, win_len, n_mels)
    return _MEL_PLANS[key]


def logmel(wav, fs=16000, n_fft=1024, win_len=1024, hop=512, n_mels=64):
    """(B, T) or (T,) waveform -> (B, n_mels, 1 + T // hop) dB (or (n_mels, frames) for 1-D input)."""
    _require_gpu(wav)
    squeeze = wav.dim() == 1
    x = _f32c(wav.reshape(-1, wav.shape[-1]))
    B, T = x.shape
    plan = _mel_plan(x.device, fs, n_fft, win_len, n_mels)
    out = torch.empty((B, n_mels, 1 + T // hop), dtype=torch.float32, device=x.device)
    with _timed("logmel", (B, T)):
        check(lib.grafp_logmel_f32(_p(x), x.stride(0), B, T, n_fft, hop, n_mels, _p(plan.window), _p(plan.twiddle),
                                   _p(plan.fb), _p(plan.band_lo), _p(plan.band_hi), _p(out), _stream()), "logmel")
    return out[0] if squeeze else out


def unfold_segments(spec, size, step):
    """(n_mels, n_frames) -> contiguous (n_seg, n_mels, size); values of
    `spec.transpose(1,0).unfold(0, size, step)` (modules/transformations.py:89-90)."""
    _require_gpu(spec)
    s = _f32c(spec)
    n_mels, n_frames = s.shape
    if n_frames < size:
        return torch.empty((0, n_mels, size), dtype=torch.float32, device=s.device)
    n_seg = (n_frames - size) // step + 1
    seg = torch.empty((n_seg, n_mels, size), dtype=torch.float32, device=s.device)
    check(lib.grafp_unfold_segments_f32(_p(s), n_mels, n_frames, size, step, _p(seg), _stream()), "unfold_segments")
    return seg


# ------------------------------------------------------------------------------------------------
# device-side augmentation (modules/transformations.py:25-48)
# ------------------------------------------------------------------------------------------------
def _i32c(t, device):
    return t.detach().to(device=device, dtype=torch.int32).contiguous()


def _bank(bank, lens, device, starts=None):
    """(flat f32 buffer, int64 starts, int32 lengths) of a recording bank.  bank: a flat 1-D buffer with `starts`, or a
    2-D (n, Lmax) array whose row r holds recording r in its first lens[r] elements."""
    bank = _f32c(bank)
    lens = _i32c(lens, device)
    if bank.dim() == 2:
        starts = torch.arange(bank.shape[0], device=device, dtype=torch.int64) * bank.shape[1]
        bank = bank.reshape(-1)
    else:
        starts = starts.detach().to(device=device, dtype=torch.int64).contiguous()
    return bank, starts, lens


def ir_convolve(x, ir_bank, ir_len, ir_index=None, ir_start=None):
    """ApplyImpulseResponse for a batch: x (B,T) f32; the impulse responses as a ragged bank (flat buffer + ir_start
    + ir_len) or a padded (n_ir, Lmax) array + ir_len; ir_index (B) recording per signal (< 0: copied unchanged;
    None: recording 0) -> (B,T), the full convolution truncated to T (grafp_ir_convolve_f32)."""
    _require_gpu(x, ir_bank)
    squeeze = x.dim() == 1
    x2 = _f32c(x.reshape(1, -1) if squeeze else x)
    bank, starts, lens = _bank(ir_bank, ir_len, x2.device, ir_start)
    B, T = x2.shape
    out = torch.empty_like(x2)
    idx = None if ir_index is None else _i32c(ir_index, x2.device)   # (locals: a temporary's block could be re-used)
    check(lib.grafp_ir_convolve_f32(_p(x2), T, B, T, _p(bank), _p(starts), lens.numel(), _p(lens), _p(idx), _p(out), T,
                                    _stream()), "ir_convolve")
    return out[0] if squeeze else out


def mix_snr(x, noise_bank, noise_len, noise_index, noise_offset, snr_db, noise_start=None):
    """AddBackgroundNoise for a batch: out = x + rms(x)/10^(snr/20) * n/(rms(n)+1e-8) with n the circular read of
    recording noise_index[b] (ragged bank: flat buffer + noise_start + noise_len, or a padded (n, Lmax) array) from
    noise_offset[b]; noise_index < 0: copied unchanged (grafp_mix_snr_f32)."""
    _require_gpu(x, noise_bank)
    squeeze = x.dim() == 1
    x2 = _f32c(x.reshape(1, -1) if squeeze else x)
    bank, starts, lens = _bank(noise_bank, noise_len, x2.device, noise_start)
    B, T = x2.shape
    out = torch.empty_like(x2)
    nbytes = lib.grafp_mix_snr_workspace(B, T)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x2.device)
    idx, off = (_i32c(t, x2.device) for t in (noise_index, noise_offset))
    snr = _f32c(snr_db.to(x2.device))
    check(lib.grafp_mix_snr_f32(_p(x2), T, B, T, _p(bank), _p(starts), lens.numel(), _p(lens), _p(idx), _p(off),
                                _p(snr), _p(out), T, _p(ws), nbytes, _stream()), "mix_snr")
    return out[0] if squeeze else out


# ------------------------------------------------------------------------------------------------
# K2  peak extractor
# ------------------------------------------------------------------------------------------------
_RAMPS = {}


def _ramps(device, H, W):
    key = (str(device), H, W)
    if key not in _RAMPS:
        _RAMPS[key] = (torch.linspace(0, 1, steps=W).to(device), torch.linspace(0, 1, steps=H).to(device))
    return _RAMPS[key]


class _PeakExtract(torch.autograd.Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, spec, weight, bias, stride_h):
        _require_gpu(spec, weight, bias)
        spec, weight, bias = _f32c(spec), _f32c(weight), _f32c(bias)
        B, H, W = spec.shape
        F, cin, KH, KW = weight.shape
        if cin != 3:
            raise ValueError("peak extractor expects 3 input planes [T-ramp, F-ramp, spectrogram]")
        Ho = (H + 2 * (KH // 2) - KH) // stride_h + 1
        t_ramp, f_ramp = _ramps(spec.device, H, W)
        out = torch.empty((B, F, Ho * W), dtype=torch.float32, device=spec.device)
        with _timed("peak_extract_fwd", (B,)):
            check(lib.grafp_peak_extract_fwd_f32(_p(spec), B, H, W, _p(weight), _p(bias), F, KH, KW, stride_h,
                                                 _p(t_ramp), _p(f_ramp), _p(out), _stream()), "peak_extract_fwd")
        ctx.save_for_backward(spec, out)
        ctx.geom = (B, H, W, F, KH, KW, stride_h)
        return out

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        spec, out = ctx.saved_tensors
        B, H, W, F, KH, KW, stride_h = ctx.geom
        g = _f32c(grad_out)
        dw = torch.empty((F, 3, KH, KW), dtype=torch.float32, device=spec.device)
        db = torch.empty((F,), dtype=torch.float32, device=spec.device)
        t_ramp, f_ramp = _ramps(spec.device, H, W)
        nbytes = lib.grafp_peak_extract_bwd_workspace(B, F, KH, KW)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=spec.device)      # per-workgroup partial sums
        with _timed("peak_extract_bwd", (B,)):
            check(lib.grafp_peak_extract_bwd_f32(_p(spec), B, H, W, F, KH, KW, stride_h, _p(t_ramp), _p(f_ramp),
                                                 _p(out), _p(g), _p(dw), _p(db), _p(ws), nbytes, _stream()),
                  "peak_extract_bwd")
        return None, dw, db, None


def peak_extract(spec, weight, bias, stride_h):
    """(B,H,W) log-mel -> (B,F,Ho*W) node features; differentiable w.r.t. weight and bias."""
    return _PeakExtract.apply(spec, weight, bias, int(stride_h))


# ------------------------------------------------------------------------------------------------
# K3-K5  k-NN graph
# ------------------------------------------------------------------------------------------------
_DT = {torch.float32: 0, torch.bfloat16: 1}


def _act_view(x, layout):
    """Return (tensor, B, C, N, stride_b, stride_c) for an activation in 'bcn' = (B,C,N) or 'cbn' = (C,B,N)
    layout (N contiguous), f32 or bf16, without copying when it already is contiguous."""
    if x.dtype not in _DT:
        x = x.float()
    x = x.contiguous()
    if layout == "bcn":
        B, C, N = x.shape
        return x, B, C, N, C * N, N
    C, B, N = x.shape
    return x, B, C, N, N, B * N


def knn_graph(x, k, normalize=True, layout="bcn", index_dtype=torch.int64, prefilter=None):
    """x (B,C,N) / (B,C,N,1) [layout 'bcn'] or (C,B,N) [layout 'cbn'], f32 or bf16 -> int64 (B,N,k)
    nearest-neighbour indices (ascending distance, ties to the lowest index).  Non-differentiable, as in the
    reference (torch_edge.py:78 `no_grad`).  bf16 inputs are widened exactly; all arithmetic is f32."""
    _require_gpu(x)
    if x.dim() == 4:
        x = x.squeeze(-1)
    x, B, C, N, sb, sc = _act_view(x.detach(), layout)
    idx = torch.empty((B, N, k), dtype=index_dtype, device=x.device)
    if (prefilter is None and switches.knn_split and normalize and index_dtype in (torch.int32, torch.int64)
            and lib.grafp_knn_split_preferred_for(_DT[x.dtype], C, N, k)):
        # split-bf16 Gram matrix + certified order, exact recomputation of the near-ties (knn_split.hip): the same
        # indices bit for bit, ~2x faster than the exact-f32 MFMA kernel below where the library prefers it (f32
        # inputs: C <= 128; bf16 inputs, whose raw features are exact bf16 operands: every stage)
        knn_graph_split(x, k, layout="raw", index_dtype=index_dtype, _view=(x, B, C, N, sb, sc), _out=idx)
        return idx
    if prefilter is None:
        # Off by default.  Measured on MI355X: on random features the pre-filter path wins at C=64, N=1024 (600 vs
        # 920 us per 512 clips), but on the encoder's own early-block features the neighbours are closer than the bf16
        # rounding margin (third-nearest distance ~0.1 vs a margin of 0.03 on each side), the survivor lists overflow
        # and the exact fallback makes it 5-10x slower than the all-f32 kernel.  Kept for callers with well
        # separated data; results are identical either way.
        prefilter = False
    if prefilter and lib.grafp_knn_pre_supported(C, N, k):
        # bf16 pre-filter + exact f32 rescoring (knn_pre.hip): the same indices, several times faster
        nbytes = lib.grafp_knn_pre_workspace(B, C, N)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
        with _timed("knn_topk", (B, C, N, k)):
            check(lib.grafp_knn_graph_pre(_p(x), _DT[x.dtype], sb, sc, B, C, N, k, int(bool(normalize)), _p(idx),
                                          int(index_dtype == torch.int32), _p(ws), nbytes, _stream()), "knn_graph_pre")
        return idx
    xn = torch.empty((B, C, N), dtype=torch.float32, device=x.device)
    sq = torch.empty((B, N), dtype=torch.float32, device=x.device)
    with _timed("knn_normalize", (B, C, N, k)):
        check(lib.grafp_knn_normalize_strided(_p(x), _DT[x.dtype], sb, sc, B, C, N, int(bool(normalize)), _p(xn),
                                              _p(sq), _stream()), "knn_normalize")
    with _timed("knn_topk", (B, C, N, k)):
        topk = lib.grafp_knn_topk_i32 if index_dtype == torch.int32 else lib.grafp_knn_topk_f32
        check(topk(_p(xn), _p(sq), B, C, N, k, _p(idx), _stream()), "knn_topk")
    return idx


def knn_graph_split(x, k, layout="bcn", index_dtype=torch.int64, return_uncertified=False, _view=None, _out=None):
    """The k-NN graph through grafp_knn_graph_split (see knn_graph): (B, N, k) indices; with return_uncertified also a
    0-d int32 device tensor = the number of queries that were recomputed exactly."""
    if _view is None:
        _require_gpu(x)
        if x.dim() == 4:
            x = x.squeeze(-1)
        x, B, C, N, sb, sc = _act_view(x.detach(), layout)
    else:
        x, B, C, N, sb, sc = _view
    if not lib.grafp_knn_split_supported(C, N, k):
        raise ValueError(f"knn_graph_split: unsupported shape C={C} N={N} k={k}")
    idx = _out if _out is not None else torch.empty((B, N, k), dtype=index_dtype, device=x.device)
    nbytes = lib.grafp_knn_split_workspace_for(_DT[x.dtype], B, C, N)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    unc = torch.zeros((), dtype=torch.int32, device=x.device) if return_uncertified else None
    with _timed("knn_split", (B, C, N, k, x.element_size())):
        check(lib.grafp_knn_graph_split(_p(x), _DT[x.dtype], sb, sc, B, C, N, k, _p(idx),
                                        int(idx.dtype == torch.int32), _p(ws), nbytes, _p(unc), _stream()),
              "knn_graph_split")
    return (idx, unc) if return_uncertified else idx


# ------------------------------------------------------------------------------------------------
# K6-K7  gather + max-relative
# ------------------------------------------------------------------------------------------------
class _MaxRelative(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, layout):
        _require_gpu(x, idx)
        x, B, C, N, sb, sc = _act_view(x.detach(), layout)
        idx = (idx if idx.dtype == torch.int32 else idx.to(torch.int64)).contiguous()
        K = idx.shape[-1]
        if tuple(idx.shape) != (B, N, K):
            raise ValueError(f"idx shape {tuple(idx.shape)} does not match activations B={B} N={N}")
        out = torch.empty((B, 2 * C, N) if layout == "bcn" else (2 * C, B, N), dtype=x.dtype, device=x.device)
        o_sb, o_sc = (2 * C * N, N) if layout == "bcn" else (N, B * N)
        ctx.layout, ctx.dims = layout, (B, C, N, tuple(x.shape), x.dtype)
        ctx.from_arg = bool(switches.mrconv_arg and ctx.needs_input_grad[0]
                            and lib.grafp_mrconv_arg_supported(_DT[x.dtype], sb, sc, o_sb, o_sc, N, K))
        if ctx.from_arg:
            # training: record which neighbour won (2 bits per element) -- backward then needs neither x nor the gather
            arg = torch.empty((B, C, N // 4), dtype=torch.uint8, device=x.device)
            with _timed("mrconv_fwd", (B, C, N, K)):
                check(lib.grafp_mrconv_fwd_arg(_p(x), _DT[x.dtype], sb, sc, _p(idx), int(idx.dtype == torch.int32), B, C,
                                               N, K, _p(out), o_sb, o_sc, _p(arg), _stream()), "mrconv_fwd_arg")
            ctx.save_for_backward(arg, idx)
            return out
        fwd = lib.grafp_mrconv_fwd_strided_i32 if idx.dtype == torch.int32 else lib.grafp_mrconv_fwd_strided
        with _timed("mrconv_fwd", (B, C, N, K)):
            check(fwd(_p(x), _DT[x.dtype], sb, sc, _p(idx), B, C, N, K, _p(out), o_sb, o_sc, _stream()), "mrconv_fwd")
        ctx.save_for_backward(x, idx)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, idx = ctx.saved_tensors
        layout = ctx.layout
        B, C, N, shape, dtype = ctx.dims
        K = idx.shape[-1]
        g = grad_out.detach().to(dtype).contiguous()
        g_sb, g_sc = (2 * C * N, N) if layout == "bcn" else (N, B * N)
        sb, sc = (C * N, N) if layout == "bcn" else (N, B * N)
        if ctx.from_arg:
            dx = torch.empty(shape, dtype=dtype, device=g.device)
            with _timed("mrconv_bwd", (B, C, N, K)):
                check(lib.grafp_mrconv_bwd_arg(_p(x), _DT[dtype], _p(idx), int(idx.dtype == torch.int32), _p(g), g_sb,
                                               g_sc, B, C, N, K, _p(dx), sb, sc, _stream()), "mrconv_bwd_arg")
            return dx, None, None
        dx = torch.empty_like(x)
        with _timed("mrconv_bwd", (B, C, N, K)):
            bwd = lib.grafp_mrconv_bwd_strided_i32 if idx.dtype == torch.int32 else lib.grafp_mrconv_bwd_strided
            check(bwd(_p(x), _DT[x.dtype], sb, sc, _p(idx), _p(g), g_sb, g_sc, B, C, N, K, _p(dx), _stream()), "mrconv_bwd")
        return dx, None, None


def max_relative(x, idx, layout="bcn"):
    """x (B,C,N) ['bcn'] or (C,B,N) ['cbn'], idx (B,N,K) -> (B,2C,N) / (2C,B,N): channel 2c = x[c],
    channel 2c+1 = max_k(x[c, idx] - x[c]).  f32 or bf16 in = out dtype (arithmetic in f32)."""
    return _MaxRelative.apply(x, idx, layout)


# ------------------------------------------------------------------------------------------------
# K8/K9 glue  fused [conv bias] + BatchNorm + activation + residual on (C, M) rows
# ------------------------------------------------------------------------------------------------
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


_BN_SYNC = {}


def _bn_sync(device, C, M):
    """The rendezvous buffer of the single-pass BatchNorm kernels: filled with ones once per (device, stream); every
    call leaves it that way (include/grafp_hip.h, grafp_bn_fwd_1pass).  switches.bn_two_pass selects the two-pass
    kernels (None)."""
    if switches.bn_two_pass:
        return None
    key = (device, _raw_stream(device.index if device.index is not None else _cur_device()))
    buf = _BN_SYNC.get(key)
    need = int(lib.grafp_bn_sync_bytes(C, M))
    if buf is None or buf.numel() * 8 < need:
        buf = torch.full((max(1 << 18, (need + 7) // 8),), -1, dtype=torch.int64, device=device)
        _BN_SYNC[key] = buf
    return buf


class _BnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, pre_bias, residual, running_mean, running_var, training, momentum, eps, act,
                slope, groups):
        _require_gpu(x, gamma, beta)
        if x.dtype not in _DT:
            x = x.float()
        x = x.detach().contiguous()
        C = x.shape[0]
        M = x.numel() // C
        res = None if residual is None else residual.detach().to(x.dtype).contiguous()
        g32, b32 = _f32c(gamma), _f32c(beta)
        pb = None if pre_bias is None else _f32c(pre_bias)
        out = torch.empty_like(x)
        mean = torch.empty((C, groups), dtype=torch.float32, device=x.device)
        invstd = torch.empty((C, groups), dtype=torch.float32, device=x.device)
        nbytes = lib.grafp_bn_workspace(C, M)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
        with _timed("bn_fwd", (C, M, x.element_size())):
            check(lib.grafp_bn_fwd_1pass(_p(x), _DT[x.dtype], C, M, groups, _p(pb), _p(g32), _p(b32), _p(res), act,
                                         float(slope), float(eps), float(momentum), int(bool(training)),
                                         _p(running_mean), _p(running_var), _p(out), _p(mean), _p(invstd), _p(ws),
                                         nbytes, _p(_bn_sync(x.device, C, M)), switches.bn_spin_limit, _stream()), "bn_fwd")
        ctx.save_for_backward(x, g32, b32, pb if pb is not None else mean.new_empty(0), mean, invstd)
        ctx.cfg = (C, M, act, float(slope), bool(training), pre_bias is not None, residual is not None, groups)
        return out

    @staticmethod
    def backward(ctx, dz):
        x, g32, b32, pb, mean, invstd = ctx.saved_tensors
        C, M, act, slope, training, has_pb, has_res, groups = ctx.cfg
        dz = dz.detach().to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        # separate tensors (not views of one buffer): autograd adopts a fresh whole tensor as .grad without a copy
        dgamma = torch.empty((C,), dtype=torch.float32, device=x.device)
        dbeta = torch.empty((C,), dtype=torch.float32, device=x.device)
        # d(pre_bias), written by the kernel: zero under batch statistics, ga*invstd*sum(dy) in eval mode
        dpb = torch.empty((C,), dtype=torch.float32, device=x.device) if has_pb else None
        nbytes = lib.grafp_bn_workspace(C, M)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
        with _timed("bn_bwd", (C, M, x.element_size())):
            check(lib.grafp_bn_bwd_1pass(_p(x), _p(dz), _DT[x.dtype], C, M, groups, _p(pb) if has_pb else None, _p(g32),
                                         _p(b32), _p(mean), _p(invstd), act, slope, int(training), _p(dx), _p(dgamma),
                                         _p(dbeta), _p(dpb) if has_pb else None, _p(ws), nbytes,
                                         _p(_bn_sync(x.device, C, M)), switches.bn_spin_limit, _stream()), "bn_bwd")
        return dx, dgamma, dbeta, dpb, (dz if has_res else None), None, None, None, None, None, None, None, None


def bn_act(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, pre_bias=None, residual=None,
           act=ACT_NONE, slope=0.0, groups=1):
    """z = act(BatchNorm(x + pre_bias)) + residual over ROWS of x (C, ...): one fused HIP forward (2 passes) and
    backward (2 passes).  x / residual / z share a dtype (f32 or bf16); parameters and statistics are f32.
    running_mean / running_var are updated in place when training.  groups > 1: each row is that many equal
    column segments (the views of a contrastive batch) normalised with their own batch statistics."""
    return _BnAct.apply(x, gamma, beta, pre_bias, residual, running_mean, running_var, training, momentum, eps, act,
                        slope, int(groups))


# ------------------------------------------------------------------------------------------------
# K9  1x1 convolution on (C, M) rows: library GEMMs forward / input-gradient, hand-written weight gradient
# ------------------------------------------------------------------------------------------------
class _Stride2Taps(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        x = x.detach().contiguous()
        if x.dtype not in _DT:
            x = x.float()
        N = x.shape[-1]
        rows = x.numel() // N
        n_out = (N - 1) // 2 + 1
        out = torch.empty((3,) + tuple(x.shape[:-1]) + (n_out,), dtype=x.dtype, device=x.device)
        check(lib.grafp_stride2_taps_fwd(_p(x), _DT[x.dtype], rows, N, _p(out), _stream()), "stride2_taps_fwd")
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        shape = ctx.shape
        N = shape[-1]
        g = g.detach().contiguous()
        if g.dtype not in _DT:
            g = g.float()
        dx = torch.empty(shape, dtype=g.dtype, device=g.device)
        check(lib.grafp_stride2_taps_bwd(_p(g), _DT[g.dtype], dx.numel() // N, N, _p(dx), _stream()), "stride2_taps_bwd")
        return dx


def stride2_taps(x):
    """x (..., N) -> (3, ..., n_out): out[t][...][j] = x[...][2j + t - 1] (zero outside), the operand of the 3-tap
    stride-2 node convolution of Downsample as one gather (and one scatter-free transpose in backward)."""
    return _Stride2Taps.apply(x)


_BD_MASKS = {}


def _block_diag_weight(w, groups, dtype=None):
    """(Cout, Cin/g) grouped weight -> dense block-diagonal (Cout, Cin) in `dtype`: for the small groups of the
    max-relative conv one dense GEMM (memory-bound either way) beats a 4-batch GEMM of 32x32 problems.  ONE
    elementwise launch (broadcast product with a cached 0/1 mask, cast included) instead of the fill + one copy per
    block + cast of torch.block_diag."""
    cout, cin_g = w.shape
    dtype = dtype or w.dtype
    key = (groups, w.device, w.dtype)
    mask = _BD_MASKS.get(key)
    if mask is None:
        mask = torch.eye(groups, device=w.device, dtype=w.dtype).reshape(groups, 1, groups, 1)
        _BD_MASKS[key] = mask
    out = torch.empty((groups, cout // groups, groups, cin_g), dtype=dtype, device=w.device)
    torch.mul(w.reshape(groups, cout // groups, 1, cin_g), mask, out=out)
    return out.reshape(cout, groups * cin_g)


def split_planes(x):
    """f32 (..., M) -> (hi, lo) bf16 planes with x ~= hi + lo to 2^-17 (grafp_split_bf16_planes); rows stay rows."""
    _require_gpu(x)
    x = _f32c(x)
    hi = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    lo = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib.grafp_split_bf16_planes(_p(x), x.numel(), _p(hi), _p(lo), _stream()), "split_bf16_planes")
    return hi, lo


def split_gemm_supported(R, K, M):
    # (measured: at stages 0-1 the products are HBM-bound and the split's extra passes cost what its matrix rate saves)
    return bool(switches.f32_split_gemm) and R % 32 == 0 and K % 32 == 0 and M % 128 == 0 and max(R, K) >= 256


def split_gemm_f32(w, x):
    """y = w x for f32 w (R, K) and f32 x (K, M) on the bf16 matrix cores: operands split into bf16 hi / lo planes, y =
    Wh Xh + Wh Xl + Wl Xh with every partial product exact and f32 accumulation (~2^-16 relative; the f32 mode's forward
    and data-gradient products -- grafp_conv1x1_gemm_split_f32)."""
    R, K = w.shape
    M = x.shape[1]
    planes = torch.empty((2 * K, M), dtype=torch.bfloat16, device=x.device)
    x = _f32c(x)
    check(lib.grafp_split_bf16_planes(_p(x), x.numel(), _p(planes[:K]), _p(planes[K:]), _stream()), "split_bf16_planes")
    w = _f32c(w)
    wh = w.to(torch.bfloat16)
    wl = (w - wh.float()).to(torch.bfloat16)
    w3 = torch.cat((wh, wh, wl), dim=1).contiguous()
    y = torch.empty((R, M), dtype=torch.float32, device=x.device)
    with _timed("conv1x1_gemm_split", (R, K, M)):
        check(lib.grafp_conv1x1_gemm_split_f32(_p(w3), _p(planes), R, K, M, _p(y), _stream()), "conv1x1_gemm_split")
    return y


class _Conv1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, groups, w_lowp=None):
        """x (Cin, M) f32/bf16 rows, w (Cout, Cin/groups) parameter view -> (Cout, M) in x's dtype.
        w_lowp: optional copy of w already in x's dtype (see lowp_weights), saves the per-call cast launch."""
        x = x.detach()
        # w may be the 4-D Conv2d parameter itself (Cout, Cin/g, 1, 1): taking it un-reshaped keeps a view node out of
        # the graph, so the weight gradient returned below is adopted as .grad without a copy
        w2 = w.detach().reshape(w.shape[0], -1)
        batched = groups > 1 and w2.shape[1] >= switches.grouped_bmm_min and x.is_cuda
        if batched:
            # wide groups (stages 2-3: 128 / 256 channels per group): a `groups`-batch GEMM; the dense block-diagonal
            # form spends 4x the flops, which at 1024 x 1024 is no longer free next to the operand traffic
            w3 = w2.to(x.dtype).reshape(groups, w2.shape[0] // groups, w2.shape[1])
            y = torch.bmm(w3, x.reshape(groups, w2.shape[1], -1)).reshape(w2.shape[0], -1)
            ctx.save_for_backward(x, w3)
            ctx.groups, ctx.wshape, ctx.wfull, ctx.batched = groups, tuple(w2.shape), tuple(w.shape), True
            return y
        ctx.batched = False
        if groups > 1:
            dense = _block_diag_weight(w2, groups, x.dtype)
        elif w_lowp is not None and w_lowp.dtype == x.dtype and w_lowp.numel() == w.numel():
            dense = w_lowp.detach().reshape(w2.shape)
        else:
            dense = w2.to(x.dtype)
        ctx.split = bool(x.is_cuda and x.dtype == torch.float32 and x.numel() % 8 == 0
                         and split_gemm_supported(dense.shape[0], dense.shape[1], x.shape[1])
                         and split_gemm_supported(dense.shape[1], dense.shape[0], x.shape[1]))
        y = split_gemm_f32(dense, x) if ctx.split else torch.mm(dense, x)
        ctx.save_for_backward(x, dense)
        ctx.groups, ctx.wshape, ctx.wfull = groups, tuple(w2.shape), tuple(w.shape)
        return y

    @staticmethod
    def backward(ctx, g):
        x, dense = ctx.saved_tensors
        groups = ctx.groups
        cout, cin_g = ctx.wshape
        g = g.detach().to(x.dtype).contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.batched:
                dx = torch.bmm(dense.transpose(1, 2), g.reshape(groups, cout // groups, -1)).reshape(x.shape)
            elif getattr(ctx, "split", False):
                dx = split_gemm_f32(dense.t().contiguous(), g)
            else:
                dx = torch.mm(dense.t(), g)
        dw = None
        if ctx.needs_input_grad[1]:
            cin, M = x.shape
            if x.dtype == torch.bfloat16 and x.is_cuda:
                dw = torch.empty((cout, cin_g), dtype=torch.float32, device=x.device)
                nbytes = lib.grafp_conv1x1_wgrad_workspace(cout, cin, groups, M)
                ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
                with _timed("conv1x1_wgrad", (cout, cin, groups, M)):
                    check(lib.grafp_conv1x1_wgrad_bf16(_p(g), _p(x), cout, cin, groups, M, _p(dw), _p(ws), nbytes,
                                                       _stream()), "conv1x1_wgrad")
            elif x.dtype == torch.float32 and x.is_cuda and not switches.wgrad_f32_library:
                # f32 step: split-bf16 (hi/lo) x 3 MFMAs inside the same streaming kernel; the library's f32 GEMM
                # runs this tall-K shape ~3.5x slower (switches.wgrad_f32_library selects it)
                dw = torch.empty((cout, cin_g), dtype=torch.float32, device=x.device)
                nbytes = lib.grafp_conv1x1_wgrad_f32_workspace(cout, cin, groups, M)
                ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
                with _timed("conv1x1_wgrad", (cout, cin, groups, M)):
                    check(lib.grafp_conv1x1_wgrad_f32(_p(g), _p(x), cout, cin, groups, M, _p(dw), _p(ws), nbytes,
                                                      _stream()), "conv1x1_wgrad_f32")
            elif groups == 1:
                dw = torch.mm(g, x.t()).float()
            else:
                dw = torch.bmm(g.reshape(groups, cout // groups, M), x.reshape(groups, cin_g, M).transpose(1, 2))
                dw = dw.reshape(cout, cin_g).float()
        return dx, (None if dw is None else dw.reshape(ctx.wfull)), None, None


def conv1x1_rows(x, w, groups=1, w_lowp=None):
    """y = W x over (C, M) rows (x contiguous 2-D).  Under autocast f32 inputs are lowered to the autocast dtype
    (the rest of a block already flows in it); the weight gradient of bf16 operands is the hand-written split-K
    kernel (grafp_conv1x1_wgrad_bf16), everything else a plain library GEMM."""
    if torch.is_autocast_enabled() and x.is_cuda and x.dtype == torch.float32:
        x = x.to(torch.get_autocast_dtype("cuda"))
    return _Conv1x1.apply(x.contiguous(), w, groups, w_lowp)


class lowp_weights:
    """The weights of all 1x1 convolutions prepared for one forward + backward pass: the bf16 copy (forward operand),
    the per-group transposed bf16 copy (data-gradient operand) and, for the first layer of a residual block (conv.
    _shortcut_first), [W^T | I] -- ONE launch of grafp_weights_prepare for every layer whose rows and columns per
    group are multiples of 32, one multi-tensor cast for the rest (the 8-channel stem).  `refresh()` before the layers
    run; conv._w_lowp, conv._w_t and conv._w_aug are picked up by encoder/_dense.conv_bn_act.
    The buffers are shared between passes and rewritten by EVERY refresh() (one launch; whether the weights changed
    cannot be seen from the host: fused optimizers and replayed graphs update parameters without touching their version
    counters -- a skip-when-unchanged variant trained on stale copies and was caught by the hit-rate test).  Each refresh
    advances a generation number; a backward pass whose forward pass saw an older generation (another encoder forward,
    possibly after an optimizer step, between the two) raises instead of differentiating against whatever the buffers
    hold by then."""

    def __init__(self, convs):
        self.convs = list(convs)
        self.generation = 0             # advanced by every refresh() of THIS instance (its buffers are its own)
        self._dtype, self._dev = None, None
        self._fast, self._slow, self._slow_bufs = [], [], []
        self._table = self._tile_entry = None
        self._keep, self._src_ptrs = [], []

    def _build(self, dtype, dev):
        self._dtype, self._dev = dtype, dev
        self._fast, self._slow, self._slow_bufs, self._keep = [], [], [], []
        rows, tiles, base = [], [], 0
        for c in self.convs:
            w = c.weight
            G = c.groups
            Rg, Kg = w.shape[0] // G, w.shape[1]
            if dtype == torch.bfloat16 and Rg % 32 == 0 and Kg % 32 == 0 and w.dtype == torch.float32 and w.is_contiguous():
                lo = torch.empty((w.shape[0], Kg), dtype=dtype, device=dev)
                aug = bool(getattr(c, "_shortcut_first", False)) and G == 1
                ld_t = Rg + Kg if aug else Rg
                wt = torch.zeros((G * Kg, ld_t), dtype=dtype, device=dev)
                if aug:
                    wt[:, Rg:] = torch.eye(Kg, dtype=dtype, device=dev)
                n_t = G * (Rg // 32) * (Kg // 32)
                rows.append([w.data_ptr(), lo.data_ptr(), wt.data_ptr(), Rg, Kg, G, ld_t, base])
                tiles.append(torch.full((n_t,), len(rows) - 1, dtype=torch.int32))
                base += n_t
                self._fast.append((c, lo, wt, aug, Rg))
            elif G == 1:
                self._slow.append(c)
                self._slow_bufs.append(torch.empty(w.shape, dtype=dtype, device=dev))
        if rows:
            self._table = torch.tensor(rows, dtype=torch.int64).to(dev)
            self._tile_entry = torch.cat(tiles).to(dev)
        else:
            self._table = self._tile_entry = None

    def refresh(self, dtype):
        ws = [c.weight for c in self.convs]
        if not ws or not ws[0].is_cuda:
            return
        dev = ws[0].device
        stale = (self._dtype != dtype or self._dev != dev or
                 any(c.weight.data_ptr() != ptr for c, ptr in zip(self.convs, self._src_ptrs)))
        if stale:
            self._build(dtype, dev)
            self._src_ptrs = [c.weight.data_ptr() for c in self.convs]
            self._published = False
        self.generation += 1
        with torch.no_grad():
            if self._table is not None:
                check(lib.grafp_weights_prepare(_p(self._table), _p(self._tile_entry), int(self._tile_entry.numel()),
                                                _stream()), "weights_prepare")
            if self._slow:
                torch._foreach_copy_(self._slow_bufs, [c.weight for c in self._slow])
        # the prepared buffers are refreshed IN PLACE: the modules' attributes only change when the buffers were rebuilt
        # (or cleared) -- re-assigning 4 x 61 module attributes on every forward pass was 0.6 ms of host time per step
        if not getattr(self, "_published", False):
            self._publish()
            self._published = True

    def _publish(self):
        for c in self.convs:
            c._lowp_owner = self
        for c, lo, wt, aug, Rg in self._fast:
            c._w_lowp = lo.view(c.weight.shape) if c.weight.dim() == 4 else lo
            c._w_t, c._w_aug = (None, wt) if aug else (wt, None)
        for c, b in zip(self._slow, self._slow_bufs):
            c._w_lowp, c._w_t, c._w_aug = b, None, None

    def clear(self):
        for c in self.convs:
            c._w_lowp = c._w_t = c._w_aug = None
            c._lowp_owner = None
        self._published = False


# ------------------------------------------------------------------------------------------------
# K9  the 1x1 convolution as the hand-written streaming GEMM (gemm.hip), BatchNorm statistics / normalisation folded in
# ------------------------------------------------------------------------------------------------
def gemm_supported(R, K, groups, M, views=1):
    return bool(lib.grafp_conv1x1_gemm_supported(int(R), int(K), int(groups), int(M), int(views)))


_XL_CHECKED = set()


def _xl_selfcheck(device):
    """Once per process and device, at the first product: the four-wave tile (gemm_xl.h) keeps its accumulators in AGPRs it
    NAMES in asm text.  The build checks the generated code for that contract (Makefile -> tools/check_kernel_regs.py);
    this checks the RESULT: one product the plan sends to that tile, against the same product through the eight-wave tile
    (the identity-affine epilogue form runs on GemmL with the same grid).  A mismatch raises -- there is no fallback."""
    key = (device.type, device.index)
    if key in _XL_CHECKED or torch.cuda.is_current_stream_capturing():
        return
    R, K, M = 256, 512, 4096
    info = (ctypes.c_int * 16)()
    lib.grafp_conv1x1_gemm_plan(R, K, 1, M, 1, info)
    if info[0] != 5:                                   # GM_CFG_XL (gemm.hip): the plan rules moved, the check would be vacuous
        raise RuntimeError(f"libgrafp_hip: the self-check product {R}x{K}x{M} is planned on tile configuration {info[0]}, not "
                           "on the four-wave tile (5): update ops._xl_selfcheck together with gemm_plan")
    g = torch.Generator(device="cpu").manual_seed(0)
    w = (torch.randn(R, K, generator=g) / K ** 0.5).to(device, torch.bfloat16)
    x = torch.randn(K, M, generator=g).to(device, torch.bfloat16)
    tab = torch.tensor([1.0, 0.0], device=device).repeat(R, 1, 1).contiguous()            # (R, views = 1, 2): identity
    y, z = torch.empty((R, M), dtype=torch.bfloat16, device=device), torch.empty((R, M), dtype=torch.bfloat16, device=device)
    ys = torch.empty((R, M), dtype=torch.bfloat16, device=device)
    P = lib.grafp_conv1x1_gemm_partials(R, K, 1, M, 1)
    part = torch.empty((R, 1, max(P, 1), 3), dtype=torch.float32, device=device)
    check(lib.grafp_conv1x1_gemm_bf16(_p(w), _p(x), R, K, 1, M, 1, None, 0, 0.0, _p(y), None, _stream()), "conv1x1_gemm")
    check(lib.grafp_conv1x1_gemm_bf16(_p(w), _p(x), R, K, 1, M, 1, None, 0, 0.0, _p(ys), _p(part), _stream()),
          "conv1x1_gemm")                                                                   # the <STATS> form of training
    check(lib.grafp_conv1x1_gemm_affine_bf16(_p(w), _p(x), R, K, 1, M, 1, _p(tab), 0, 0.0, _p(z), _stream()),
          "conv1x1_gemm_affine")
    ones, zeros = torch.ones(R, device=device), torch.zeros(R, device=device)
    # batch statistics out of the partial sums (training-mode finalize on scratch running statistics)
    mean, invstd, _ = bn_finalize(part, R, K, 1, M, 1, ones, zeros, None, zeros.clone(), ones.clone(), True, 0.1, 1e-5)
    zf = z.float()
    zd = z.double()
    want_mean, want_var = zd.mean(1), zd.var(1, unbiased=False)
    what = None
    for name, yf in (("product", y.float()), ("product with statistics", ys.float())):
        # (the two tiles may add the K chunks in another association: equal up to one bf16 rounding step on a few elements)
        wrong = bool(((yf - zf).abs() > zf.abs() * 2.0 ** -7 + float(zf.abs().max()) * 2.0 ** -15).any())
        if wrong or not bool(torch.isfinite(yf).all()) or float((yf != zf).float().mean()) > 5e-3:
            what = f"{name}: max |diff| {float((yf - zf).abs().max())}"
    if what is None:
        sd = want_var.sqrt()
        if not bool(((mean[:, 0].double() - want_mean).abs() <= 1e-3 * sd + 1e-6).all()) or \
                not bool(((invstd[:, 0].double() * (want_var + 1e-5).sqrt() - 1.0).abs() <= 2e-3).all()):
            what = "statistics epilogue: batch mean / variance of the rounded outputs"
    if what is not None:
        raise RuntimeError(f"libgrafp_hip: the four-wave GEMM tile disagrees with the eight-wave tile on a {R}x{K}x{M} product "
                           f"({what}): this build broke the register contract of gemm_xl.h -- rebuild with the "
                           "pinned toolchain (`make -C grafp_amd/csrc` runs tools/check_kernel_regs.py)")
    _XL_CHECKED.add(key)                               # only a check that PASSED is remembered: a caught error re-raises


def conv1x1_gemm(w, x, groups=1, views=1, pro_tab=None, pro_act=ACT_NONE, pro_slope=0.0, stats=False):
    """y = W f(x): w (R, K/groups) bf16, x (K, M) bf16 rows -> y (R, M) bf16 [, partial statistics (R, views, P, 3) f32].
    pro_tab (K, views, 2) f32: f(x) = act(x * scale + shift) applied to the operand tile on the fly (the BatchNorm +
    activation of the layer that produced x); stats: shifted sums of the rounded outputs for bn_finalize."""
    _require_gpu(w, x)
    if w.dtype != torch.bfloat16 or x.dtype != torch.bfloat16:
        raise TypeError("conv1x1_gemm: bf16 operands")
    _xl_selfcheck(x.device)
    w, x = w.contiguous(), x.contiguous()
    R, K, M = w.shape[0], x.shape[0], x.shape[1]
    if w.shape[1] * groups != K:
        raise ValueError(f"conv1x1_gemm: weight {tuple(w.shape)} x groups {groups} does not match {K} operand rows")
    y = torch.empty((R, M), dtype=torch.bfloat16, device=x.device)
    part = None
    if stats:
        P = lib.grafp_conv1x1_gemm_partials(R, K, groups, M, views)
        part = torch.empty((R, views, max(P, 1), 3), dtype=torch.float32, device=x.device)
    tab = None if pro_tab is None else _f32c(pro_tab)
    with _timed("conv1x1_gemm", (R, K, groups, M)):
        check(lib.grafp_conv1x1_gemm_bf16(_p(w), _p(x), R, K, groups, M, views, _p(tab), int(pro_act), float(pro_slope),
                                          _p(y), _p(part), _stream()), "conv1x1_gemm")
    return (y, part) if stats else y


def conv1x1_gemm_affine(w, x, tab, groups=1, views=1, act=ACT_NONE, slope=0.0):
    """z = act(bf16(W x) * scale + shift): the eval-mode [1x1 conv -> BatchNorm -> activation] chain in one kernel
    (tab (R, views, 2) from bn_finalize(training=False)); bit-identical to conv1x1_gemm + bn_affine."""
    _require_gpu(w, x, tab)
    if w.dtype != torch.bfloat16 or x.dtype != torch.bfloat16:
        raise TypeError("conv1x1_gemm_affine: bf16 operands")
    w, x = w.contiguous(), x.contiguous()
    R, K, M = w.shape[0], x.shape[0], x.shape[1]
    if w.shape[1] * groups != K:
        raise ValueError(f"conv1x1_gemm_affine: weight {tuple(w.shape)} x groups {groups} does not match {K} operand rows")
    z = torch.empty((R, M), dtype=torch.bfloat16, device=x.device)
    tab = _f32c(tab)
    with _timed("conv1x1_gemm", (R, K, groups, M)):
        check(lib.grafp_conv1x1_gemm_affine_bf16(_p(w), _p(x), R, K, groups, M, views, _p(tab), int(act), float(slope),
                                                 _p(z), _stream()), "conv1x1_gemm_affine")
    return z


def conv1x1_gemm_cat(w, x1, x2):
    """y = W [x1; x2] for bf16 rows x1 (K1, M), x2 (K2, M) and w (R, K1 + K2): the concatenation is never materialised."""
    _require_gpu(w, x1, x2)
    if w.dtype != torch.bfloat16 or x1.dtype != torch.bfloat16 or x2.dtype != torch.bfloat16:
        raise TypeError("conv1x1_gemm_cat: bf16 operands")
    w, x1, x2 = w.contiguous(), x1.contiguous(), x2.contiguous()
    R, K1, K2, M = w.shape[0], x1.shape[0], x2.shape[0], x1.shape[1]
    if w.shape[1] != K1 + K2 or x2.shape[1] != M:
        raise ValueError(f"conv1x1_gemm_cat: weight {tuple(w.shape)} vs operands {tuple(x1.shape)} + {tuple(x2.shape)}")
    y = torch.empty((R, M), dtype=torch.bfloat16, device=x1.device)
    with _timed("conv1x1_gemm", (R, K1 + K2, 1, M, K2)):        # K2 rows meet an identity block: moved, not multiplied
        check(lib.grafp_conv1x1_gemm_cat_bf16(_p(w), _p(x1), K1, _p(x2), K2, R, M, _p(y), _stream()), "conv1x1_gemm_cat")
    return y


class ShortcutToken:
    """Hands the shortcut's gradient of a residual block from the backward of its LAST layer (which receives dZ and
    would return it unchanged for the shortcut) to the backward of its FIRST layer (whose data gradient autograd would
    add it to): `x = f(x) + x` then costs no separate accumulate kernel -- see conv1x1_gemm_cat."""
    __slots__ = ("grad",)

    def __init__(self):
        self.grad = None


_EYES = {}


def _eye_bf16(n, device):
    key = (n, str(device))
    if key not in _EYES:
        _EYES[key] = torch.eye(n, dtype=torch.bfloat16, device=device)
    return _EYES[key]


def bn_finalize(part, C, K, groups, M, views, gamma, beta, pre_bias, running_mean, running_var, training, momentum, eps):
    """GEMM partial sums (or, in eval mode, the running statistics) -> (mean (C,views), invstd (C,views),
    tab (C,views,2) = (scale, shift) with z = act(y * scale + shift)); advances the running statistics when training."""
    dev = gamma.device
    mean = torch.empty((C, views), dtype=torch.float32, device=dev)
    invstd = torch.empty((C, views), dtype=torch.float32, device=dev)
    tab = torch.empty((C, views, 2), dtype=torch.float32, device=dev)
    g32, b32 = _f32c(gamma), _f32c(beta)
    pb = None if pre_bias is None else _f32c(pre_bias)
    check(lib.grafp_bn_finalize(_p(part), C, K, groups, M, views, _p(pb), _p(g32), _p(b32), float(eps), float(momentum),
                                int(bool(training)), _p(running_mean), _p(running_var), _p(mean), _p(invstd), _p(tab),
                                _stream()), "bn_finalize")
    return mean, invstd, tab


def bn_finalize_affine(y, part, K, groups, views, gamma, beta, pre_bias, running_mean, running_var, momentum, eps,
                       residual=None, act=ACT_NONE, slope=0.0):
    """bn_finalize (training) + bn_affine in ONE launch (every workgroup combines the partial sums of its row itself):
    -> (z, mean, invstd, tab).  y (C, M) bf16 and `part` from conv1x1_gemm(..., stats=True); 1 <= views <= 4."""
    _require_gpu(y, part)
    C, M = y.shape[0], y.numel() // y.shape[0]
    mean = torch.empty((C, views), dtype=torch.float32, device=y.device)
    invstd = torch.empty((C, views), dtype=torch.float32, device=y.device)
    tab = torch.empty((C, views, 2), dtype=torch.float32, device=y.device)
    g32, b32 = _f32c(gamma), _f32c(beta)
    pb = None if pre_bias is None else _f32c(pre_bias)
    res = None if residual is None else residual.to(torch.bfloat16).contiguous()
    z = torch.empty_like(y)
    with _timed("bn_affine", (C, M, res is not None)):
        check(lib.grafp_bn_finalize_affine_bf16(_p(y), _p(part), C, K, groups, M, views, _p(pb), _p(g32), _p(b32),
                                                float(eps), float(momentum), _p(running_mean), _p(running_var), _p(mean),
                                                _p(invstd), _p(tab), _p(res), int(act), float(slope), _p(z), _stream()),
              "bn_finalize_affine")
    return z, mean, invstd, tab


def bn_affine(y, tab, views=1, residual=None, act=ACT_NONE, slope=0.0):
    """z = act(y * scale + shift) + residual over bf16 (C, M) rows (one read [+ residual], one write)."""
    _require_gpu(y, tab)
    y = y.contiguous()
    C, M = y.shape[0], y.numel() // y.shape[0]
    res = None if residual is None else residual.to(torch.bfloat16).contiguous()
    out = torch.empty_like(y)
    with _timed("bn_affine", (C, M, res is not None)):
        check(lib.grafp_bn_affine_bf16(_p(y), C, M, views, _p(tab), _p(res), int(act), float(slope), _p(out), _stream()),
              "bn_affine")
    return out


def _group_transpose(wl, groups):
    """(R, K/g) grouped weight -> (K, R/g): per group the transposed block, the operand of the data-gradient GEMM."""
    R, Kg = wl.shape
    if groups == 1:
        return wl.t().contiguous()
    return wl.reshape(groups, R // groups, Kg).transpose(1, 2).reshape(groups * Kg, R // groups).contiguous()


def _bn_bwd(y, dz, C, M, views, pb, g32, b32, mean, invstd, act, slope, training):
    """BatchNorm + activation backward on (C, M) rows (grafp_bn_bwd_1pass): -> dy, dgamma, dbeta, dpre_bias."""
    dy = torch.empty_like(y)
    dgamma = torch.empty((C,), dtype=torch.float32, device=y.device)
    dbeta = torch.empty((C,), dtype=torch.float32, device=y.device)
    dpb = torch.empty((C,), dtype=torch.float32, device=y.device) if pb is not None else None
    nbytes = lib.grafp_bn_workspace(C, M)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=y.device)
    with _timed("bn_bwd", (C, M, y.element_size())):
        check(lib.grafp_bn_bwd_1pass(_p(y), _p(dz), _DT[y.dtype], C, M, views, _p(pb), _p(g32), _p(b32), _p(mean),
                                     _p(invstd), act, slope, int(training), _p(dy), _p(dgamma), _p(dbeta), _p(dpb),
                                     _p(ws), nbytes, _p(_bn_sync(y.device, C, M)), switches.bn_spin_limit, _stream()), "bn_bwd")
    return dy, dgamma, dbeta, dpb


# ---- the reductions of a step's weight gradients, deferred and batched (grafp_wgrad_reduce_multi) ----
_WGRAD_PENDING = None        # None: every weight gradient reduces its own partial sums at once; a list: they queue up


def flush_wgrad_reduce():
    """Reduce every queued weight gradient's partial sums now (one launch per 64 layers).  Called when the block below
    ends and by whoever reads a gradient before that (dist.GradSync packs a bucket while backward is still running)."""
    global _WGRAD_PENDING
    q = _WGRAD_PENDING
    _WGRAD_DEFERRED_IDS.clear()
    if not q:
        return
    _WGRAD_PENDING = []
    n = len(q)
    parts = (ctypes.c_void_p * n)(*[ws.data_ptr() for ws, _, _, _ in q])
    slices = (ctypes.c_int * n)(*[S for _, S, _, _ in q])
    sizes = (ctypes.c_int64 * n)(*[dw.numel() for _, _, dw, _ in q])
    outs = (ctypes.c_void_p * n)(*[dw.data_ptr() for _, _, dw, _ in q])
    with _timed("conv1x1_wgrad_reduce", (n,)):
        check(lib.grafp_wgrad_reduce_multi(parts, slices, sizes, outs, n, _stream()), "wgrad_reduce_multi")


@contextlib.contextmanager
def defer_wgrad_reduce():
    """Inside the block (a backward pass) the bf16 weight gradients only run their split-K kernels; the partial sums of
    all layers are reduced together when the block ends (or at flush_wgrad_reduce()): ~60 launches fewer per step, the
    same bits.  The returned gradient tensors must not be READ inside the block without a flush."""
    global _WGRAD_PENDING
    if _WGRAD_PENDING is not None:                # nested: the outer block flushes
        yield
        return
    _WGRAD_PENDING = []
    try:
        yield
        flush_wgrad_reduce()
    finally:
        _WGRAD_PENDING = None
        _WGRAD_DEFERRED_IDS.clear()


# Under data parallelism the parameter gradients live in ONE flat buffer (dist.GradSync) that the bucket all-reduces run on.
# GradSync.open() registers a function here that maps a parameter to its slice of that buffer (or None): the weight
# gradient of a layer is then reduced STRAIGHT into the slice -- autograd takes the tensor over as the parameter's .grad, and
# the bucket's pack step finds it in place (no 50 MB of multi-tensor copies per step at N = 8).
_GRAD_TARGET_OF = None


def _wgrad_bf16(g, x, cout, cin, groups, M, views=1, pro_tab=None, pro_act=ACT_NONE, pro_slope=0.0, tile=-1, may_defer=True,
                out=None):
    """dW = g f(x)^T for bf16 (rows, M) operands -> (cout, cin/groups) f32; pro_tab (cin, views, 2): f = the BatchNorm
    + activation of the layer that produced x, applied on the fly (see conv1x1_gemm).  tile: -1 = the library's rule,
    otherwise that tile configuration (grafp_conv1x1_wgrad_tile_bf16; tests).  out: a contiguous f32 tensor of
    cout * cin / groups elements on x's device to write the gradient into (see _GRAD_TARGET_OF)."""
    if out is not None and out.dtype == torch.float32 and out.is_contiguous() and out.device == x.device \
            and out.numel() == cout * (cin // groups):
        dw = out.view(cout, cin // groups)
    else:
        dw = torch.empty((cout, cin // groups), dtype=torch.float32, device=x.device)
    nbytes = lib.grafp_conv1x1_wgrad_tile_workspace(cout, cin, groups, M, views, int(tile))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    tab = None if pro_tab is None else _f32c(pro_tab)
    # may_defer = False: the gradient is READ inside this backward pass (the weight is not a leaf: Downsample's tap matrix
    # is a slice / permutation of its Conv2d parameter, and autograd scatters the gradient back through those views)
    if _WGRAD_PENDING is not None and may_defer:
        S = ctypes.c_int(0)
        with _timed("conv1x1_wgrad", (cout, cin, groups, M)):
            check(lib.grafp_conv1x1_wgrad_partials_bf16(_p(g), _p(x), cout, cin, groups, M, views, _p(tab), int(pro_act),
                                                        float(pro_slope), int(tile), _p(ws), nbytes, ctypes.byref(S),
                                                        _stream()), "conv1x1_wgrad_partials")
        _WGRAD_PENDING.append((ws, int(S.value), dw, tab))       # (the workspace stays alive until its reduction ran)
        return dw
    with _timed("conv1x1_wgrad", (cout, cin, groups, M)):
        check(lib.grafp_conv1x1_wgrad_tile_bf16(_p(g), _p(x), cout, cin, groups, M, views, _p(tab), int(pro_act),
                                                float(pro_slope), int(tile), _p(dw), _p(ws), nbytes, _stream()),
              "conv1x1_wgrad")
    return dw


conv1x1_wgrad = _wgrad_bf16


_WGRAD_DEFERRED_IDS = set()  # weights whose gradient is queued unreduced in the running defer_wgrad_reduce() block


def _may_defer(w):
    """A weight gradient may leave backward() with its split-K partial sums still unreduced (defer_wgrad_reduce) only if
    nothing can READ it before the flush: the weight is a leaf whose .grad is None (AccumulateGrad then takes the tensor
    over as it is -- with an existing .grad it would ADD the unreduced memory into it: gradient accumulation without
    zero()), it carries no tensor hook (a hook receives the gradient inside backward), every post-accumulate hook on it is
    GradSync's (which flushes before it packs a bucket; anybody else's would read unreduced memory), and it is the FIRST
    use of this weight in the pass (a weight shared by two layers gets its two gradients summed by autograd in front of
    AccumulateGrad: the second one -- and with it the first, flushed -- must be reduced by then).  Everything else reduces
    immediately: slower, never wrong."""
    if w is None or w.grad is not None or getattr(w, "_backward_hooks", None) or torch.is_grad_enabled():
        return False                                    # (create_graph = True keeps grad mode on inside backward)
    hooks = getattr(w, "_post_accumulate_grad_hooks", None)
    if hooks:
        from .dist import GradSync
        if not all(isinstance(getattr(h, "__self__", None), GradSync) for h in hooks.values()):
            return False
    if id(w) in _WGRAD_DEFERRED_IDS:
        flush_wgrad_reduce()                            # the first use's partial sums: reduced before autograd adds the two
        return False
    if _WGRAD_PENDING is not None:
        _WGRAD_DEFERRED_IDS.add(id(w))
    return True


class _ConvBnAct(torch.autograd.Function):
    """z = act(BatchNorm(W x + bias)) + residual on bf16 (C, M) rows, the bf16 training path of every
    [Conv2d(1x1) -> BatchNorm2d -> activation -> shortcut] chain: ONE hand-written GEMM whose epilogue also yields the
    batch statistics, a per-row finalize, one normalise/activate/add pass.  Backward: BatchNorm backward (single
    pass), data gradient by the same GEMM kernel on the transposed weight, weight gradient by the split-K kernel."""

    @staticmethod
    def forward(ctx, x, w, w_lowp, conv_groups, views, gamma, beta, pre_bias, residual, running_mean, running_var,
                training, momentum, eps, act, slope, token=None, token_role=0, w_t=None, w_aug=None, defer=None,
                defer_role=0, lowp_owner=None):
        x = x.detach()
        # DeferredNorm: role 1 = this layer's BatchNorm + activation are applied by its only consumer while THAT stages
        # its operand (no normalise pass, the normalised tensor is never written); role 2 = that consumer
        pro = None
        if defer is not None and defer_role == 2 and defer.tab is not None:
            pro = (defer.tab, int(defer.act), float(defer.slope))
        ctx.pro = pro
        ctx.w_leaf = bool(w.is_leaf)                         # its gradient goes straight to AccumulateGrad (never read here)
        ctx.w_param = w if w.is_leaf else None               # (looked at again in backward: see _may_defer)
        ctx.token, ctx.token_role = token, token_role        # 1: first layer of the block (consumes), 2: last (provides)
        ctx.w_t, ctx.w_aug = w_t, w_aug                      # prepared with the forward operand (lowp_weights), or None
        # shared prepared buffers in use: remember which preparation this forward pass saw
        shared = lowp_owner is not None and (w_lowp is not None or w_t is not None or w_aug is not None)
        ctx.lowp_owner, ctx.lowp_gen = (lowp_owner, lowp_owner.generation) if shared else (None, None)
        K, M = x.shape
        R = w.shape[0]
        if w_lowp is not None and w_lowp.dtype == torch.bfloat16 and w_lowp.numel() == w.numel():
            wl = w_lowp.detach().reshape(R, -1)
        else:
            wl = w.detach().reshape(R, -1).to(torch.bfloat16)
        g32, b32 = _f32c(gamma), _f32c(beta)
        pb = None if pre_bias is None else _f32c(pre_bias)
        if pro is not None:
            if not training:
                raise RuntimeError("conv_bn_act: a deferred normalisation reached an eval-mode consumer")
            y, part = conv1x1_gemm(wl, x, conv_groups, views, pro_tab=pro[0], pro_act=pro[1], pro_slope=pro[2], stats=True)
        elif training:
            y, part = conv1x1_gemm(wl, x, conv_groups, views, stats=True)
        else:
            y, part = conv1x1_gemm(wl, x, conv_groups, views), None
        res = None if residual is None else residual.detach().to(torch.bfloat16).contiguous()
        if defer is not None and defer_role == 1:
            if not training or res is not None:
                raise RuntimeError("conv_bn_act: only a training-mode layer without a shortcut can defer its normalisation")
            mean, invstd, tab = bn_finalize(part, R, K, conv_groups, M, views, g32, b32, pb, running_mean, running_var,
                                            True, momentum, eps)
            defer.tab, defer.act, defer.slope = tab, act, slope
            z = y                                            # stands in for act(BN(y)): the consumer normalises on load
        elif training and views <= 4:
            z, mean, invstd, tab = bn_finalize_affine(y, part, K, conv_groups, views, g32, b32, pb, running_mean,
                                                      running_var, momentum, eps, res, act, slope)
        else:
            mean, invstd, tab = bn_finalize(part, R, K, conv_groups, M, views, g32, b32, pb, running_mean, running_var,
                                            training, momentum, eps)
            z = bn_affine(y, tab, views, res, act, slope)
        ctx.save_for_backward(x, wl, y, mean, invstd, g32, b32, pb if pb is not None else mean.new_empty(0))
        ctx.cfg = (R, K, M, conv_groups, views, act, float(slope), bool(training), pb is not None, residual is not None,
                   tuple(w.shape))
        return z

    @staticmethod
    def backward(ctx, dz):
        x, wl, y, mean, invstd, g32, b32, pb = ctx.saved_tensors
        R, K, M, cg, views, act, slope, training, has_pb, has_res, wfull = ctx.cfg
        if ctx.lowp_owner is not None and ctx.lowp_gen != ctx.lowp_owner.generation:
            raise RuntimeError("conv_bn_act backward: the shared low-precision weight buffers were re-prepared (another forward "
                               "of THIS encoder under autocast, possibly after an optimizer step) between this layer's "
                               "forward and backward pass; run backward before the next forward")
        dz = dz.detach().to(torch.bfloat16).contiguous()
        dy, dgamma, dbeta, dpb = _bn_bwd(y, dz, R, M, views, pb if has_pb else None, g32, b32, mean, invstd, act, slope,
                                         training)
        dx = None
        tok = ctx.token
        if ctx.needs_input_grad[0]:
            short = tok.grad if (tok is not None and ctx.token_role == 1) else None
            w_t, w_aug = ctx.w_t, ctx.w_aug
            if short is not None:
                # dX = [W^T | I] [dY; dZ_shortcut]: the block input's two gradients in one product, rounded once
                tok.grad = None
                if w_aug is None:
                    w_aug = torch.cat((wl.t(), _eye_bf16(K, wl.device)), dim=1)
                dx = conv1x1_gemm_cat(w_aug, dy, short)
            elif w_t is not None and w_t.is_contiguous():
                dx = conv1x1_gemm(w_t, dy, cg, 1)
            else:
                dx = conv1x1_gemm(_group_transpose(wl, cg), dy, cg, 1)
        dw = None
        if ctx.needs_input_grad[1]:
            md = _may_defer(ctx.w_param)
            # (a deferred gradient is the FIRST of this weight in the pass and nothing reads it before the flush: it may be
            #  produced in place, in the parameter's slice of the data-parallel flat buffer)
            tgt = _GRAD_TARGET_OF(ctx.w_param) if (md and _WGRAD_PENDING is not None and _GRAD_TARGET_OF is not None) else None
            if ctx.pro is not None:                        # x is the producer's raw output: the same transform on load
                dw = _wgrad_bf16(dy, x, R, K, cg, M, views, ctx.pro[0], ctx.pro[1], ctx.pro[2], may_defer=md,
                                 out=tgt).reshape(wfull)
            else:
                dw = _wgrad_bf16(dy, x, R, K, cg, M, may_defer=md, out=tgt).reshape(wfull)
        dres = dz if has_res else None
        if has_res and tok is not None and ctx.token_role == 2 and tok.grad is None:
            tok.grad, dres = dz, None                      # the first layer's backward adds it (see above)
        return (dx, dw, None, None, None, dgamma, dbeta, dpb, dres, None, None, None, None, None, None, None, None, None,
                None, None, None, None, None)


def conv_bn_act_supported(x, cout, conv_groups, views):
    """The fused bf16 path applies to HIP bf16 (K, M) rows whose shape both GEMMs (forward, data gradient) accept."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2):
        return False
    K, M = x.shape
    return conv_bn_act_shape_supported(K, M, cout, conv_groups, views)


def conv_bn_act_shape_supported(K, M, cout, conv_groups, views):
    if not switches.fused_conv_bn:
        return False
    return gemm_supported(cout, K, conv_groups, M, views) and gemm_supported(K, cout, conv_groups, M, 1)


class DeferredNorm:
    """Shared by a layer whose BatchNorm + activation output has exactly ONE consumer, another conv_bn_act (role 1), and
    that consumer (role 2): the producer returns its raw convolution output and leaves (scale, shift) here, the consumer
    applies them to the operand tile while it is staged (conv1x1_gemm PRO, and again in its weight gradient).  Saves the
    normalise pass (one read + one write of the widest tensors of a block) where the GEMM is HBM-bound."""

    def __init__(self):
        self.tab, self.act, self.slope = None, ACT_NONE, 0.0


def defer_norm_pays(consumer_rows, consumer_operand_rows, M):
    """Measured at 512 and 2048 clip-views: the consumer's output rows <= 128 (stages 0-1): skipped
    pass 200-400 us against +30-115 us in the GEMM and +10-35 us in the weight gradient; beyond, the products are
    matrix-bound and the in-LDS transform costs more than the pass."""
    return bool(switches.defer_norm) and consumer_rows <= 128


def conv_bn_act(x, w, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, pre_bias=None,
                residual=None, act=ACT_NONE, slope=0.0, conv_groups=1, views=1, w_lowp=None, token=None, token_role=0,
                w_t=None, w_aug=None, defer=None, defer_role=0, lowp_owner=None):
    """act(BatchNorm(W x + pre_bias)) + residual for bf16 (K, M) rows (see _ConvBnAct).  token / token_role: a
    ShortcutToken shared by the first (role 1: its input IS the shortcut) and the last layer (role 2: `residual` is that
    same input) of a residual block."""
    if not training and residual is None and switches.fused_eval_affine and not torch.is_grad_enabled():
        # inference without a shortcut (fc1, the grouped conv, ffn1, Downsample): the normalisation rides the GEMM's
        # epilogue -- no y, no second pass, no autograd node
        x = x.detach().contiguous()
        K, M = x.shape
        R = w.shape[0]
        if w_lowp is not None and w_lowp.dtype == torch.bfloat16 and w_lowp.numel() == w.numel():
            wl = w_lowp.detach().reshape(R, -1)
        else:
            wl = w.detach().reshape(R, -1).to(torch.bfloat16)
        _, _, tab = bn_finalize(None, R, K, int(conv_groups), M, int(views), gamma, beta, pre_bias, running_mean,
                                running_var, False, momentum, eps)
        return conv1x1_gemm_affine(wl, x, tab, int(conv_groups), int(views), int(act), float(slope))
    if defer is not None and not training:
        raise RuntimeError("conv_bn_act: deferred normalisation is a training-mode construct")
    return _ConvBnAct.apply(x.contiguous(), w, w_lowp, int(conv_groups), int(views), gamma, beta, pre_bias, residual,
                            running_mean, running_var, bool(training), float(momentum), float(eps), int(act),
                            float(slope), token, int(token_role), w_t, w_aug, defer, int(defer_role), lowp_owner)


def shortcut_token_supported(x, conv_groups=1):
    """The fused shortcut gradient needs the (K, K + K) x M data-gradient product of the block's first layer."""
    if not switches.shortcut_fusion or conv_groups != 1:
        return False
    K, M = x.shape
    return x.is_cuda and x.dtype == torch.bfloat16 and bool(lib.grafp_conv1x1_gemm_supported(K, 2 * K, 1, M, 1))


# ------------------------------------------------------------------------------------------------
# K12  NT-Xent
# ------------------------------------------------------------------------------------------------
class _NTXent(torch.autograd.Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, z_i, z_j, tau, zi_all, zj_all, row_begin):
        _require_gpu(z_i, z_j)
        n_local, D = z_i.shape
        if zi_all is None:
            zi_all, zj_all, row_begin = z_i, z_j, 0
        zi_all, zj_all = _f32c(zi_all), _f32c(zj_all)
        B_all = zi_all.shape[0]
        npart = lib.grafp_ntxent_num_partials(n_local)
        part = torch.empty((npart,), dtype=torch.float32, device=z_i.device)
        dzi = torch.empty((n_local, D), dtype=torch.float32, device=z_i.device)
        dzj = torch.empty((n_local, D), dtype=torch.float32, device=z_i.device)
        nbytes = lib.grafp_ntxent_workspace(B_all)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=z_i.device)
        with _timed("ntxent", (B_all, n_local, D)):
            check(lib.grafp_ntxent_fwd_bwd_f32(_p(zi_all), _p(zj_all), B_all, D, row_begin, n_local, float(tau),
                                               _p(part), _p(dzi), _p(dzj), _p(ws), nbytes, _stream()), "ntxent")
        ctx.save_for_backward(dzi, dzj)
        return part.sum() / (2.0 * B_all)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g):
        dzi, dzj = ctx.saved_tensors
        return g * dzi, g * dzj, None, None, None, None


def ntxent(z_i, z_j, tau, zi_all=None, zj_all=None, row_begin=0):
    """NT-Xent of (z_i, z_j) (each (B,D)).  With zi_all/zj_all (the all-gathered embeddings, no grad) the
    negatives are global: the value is this rank's share of the global mean loss and backward yields the
    gradient of the GLOBAL loss w.r.t. the local rows."""
    return _NTXent.apply(z_i, z_j, float(tau), zi_all, zj_all, int(row_begin))


# ------------------------------------------------------------------------------------------------
# K13  brute-force search
# ------------------------------------------------------------------------------------------------
def row_sqnorm(m):
    _require_gpu(m)
    m = _f32c(m)
    out = torch.empty((m.shape[0],), dtype=torch.float32, device=m.device)
    check(lib.grafp_row_sqnorm_f32(_p(m), m.shape[0], m.shape[1], _p(out), _stream()), "row_sqnorm")
    return out


def rows_to_bf16(db):
    """Round-to-nearest-even bf16 copy of the resident (n,128) f32 database for the pre-filter scan."""
    _require_gpu(db)
    db = _f32c(db)
    out = torch.empty(db.shape, dtype=torch.bfloat16, device=db.device)
    check(lib.grafp_f32_to_bf16(_p(db), db.numel(), _p(out), _stream()), "f32_to_bf16")
    return out


def search_l2(db, db_sqnorm, q, k, id_base=0, max_queries_per_launch=4096, db_bf16=None):
    """Exact squared-L2 top-k of q (nq,128) against the resident db (n,128): (dist f32, ids int64), (nq,k).
    db_bf16 (rows_to_bf16(db)): same results through the bf16 pre-filter scan + exact f32 rescoring."""
    _require_gpu(db, db_sqnorm, q)
    q = _f32c(q)
    n, d = db.shape
    nq = q.shape[0]
    out_d = torch.empty((nq, k), dtype=torch.float32, device=db.device)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=db.device)
    pre = db_bf16 is not None
    if pre and (db_bf16.dtype != torch.bfloat16 or db_bf16.shape != db.shape or not db_bf16.is_contiguous()):
        raise ValueError("search_l2: db_bf16 must be a contiguous bf16 tensor of db's shape")
    for s in range(0, nq, max_queries_per_launch):
        e = min(nq, s + max_queries_per_launch)
        nbytes = (lib.grafp_knn_search_pre_workspace if pre else lib.grafp_knn_search_workspace)(n, e - s, d, k)
        ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=db.device)
        with _timed("knn_search", (n, e - s, k)):
            if pre:
                check(lib.grafp_knn_search_l2_pre(_p(db), _p(db_bf16), _p(db_sqnorm), n, _p(q[s:e]), e - s, d, k,
                                                  int(id_base), _p(out_d[s:e]), _p(out_i[s:e]), _p(ws), nbytes,
                                                  _stream()), "knn_search_l2_pre")
            else:
                check(lib.grafp_knn_search_l2_f32(_p(db), _p(db_sqnorm), n, _p(q[s:e]), e - s, d, k, int(id_base),
                                                  _p(out_d[s:e]), _p(out_i[s:e]), _p(ws), nbytes, _stream()),
                      "knn_search_l2")
    return out_d, out_i


def merge_topk(part_d, part_i):
    """(P,nq,k) partial lists (id < 0 = empty) -> (nq,k) by (distance, id)."""
    _require_gpu(part_d, part_i)
    part_d = _f32c(part_d)
    part_i = part_i.to(torch.int64).contiguous()
    P, nq, k = part_d.shape
    out_d = torch.empty((nq, k), dtype=torch.float32, device=part_d.device)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=part_d.device)
    with _timed("merge_topk", (P, nq, k)):
        check(lib.grafp_merge_topk(_p(part_d), _p(part_i), P, nq, k, _p(out_d), _p(out_i), _stream()), "merge_topk")
    return out_d, out_i


def seq_rerank(index_rows, q_rows, topk_ids, item_row, item_len, top=10, shard=None, max_len=None):
    """Sequence-level rerank of batched segment-search results (eval.py:272-290, one workgroup per item).
    index_rows (n,128) f32 resident database, q_rows (n_q,128) f32, topk_ids (n_q,k) int64, item_row (n_items) int64
    first query row of each item, item_len (n_items) int32 segments per item.
    Returns (ids int64 (n_items, top) best first, -1 padded; scores f32 (n_items, top), -inf padded).
    shard = (row_base, n_total, id_lo, id_hi): index_rows holds global rows [row_base, row_base + len) of an n_total-row
    index and only candidates starting in [id_lo, id_hi) are scored (dist.ShardedFlatL2Index.rerank merges shards).
    max_len: the longest item (segments), an upper bound is fine; the caller then also vouches that every item lies
    inside q_rows (the kernel clamps nothing)."""
    _require_gpu(index_rows, q_rows, topk_ids, item_row, item_len)
    index_rows, q_rows = _f32c(index_rows), _f32c(q_rows)
    topk_ids = topk_ids.to(torch.int64).contiguous()
    item_row = item_row.to(torch.int64).contiguous()
    item_len = item_len.to(torch.int32).contiguous()
    n_items = item_row.shape[0]
    out_i = torch.empty((n_items, top), dtype=torch.int64, device=index_rows.device)
    out_s = torch.empty((n_items, top), dtype=torch.float32, device=index_rows.device)
    if n_items == 0:
        return out_i, out_s
    if max_len is None:
        # convenience path with three host synchronisations (length bound + range check); callers that know the longest
        # sequence pass it (eval.py, dist.py, bench.py do) and stay asynchronous / graph-capturable like the other ops
        max_len = int(item_len.max().item())
        if int((item_row + item_len.to(torch.int64)).max().item()) > q_rows.shape[0] or int(item_row.min().item()) < 0:
            raise ValueError("seq_rerank: an item reaches outside q_rows")
    max_len = int(max_len)
    with _timed("seq_rerank", (n_items, max_len, topk_ids.shape[1])):
        if shard is None:
            check(lib.grafp_seq_rerank_f32(_p(index_rows), index_rows.shape[0], _p(q_rows), q_rows.shape[0],
                                           _p(topk_ids), topk_ids.shape[1], _p(item_row), _p(item_len), n_items, max_len,
                                           top, _p(out_i), _p(out_s), _stream()), "seq_rerank")
        else:
            row_base, n_total, id_lo, id_hi = (int(v) for v in shard)
            check(lib.grafp_seq_rerank_shard_f32(_p(index_rows), index_rows.shape[0], row_base, n_total, id_lo, id_hi,
                                                 _p(q_rows), q_rows.shape[0], _p(topk_ids), topk_ids.shape[1],
                                                 _p(item_row), _p(item_len), n_items, max_len, top, _p(out_i),
                                                 _p(out_s), _stream()), "seq_rerank_shard")
    return out_i, out_s


class FlatL2Index:
    """Drop-in for the subset of faiss.IndexFlatL2 that eval.py uses: d, ntotal, add(x), search(q, k).
    The database lives in HBM; `add` also computes the per-row squared norms once."""

    def __init__(self, d=128, device=None, id_base=0, prefilter=True):
        if not torch.cuda.is_available():
            raise RuntimeError("FlatL2Index needs a HIP device: there is no CPU fallback")
        self.d = d
        self.prefilter = bool(prefilter)     # bf16 pre-filter scan + exact rescoring (same results, 2-4x faster)
        self._bf16 = None
        self.device = torch.device(device if device is not None else "cuda")
        self.id_base = int(id_base)
        self._chunks = []
        self._db = None
        self._sq = None
        self.ntotal = 0
        self.nprobe = 1            # attribute assigned at eval.py:122; meaningless for exact search

    def add(self, x):
        t = torch.as_tensor(np.ascontiguousarray(x) if isinstance(x, np.ndarray) else x)
        t = t.to(self.device, dtype=torch.float32).reshape(-1, self.d).contiguous()
        self._chunks.append(t)
        self.ntotal += t.shape[0]
        self._db = None

    def _materialise(self):
        if self._db is None:
            self._db = self._chunks[0] if len(self._chunks) == 1 else torch.cat(self._chunks, dim=0)
            self._chunks = [self._db]
            self._sq = row_sqnorm(self._db)
            self._bf16 = rows_to_bf16(self._db) if self.prefilter else None
        return self._db, self._sq

    def rows(self):
        """The resident (ntotal, d) f32 database tensor (what faiss calls reconstruct_n(0, ntotal))."""
        return self._materialise()[0]

    def search(self, q, k):
        """numpy in -> numpy out (D float32 (nq,k), I int64 (nq,k)), like faiss; tensors in -> tensors out."""
        as_numpy = isinstance(q, np.ndarray)
        qt = torch.as_tensor(np.ascontiguousarray(q) if as_numpy else q).to(self.device, dtype=torch.float32)
        if self.ntotal == 0:
            D = torch.full((qt.shape[0], k), float("inf"), device=self.device)
            I = torch.full((qt.shape[0], k), -1, dtype=torch.int64, device=self.device)
        else:
            db, sq = self._materialise()
            D, I = search_l2(db, sq, qt.reshape(-1, self.d), k, self.id_base, db_bf16=self._bf16)
        return (D.cpu().numpy(), I.cpu().numpy()) if as_numpy else (D, I)
