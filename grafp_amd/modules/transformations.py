"""On-device preprocessing (mirror of modules/transformations.py:9-116).

log-mel and whole-track segmentation are HIP kernels (ops.logmel / ops.unfold_segments).  The IR / background-noise
augmentation chains (:25-48, torch_audiomentations) run ON THE DEVICE for the whole batch (ops.ir_convolve /
ops.mix_snr, csrc/augment.hip) instead of clip by clip on DataLoader workers (:67-75): `ir_dir` / `noise_dir` name
the recordings (a list of files, a directory, or an array/tensor bank), which are decoded once into ragged banks (one flat buffer,
no padding to the longest file) resident in HBM; every step draws, per clip and on the device, whether each transform applies (ir_prob / noise_prob), which
recording, the noise offset and the SNR (uniform in tr_snr / val_snr dB).  With both unset the transforms are
identities, as in the reference.  Decoding is limited to what the image can do without torchaudio: `.npy` arrays and
PCM `.wav` files (stdlib `wave`) already at cfg['fs'].
"""
import glob
import os
import wave

import numpy as np
import torch
from torch import nn

from .. import ops


def _read_audio(path, fs):
    """One mono f32 recording at sample rate fs from a .npy array or a PCM .wav file."""
    if path.endswith(".npy"):
        return np.asarray(np.load(path), dtype=np.float32).reshape(-1)
    with wave.open(path, "rb") as w:
        if w.getframerate() != fs:
            raise ValueError(f"{path}: sample rate {w.getframerate()} != cfg['fs'] = {fs} (no resampler on this path)")
        width, ch, n = w.getsampwidth(), w.getnchannels(), w.getnframes()
        raw = w.readframes(n)
    if width == 2:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        a = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"{path}: unsupported PCM sample width {width}")
    return a.reshape(-1, ch).mean(axis=1) if ch > 1 else a


def load_bank(src, fs, max_len=None):
    """Recordings -> ragged bank on the CPU: (flat f32 buffer, starts (n) int64, lengths (n) int32); recording r is
    flat[starts[r] : starts[r] + lengths[r]] (no padding to the longest file).
    src: directory (searched recursively for .wav/.npy), list of files, 2-D array/tensor (rows = recordings), or a
    list of 1-D arrays.  max_len truncates every recording."""
    if isinstance(src, torch.Tensor):
        src = src.detach().cpu().numpy()
    if isinstance(src, np.ndarray):
        rows = [np.asarray(r, dtype=np.float32).reshape(-1) for r in (src if src.ndim == 2 else [src])]
    else:
        if isinstance(src, str):
            if os.path.isdir(src):
                src = sorted(glob.glob(os.path.join(src, "**", "*.wav"), recursive=True) +
                             glob.glob(os.path.join(src, "**", "*.npy"), recursive=True))
            else:
                src = [src]
        rows = [(_read_audio(r, fs) if isinstance(r, str) else np.asarray(r, dtype=np.float32).reshape(-1)) for r in src]
    rows = [r[:max_len] if max_len else r for r in rows if r.size > 0]
    if not rows:
        raise ValueError("no recordings found for the augmentation bank")
    lens = np.array([r.size for r in rows], dtype=np.int32)
    starts = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.int64)]).astype(np.int64)
    return torch.from_numpy(np.concatenate(rows)), torch.from_numpy(starts), torch.from_numpy(lens)


class GPUTransformNeuralfp(nn.Module):
    def __init__(self, cfg, ir_dir, noise_dir, train=True, cpu=False, abl=False):
        super().__init__()
        self.sample_rate = cfg["fs"]
        self.ir_dir, self.noise_dir = ir_dir, noise_dir
        self.overlap, self.arch, self.n_frames = cfg["overlap"], cfg["arch"], cfg["n_frames"]
        self.train, self.cpu, self.cfg, self.abl = train, cpu, cfg, abl   # `train` shadows nn.Module.train as in the reference
        self.seed, self._gen = cfg.get("aug_seed"), None      # optional: reproducible augmentation draws
        has_ir = ir_dir is not None and not (isinstance(ir_dir, (list, tuple, str)) and len(ir_dir) == 0)
        has_noise = noise_dir is not None and not (isinstance(noise_dir, (list, tuple, str)) and len(noise_dir) == 0)
        if has_ir:
            bank, starts, lens = load_bank(ir_dir, cfg["fs"])
            self.register_buffer("ir_bank", bank, persistent=False)
            self.register_buffer("ir_start", starts, persistent=False)
            self.register_buffer("ir_len", lens, persistent=False)
        else:
            self.ir_bank = self.ir_start = self.ir_len = None
        if has_noise:
            bank, starts, lens = load_bank(noise_dir, cfg["fs"])
            self.register_buffer("noise_bank", bank, persistent=False)
            self.register_buffer("noise_start", starts, persistent=False)
            self.register_buffer("noise_len", lens, persistent=False)
        else:
            self.noise_bank = self.noise_start = self.noise_len = None

    # ---- the two torch_audiomentations chains (:25-48), batched on the device -------------------------
    def _rng(self, device):
        """Generator of the per-clip draws: torch's default device generator (advances correctly under HIP-graph
        capture and replay), or -- with cfg['aug_seed'] -- a private seeded one (reproducible draws; eager only)."""
        if self.seed is None:
            return None
        if self._gen is None or self._gen.device != device:
            self._gen = torch.Generator(device=device)
            self._gen.manual_seed(int(self.seed))
        return self._gen

    def augment(self, x, ir_prob, noise_prob, snr_range):
        """Compose([ApplyImpulseResponse(p=ir_prob), AddBackgroundNoise(snr_range, p=noise_prob)]) on x (B,T) or
        (T,): per signal, independent draws of apply/skip, recording, noise offset and SNR."""
        if self.ir_bank is None and self.noise_bank is None:
            return x
        squeeze = x.dim() == 1
        x = x.reshape(1, -1) if squeeze else x.reshape(-1, x.shape[-1])
        dev, B = x.device, x.shape[0]
        g = self._rng(dev)
        if self.ir_bank is not None:
            pick = torch.randint(0, self.ir_len.numel(), (B,), device=dev, generator=g)
            keep = torch.rand(B, device=dev, generator=g) < ir_prob
            x = ops.ir_convolve(x, self.ir_bank, self.ir_len, torch.where(keep, pick, torch.full_like(pick, -1)), self.ir_start)
        if self.noise_bank is not None:
            pick = torch.randint(0, self.noise_len.numel(), (B,), device=dev, generator=g)
            keep = torch.rand(B, device=dev, generator=g) < noise_prob
            off = (torch.rand(B, device=dev, generator=g) * self.noise_len[pick]).long().clamp_(min=0)
            lo, hi = float(snr_range[0]), float(snr_range[1])
            snr = lo + (hi - lo) * torch.rand(B, device=dev, generator=g)
            x = ops.mix_snr(x, self.noise_bank, self.noise_len, torch.where(keep, pick, torch.full_like(pick, -1)), off, snr,
                            self.noise_start)
        return x[0] if squeeze else x

    def _has_banks(self):
        return self.ir_bank is not None or self.noise_bank is not None

    def train_transform(self, x):
        if not self._has_banks():
            return x
        return self.augment(x, self.cfg["ir_prob"], self.cfg["noise_prob"], self.cfg["tr_snr"])

    def val_transform(self, x):
        return self.augment(x, 1.0, 1.0, self.cfg["val_snr"]) if self._has_banks() else x

    def ablation(self, x):
        return self.augment(x, 0.0, 1.0, self.cfg["val_snr"]) if self._has_banks() else x

    def logmelspec(self, x):
        c = self.cfg
        return ops.logmel(x, c["fs"], c["n_fft"], c["win_len"], c["hop_len"], c["n_mels"])

    def _segments(self, track):
        step = int(self.n_frames * (1 - self.overlap))
        return ops.unfold_segments(self.logmelspec(track.reshape(-1)), self.n_frames, step)

    def forward(self, x_i, x_j):
        if self.cpu:                      # DataLoader-worker branch (:67-75): the augmentation of a clip is deferred
            # to the batched device transform of the train branch below (workers never touch the GPU)
            return x_i, x_j.flatten()[:int(self.sample_rate * self.cfg["dur"])]
        if self.train:                    # (:77-85) both views -> (B, n_mels, n_frames)
            return self.logmelspec(x_i), self.logmelspec(self.train_transform(x_j))
        X_i = self._segments(x_i)         # (:87-113) whole track -> (n_seg, n_mels, n_frames)
        if x_j is None:
            return X_i, X_i
        x_j = self.ablation(x_j.reshape(-1)) if self.abl else self.val_transform(x_j.reshape(-1))
        return X_i, self._segments(x_j)
