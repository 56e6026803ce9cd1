"""On-device preprocessing (mirror of modules/transformations.py:9-116).

log-mel and whole-track segmentation are HIP kernels (ops.logmel / ops.unfold_segments).  The IR /
background-noise augmentation chains (:25-48) depend on torch_audiomentations and the MUSAN / AIR corpora
and are out of scope: with ir_dir and noise_dir unset they are identities, as in the
reference; if either is given this raises instead of silently skipping augmentation.
"""
import torch
from torch import nn

from .. import ops


class GPUTransformNeuralfp(nn.Module):
    def __init__(self, cfg, ir_dir, noise_dir, train=True, cpu=False, abl=False):
        super().__init__()
        self.sample_rate = cfg["fs"]
        self.ir_dir, self.noise_dir = ir_dir, noise_dir
        self.overlap, self.arch, self.n_frames = cfg["overlap"], cfg["arch"], cfg["n_frames"]
        self.train, self.cpu, self.cfg, self.abl = train, cpu, cfg, abl   # `train` shadows nn.Module.train as in the reference
        if ir_dir or noise_dir:
            raise NotImplementedError(
                "IR / background-noise augmentation needs torch_audiomentations and the MUSAN/AIR data; "
                "only the ir_dir=None, noise_dir=None (identity) configuration is provided")

    def logmelspec(self, x):
        c = self.cfg
        return ops.logmel(x, c["fs"], c["n_fft"], c["win_len"], c["hop_len"], c["n_mels"])

    def _segments(self, track):
        step = int(self.n_frames * (1 - self.overlap))
        return ops.unfold_segments(self.logmelspec(track.reshape(-1)), self.n_frames, step)

    def forward(self, x_i, x_j):
        if self.cpu:                      # DataLoader-worker branch (:67-75): identity augmentation
            return x_i, x_j.flatten()[:int(self.sample_rate * self.cfg["dur"])]
        if self.train:                    # (:77-85) both views -> (B, n_mels, n_frames)
            return self.logmelspec(x_i), self.logmelspec(x_j)
        X_i = self._segments(x_i)         # (:87-113) whole track -> (n_seg, n_mels, n_frames)
        if x_j is None:
            return X_i, X_i
        return X_i, self._segments(x_j)
