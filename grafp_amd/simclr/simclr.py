"""SimCLR wrapper (mirror of simclr/simclr.py:8-47): peak extractor -> encoder -> projector -> L2 norm,
the two views one after the other through the same modules (so BatchNorm statistics are per view)."""
import torch.nn.functional as F
from torch import nn

from ..peak_extractor import GPUPeakExtractorv2


class SimCLR(nn.Module):
    def __init__(self, cfg, encoder):
        super().__init__()
        self.encoder = encoder
        self.cfg = cfg
        d, h, u = cfg["d"], cfg["h"], cfg["u"]
        self.peak_extractor = GPUPeakExtractorv2(cfg) if cfg["arch"] == "grafp" else None
        self.projector = nn.Sequential(nn.Linear(h, d * u), nn.ELU(), nn.Linear(d * u, d))

    def embed(self, x):
        if self.peak_extractor is not None:
            x = self.peak_extractor(x)
        h = self.encoder(x)
        return h, F.normalize(self.projector(h), p=2)

    def forward(self, x_i, x_j):
        h_i, z_i = self.embed(x_i)
        h_j, z_j = self.embed(x_j)
        return h_i, h_j, z_i, z_j
