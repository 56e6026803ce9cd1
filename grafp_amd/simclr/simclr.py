"""SimCLR wrapper (mirror of simclr/simclr.py:8-47): peak extractor -> encoder -> projector -> L2 norm.

The reference sends the two views through the same modules one after the other (BatchNorm statistics per view,
running statistics advanced twice).  Here both views travel as ONE stacked batch: every per-clip kernel (peak
extractor, k-NN graph, max-relative) is indifferent to it, every GEMM sees twice the columns (one weight-gradient
GEMM per layer instead of two, no gradient-accumulation adds), and the fused BatchNorm kernel keeps the per-view
statistics and the two running-stat updates (`groups=2`) -- same numbers, half the launches.  Set
`fuse_views = False` for the literal sequential order."""
import torch
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

    fuse_views = True

    def embed(self, x, views=1):
        if self.peak_extractor is not None:
            x = self.peak_extractor(x)
        h = self.encoder(x, views=views) if views > 1 else self.encoder(x)
        # the projector head is two (B x 1024 x 128)-sized products: it stays in f32 under bf16 autocast
        with torch.autocast(h.device.type, enabled=False):
            return h, F.normalize(self.projector(h.float()), p=2)

    def forward(self, x_i, x_j):
        if self.fuse_views and x_i.shape == x_j.shape and x_i.shape[0] > 0:
            B = x_i.shape[0]
            h, z = self.embed(torch.cat((x_i, x_j), dim=0), views=2)
            return h[:B], h[B:], z[:B], z[B:]
        h_i, z_i = self.embed(x_i)
        h_j, z_j = self.embed(x_j)
        return h_i, h_j, z_i, z_j
