"""NT-Xent loss (mirror of simclr/ntxent.py:4-29): one fused HIP forward+backward instead of a 2B-step
Python loop over a materialised similarity matrix."""
from .. import ops


def ntxent_loss(z_i, z_j, cfg):
    """z_i, z_j (B, d) -> 0-d loss tensor (differentiable); reads cfg['tau']."""
    return ops.ntxent(z_i, z_j, cfg["tau"])
