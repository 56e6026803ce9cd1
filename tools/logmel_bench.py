"""log-mel kernel alone: microseconds per call and bytes / time against the HBM roofline (64 000 B in + 8 192 B out per clip)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd import ops
from tools.gemm_bench import timeit
for B in (256, 512, 2048, 4096):
    x = 0.1 * torch.randn(B, 16000, device="cuda:0")
    t = timeit(lambda: ops.logmel(x), reps=50)
    by = B * (64000.0 + 8192.0)
    print(f"{B:5d} clips: {t:7.1f} us  {by / t / 1e6:6.3f} TB/s = {by / t / 1e6 / 8.0:5.3f} of 8 TB/s", flush=True)
