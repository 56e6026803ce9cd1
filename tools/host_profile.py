"""Where does the HOST time of one eager 128-pair training step go?  (the eager step is host-bound at small batches)
    python tools/host_profile.py [pairs]   -> cProfile of 5 steps, top functions by own time and by cumulative time"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd.train import Trainer, build_model, synthetic_batch  # noqa: E402
from grafp_amd.util import load_config  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
cfg = load_config()
cfg["bsz_train"] = pairs
torch.manual_seed(0)
model = build_model(cfg, device=dev)
tr = Trainer(cfg, model, dev, amp_dtype=torch.bfloat16)
x_i, x_j = synthetic_batch(pairs, 1, dev)
for _ in range(5):
    tr.step(x_i, x_j)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10):
    tr.step(x_i, x_j)
torch.cuda.synchronize()
print(f"eager step at {pairs} pairs: {(time.perf_counter() - t0) * 100:.2f} ms per step (10 steps)")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.step(x_i, x_j)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(40)
