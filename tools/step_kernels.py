"""Every device kernel of ONE training step at a small per-GPU batch, in launch order: name, microseconds, and the idle gap
in front of it -- the fixed costs (tiny torch kernels, launch gaps) that decide the 128-pairs-per-GPU step of BASELINE
config 3.    python tools/step_kernels.py [pairs] [--graph]
Prints (1) totals: wall time of the step, sum of kernel time, number of kernels, sum of gaps; (2) the kernels that are NOT
hand-written (torch / library), by name; (3) the 40 largest gaps with the kernels on both sides."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd.train import Trainer, build_model, synthetic_batch  # noqa: E402
from grafp_amd.util import load_config  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 128
graph = "--graph" in sys.argv
dev = torch.device("cuda:0")
cfg = load_config()
cfg["bsz_train"] = pairs
torch.manual_seed(0)
model = build_model(cfg, device=dev)
tr = Trainer(cfg, model, dev, amp_dtype=torch.bfloat16)
x_i, x_j = synthetic_batch(pairs, 1, dev)
step = tr.step_graph if graph else tr.step
for _ in range(5):
    step(x_i, x_j)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(x_i, x_j)
    torch.cuda.synchronize()
ks = []
for e in prof.events():
    if str(e.device_type).endswith("CUDA") and e.time_range is not None:
        dur = e.device_time if hasattr(e, "device_time") else e.cuda_time
        ks.append((e.time_range.start, e.time_range.end, dur, e.name))
ks.sort()
t0, t1 = ks[0][0], max(k[1] for k in ks)
busy = sum(k[2] for k in ks)
gaps = []
end = ks[0][1]
for i in range(1, len(ks)):
    g = ks[i][0] - end
    if g > 0:
        gaps.append((g, ks[i - 1][3][:70], ks[i][3][:70]))
    end = max(end, ks[i][1])
print(f"{'graph replay' if graph else 'eager'} step at {pairs} pairs: {len(ks)} kernels, first start to last end {(t1 - t0):.0f} us, "
      f"kernel time {busy:.0f} us, idle between kernels {sum(g[0] for g in gaps):.0f} us in {len(gaps)} gaps")
other = collections.defaultdict(lambda: [0, 0.0])
mine = collections.defaultdict(lambda: [0, 0.0])
for _, _, dur, name in ks:
    d = mine if "grafp::" in name else other
    d[name[:100]][0] += 1
    d[name[:100]][1] += dur
print(f"--- hand-written kernels: {sum(v[0] for v in mine.values())} launches, {sum(v[1] for v in mine.values()):.0f} us")
for name, (n, us) in sorted(mine.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{n:5d} {us:9.1f} us  {name}")
print(f"--- torch / library kernels: {sum(v[0] for v in other.values())} launches, {sum(v[1] for v in other.values()):.0f} us")
for name, (n, us) in sorted(other.items(), key=lambda kv: -kv[1][0]):
    print(f"{n:5d} {us:9.1f} us  {name}")
print("--- largest gaps (us | before | after)")
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print(f"{g:8.1f} | {a} | {b}")
hist = collections.Counter(min(int(g[0]), 20) for g in gaps)
print("--- gap histogram (us: count):", dict(sorted(hist.items())))
