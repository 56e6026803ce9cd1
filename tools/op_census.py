"""Which torch ops does one training step launch?  (fixed per-step costs that matter at small per-GPU batches)"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd.train import Trainer, build_model, synthetic_batch
from grafp_amd.util import load_config
dev = torch.device("cuda:0")
cfg = load_config(); cfg["bsz_train"] = 128
model = build_model(cfg, device=dev)
tr = Trainer(cfg, model, dev, amp_dtype=torch.bfloat16)
x_i, x_j = synthetic_batch(128, 1, dev)
for _ in range(3):
    tr.step(x_i, x_j)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(x_i, x_j)
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::copy_", "aten::add", "aten::add_", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::clone")]
cnt = collections.Counter()
for e in ev:
    st = [s for s in (e.stack or []) if "grafp_amd" in s or "bench" in s]
    cnt[(e.name, st[0] if st else "(torch internal)")] += 1
for (name, where), n in cnt.most_common(40):
    print(f"{n:5d} {name:16s} {where}")

# device kernels of the same step: name -> launches, total microseconds
kc = collections.Counter()
kt = collections.Counter()
for e in prof.events():
    if str(e.device_type).endswith("CUDA"):
        kc[e.name[:90]] += 1
        kt[e.name[:90]] += e.device_time if hasattr(e, "device_time") else e.cuda_time
print("--- device kernels of one step")
for name, n in kc.most_common(25):
    print(f"{n:5d} {kt[name]:9.1f} us  {name}")
# who launches the fills?
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::new_zeros", "aten::full"):
        st = [s for s in (e.stack or [])][:6]
        print(e.name, "|", " <- ".join(s.split("/")[-1] for s in st[:5]))
