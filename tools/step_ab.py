"""A/B of an ops switch on the eager training step: python tools/step_ab.py B name=v0,v1,... [graph]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd.train import Trainer, build_model, synthetic_batch
from grafp_amd.util import load_config
from grafp_amd import ops
B = int(sys.argv[1]); name, vals = sys.argv[2].split("="); vals = [int(v) for v in vals.split(",")]
graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
device = torch.device("cuda", 0); torch.cuda.set_device(0)
cfg = load_config(); cfg["bsz_train"] = B
x_i, x_j = synthetic_batch(B, 7, device)
for rep in range(2):
    for v in vals:
        setattr(ops.switches, name, v)
        torch.manual_seed(1234)
        model = build_model(cfg, device=device)
        tr = Trainer(cfg, model, device, amp_dtype=torch.bfloat16)
        step = tr.step_graph if graph else tr.step
        losses = []
        for _ in range(4): losses.append(float(step(x_i, x_j)))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 10
        for _ in range(n): l = step(x_i, x_j)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
        print(f"B={B} {name}={v} {'graph' if graph else 'eager'}: {dt:8.3f} ms/step  loss {losses[-1]:.5f} -> {float(l):.5f}  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
        del tr, model
        torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
