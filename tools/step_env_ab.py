"""A/B of a measurement-build plan knob (csrc/tuning.h) on the eager training step, one process, alternating:
    GRAFP_HIP_LIB=$PWD/grafp_amd/libgrafp_hip_measure.so python tools/step_env_ab.py PAIRS NAME=v0,v1,...
(the knob is an environment variable the measurement build reads at every launch)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd.train import Trainer, build_model, synthetic_batch
from grafp_amd.util import load_config
B = int(sys.argv[1]); name, vals = sys.argv[2].split("="); vals = vals.split(",")
device = torch.device("cuda", 0); torch.cuda.set_device(0)
cfg = load_config(); cfg["bsz_train"] = B
x_i, x_j = synthetic_batch(B, 7, device)
torch.manual_seed(1234)
model = build_model(cfg, device=device)
tr = Trainer(cfg, model, device, amp_dtype=torch.bfloat16)
for _ in range(3): tr.step(x_i, x_j)
for rep in range(3):
    for v in vals:
        os.environ[name] = v
        tr.step(x_i, x_j)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 10
        for _ in range(n): l = tr.step(x_i, x_j)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
        print(f"pairs={B} {name}={v}: {dt:8.3f} ms/step  loss {float(l):.5f}", flush=True)
