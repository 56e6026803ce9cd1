"""A/B of a measurement-build plan knob (csrc/tuning.h) on the GRAPH-REPLAYED training step, one process, alternating:
    GRAFP_HIP_LIB=$PWD/grafp_amd/libgrafp_hip_measure.so python tools/step_env_graph_ab.py PAIRS[,PAIRS] NAME=v0,v1,... [reps]
The knob is read when a launch is ENQUEUED, i.e. while the step is captured: every value gets its own Trainer and its own
capture of the same model; replays are timed (best of three runs of 20).  Graph replay is lease-independent (no host in
the loop), which is what a 1 % effect at 128 pairs needs."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd.train import Trainer, build_model, synthetic_batch  # noqa: E402
from grafp_amd.util import load_config  # noqa: E402

pairs_list = [int(v) for v in sys.argv[1].split(",")]
name, vals = sys.argv[2].split("=")
vals = vals.split(",")
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
for B in pairs_list:
    cfg = load_config()
    cfg["bsz_train"] = B
    x_i, x_j = synthetic_batch(B, 7, device)
    torch.manual_seed(1234)
    model = build_model(cfg, device=device)
    acc = {v: [] for v in vals}
    for rep in range(reps):
        for v in vals:
            os.environ[name] = v
            tr = Trainer(cfg, model, device, amp_dtype=torch.bfloat16)
            for _ in range(3):
                loss = tr.step_graph(x_i, x_j)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(20):
                    loss = tr.step_graph(x_i, x_j)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
            acc[v].append(best)
            print(f"pairs={B:5d} {name}={v:>6s}: graph {best:8.3f} ms/step  loss {float(loss):.5f}", flush=True)
            del tr
            torch.cuda.empty_cache()
    print(f"pairs={B:5d} summary: " + "  ".join(f"{name}={v}: {sum(a) / len(a):.3f}" for v, a in acc.items()), flush=True)
    del model
    torch.cuda.empty_cache()
