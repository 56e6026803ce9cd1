"""The f32 mode with its matrix-bound products as split-bf16 MFMAs (ops.switches.f32_split_gemm, opt-in) against the library's
f32 GEMMs: step time at 256 / 1024 pairs and the free-running embedding distance between the two on a trained model."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from grafp_amd import ops
from grafp_amd.train import Trainer, build_model, synthetic_batch
from grafp_amd.util import load_config
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
cfg = load_config()
for B in (256, 1024):
    for split in (False, True):
        ops.switches.f32_split_gemm = split
        cfg["bsz_train"] = B
        torch.manual_seed(1)
        model = build_model(cfg, device=dev)
        tr = Trainer(cfg, model, dev, amp_dtype=None)
        x_i, x_j = synthetic_batch(B, 3, dev)
        for _ in range(2): tr.step(x_i, x_j)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 4
        for _ in range(n): loss = tr.step(x_i, x_j)
        torch.cuda.synchronize()
        print(f"f32 mode B={B} split={split}: {(time.perf_counter()-t0)/n*1e3:.1f} ms/step loss {float(loss):.5f}", flush=True)
# parity of embeddings: split vs library on the same trained model (eval mode), free-running
from _retrieval_case import build_case
ops.switches.f32_split_gemm = False
case = build_case(dev, n_tracks=8, seconds=20, train_steps=40, n_test=50)
m = case["model"]; segs = case["db"][:256]
with torch.no_grad():
    z0 = m.embed(segs)[1].float()
    ops.switches.f32_split_gemm = True
    z1 = m.embed(segs)[1].float()
ops.switches.f32_split_gemm = False
r = torch.linalg.norm(z1 - z0, dim=1) / torch.linalg.norm(z0, dim=1)
print(f"free-running embedding distance, split-bf16 f32 mode vs library f32 mode (trained model, 256 segments): max {float(r.max()):.2e} mean {float(r.mean()):.2e} median {float(r.median()):.2e}")
