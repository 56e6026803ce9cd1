"""Five eager training steps at B pairs (default 1024), nothing else: the target of tools/pmc_view.sh and of ad-hoc rocprofv3 passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd.train import Trainer, build_model, synthetic_batch
from grafp_amd.util import load_config
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
device = torch.device("cuda", 0); torch.cuda.set_device(0)
cfg = load_config(); cfg["bsz_train"] = B
torch.manual_seed(1234)
model = build_model(cfg, device=device)
tr = Trainer(cfg, model, device, amp_dtype=torch.bfloat16)
x_i, x_j = synthetic_batch(B, 7, device)
for _ in range(5): tr.step(x_i, x_j)
torch.cuda.synchronize()
print("ok")
