"""tools/gemm_one.py R K M [views] [stats]: one forward 1x1-conv product, 10 launches -- for a counter pass over a single shape:
    PROG=tools/gemm_one.py tools/pmc.sh NAME FETCH_SIZE 1024 256 524288 2 1"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd import ops
R, K, M = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
views = int(sys.argv[4]) if len(sys.argv) > 4 else 2
stats = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
dev = "cuda:0"
w = (0.1 * torch.randn(R, K, device=dev)).to(torch.bfloat16)
x = torch.randn(K, M, device=dev).to(torch.bfloat16)
for _ in range(10):
    ops.conv1x1_gemm(w, x, 1, views if stats else 1, stats=stats)
torch.cuda.synchronize()
print("algorithmic MB per launch:", (R + K) * M * 2 / 1e6)
