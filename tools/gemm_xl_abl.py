"""Ablations of the four-wave tile (measurement library; results are garbage, only the time means something):
GRAFP_XL_ABL bits: 1 no W pieces, 2 no stores, 4 no X pieces, 8 no MFMAs
(10 = DMA only, 11 = X pieces only, 14 = W pieces only, 13 = stores only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd import ops
from tools.gemm_bench import timeit
dev = "cuda:0"
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
os.environ["GRAFP_GEMM_CFG"] = "5"
for name, R, K, M in (("s2 ffn1", 1024, 256, clips * 256), ("s2 ffn2", 256, 1024, clips * 256), ("s3 ffn1", 2048, 512, clips * 128),
                      ("s3 ffn2", 512, 2048, clips * 128), ("s2 fc1", 256, 256, clips * 256)):
    w = (0.1 * torch.randn(R, K, device=dev)).to(torch.bfloat16)
    x = torch.randn(K, M, device=dev).to(torch.bfloat16)
    row = []
    for abl in (0, 2, 7, 8, 10, 11, 14, 13):
        os.environ["GRAFP_XL_ABL"] = str(abl)
        row.append(f"abl{abl} {timeit(lambda: ops.conv1x1_gemm(w, x, 1, 2, stats=True)):7.1f}")
    os.environ["GRAFP_XL_ABL"] = "0"
    fl = 2.0 * R * K * M
    print(f"{name} R={R} K={K} M={M}: " + " | ".join(row) + f" us | mfma floor {fl / 2.5e15 * 1e6:.1f} us", flush=True)
