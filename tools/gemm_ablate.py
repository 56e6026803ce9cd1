"""Where does the time of the hand-written GEMM go?  Times the stage 1-3 FFN products with the library named by
GRAFP_HIP_LIB (tools/gemm_ablate.sh builds the variants: GM_ABLATE = 1 no DMA in the main loop, 2 no epilogue,
4 no MFMA, sums of those); one process per variant:

    for n in 0 1 2 3 4 5 6; do GRAFP_HIP_LIB=$([ $n = 0 ] || echo tools/_ablate/libgrafp_hip_$n.so) python tools/gemm_ablate.py 2048 $n; done
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd import ops  # noqa: E402


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    dev = "cuda:0"
    clips = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    label = sys.argv[2] if len(sys.argv) > 2 else "0"
    names = {"0": "full", "1": "no DMA", "2": "no epilogue", "3": "MFMA + LDS reads only", "4": "no MFMA",
             "5": "epilogue only", "6": "DMA only"}
    out = []
    for name, R, K, N in (("s1 ffn1", 512, 128, 512), ("s2 ffn1", 1024, 256, 256), ("s2 ffn2", 256, 1024, 256),
                          ("s3 ffn1", 2048, 512, 128), ("s3 ffn2", 512, 2048, 128)):
        M = clips * N
        w = (0.1 * torch.randn(R, K, device=dev)).to(torch.bfloat16)
        x = torch.randn(K, M, device=dev).to(torch.bfloat16)
        t = timeit(lambda: ops.conv1x1_gemm(w, x, 1, 2, stats=True))
        out.append(f"{name} {t:7.1f} us ({2.0 * R * K * M / t / 1e6:5.0f} TF)")
    print(f"{names.get(label, label):22s} | " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
