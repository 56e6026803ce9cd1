"""tools/mrconv_bench.py [--clips B]: the max-relative forward (with the arg-max record) and its backward from the
record, per stage shape, on bf16 (C, B, N) rows -- us per launch and TB/s of the algorithmic bytes (forward: x read
once, 2 C rows written; backward: the 2 C gradient rows read, C written)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=2048)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.clips
    for s, (C, N) in enumerate(((64, 1024), (128, 512), (256, 256), (512, 128))):
        x = torch.randn(C, B, N, device=dev).to(torch.bfloat16).requires_grad_(True)
        idx = torch.randint(0, N, (B, N, 3), device=dev, dtype=torch.int32)
        g = torch.randn(2 * C, B, N, device=dev).to(torch.bfloat16)
        def fwd():
            return ops.max_relative(x, idx, "cbn")
        def fb():
            y = ops.max_relative(x, idx, "cbn")
            y.backward(g)
            x.grad = None
        for f in (fwd, fb):
            for _ in range(3): f()
        res = []
        for f in (fwd, fb):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): f()
            torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 20 * 1e6)
        f_us, b_us = res[0], res[1] - res[0]
        print(f"s{s} C={C:4d} N={N:5d}: forward {f_us:7.1f} us {3 * C * B * N * 2 / f_us / 1e6:5.2f} TB/s | "
              f"backward {b_us:7.1f} us {3 * C * B * N * 2 / b_us / 1e6:5.2f} TB/s")


if __name__ == "__main__":
    main()
