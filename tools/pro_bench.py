"""Normalise-on-load again, with this round's tiles: consumer GEMM (+stats) with and without PRO, the skipped bn_affine pass,
and the consumer's weight gradient with and without PRO, at 2048 clip-views."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd import ops
dev = "cuda:0"
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
tot = [0.0, 0.0]
depth = (2, 2, 6, 2)
for stage, (C, N) in enumerate(((64, 1024), (128, 512), (256, 256), (512, 128))):
    M = clips * N
    for name, R, K, g in (("gfc2", C, 2 * C, 1), ("ffn2", C, 4 * C, 1)):
        x = torch.randn(K, M, device=dev).to(torch.bfloat16)
        w = (torch.randn(R, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        gy = torch.randn(R, M, device=dev).to(torch.bfloat16)
        tab = torch.rand(K, 2, 2, device=dev) + 0.5
        t_aff = timeit(lambda: ops.bn_affine(x, tab, 2, None, ops.ACT_RELU))
        t_g = timeit(lambda: ops.conv1x1_gemm(w, x, 1, 2, stats=True))
        t_gp = timeit(lambda: ops.conv1x1_gemm(w, x, 1, 2, pro_tab=tab, pro_act=ops.ACT_RELU, stats=True))
        t_w = timeit(lambda: ops._wgrad_bf16(gy, x, R, K, 1, M))
        t_wp = timeit(lambda: ops._wgrad_bf16(gy, x, R, K, 1, M, views=2, pro_tab=tab, pro_act=ops.ACT_RELU))
        now, pro = t_aff + t_g + t_w, t_gp + t_wp
        tot[0] += now * depth[stage]; tot[1] += pro * depth[stage]
        print(f"s{stage} {name} R={R:4d} K={K:4d}: affine {t_aff:6.1f} | gemm {t_g:6.1f} -> +pro {t_gp:6.1f} | wgrad {t_w:6.1f} -> +pro {t_wp:6.1f} | now {now:7.1f} pro {pro:7.1f} ({pro - now:+6.1f})", flush=True)
print(f"per step: now {tot[0]/1e3:.2f} ms, with PRO {tot[1]/1e3:.2f} ms")
