"""Measurement script (not a test): how far is the HIP bf16 mode from the oracle's bf16-storage restatement?

    python tests/probe_bf16.py            # on the GPU box

Prints, per stored activation (stem, every Grapher/FFN/Downsample output, every max-relative output), the relative L2
distance and the fraction of elements that differ between the HIP path (bf16 autocast) and `oracle.model` run with
q = round_bf16 on the same k-NN edges, free-running (each side consumes its OWN previous layer), then the embedding
distances.  The numbers decide which bar tests/test_gpu_bf16.py can assert.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]

from _common import RecordedGraphs, filled_state_dict, simclr_inputs  # noqa: E402


def main(B=4):
    from grafp_amd import ops
    from grafp_amd.train import build_model
    from grafp_amd.util import load_config
    from oracle import model as om
    dev = torch.device("cuda:0")
    cfg = load_config()
    cfg["bsz_train"] = B
    model = build_model(cfg)
    sd0 = model.state_dict()
    sd0.update(filled_state_dict())
    model.load_state_dict(sd0)
    model = model.to(dev).train()
    xi, xj = simclr_inputs()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    hip_acts = []
    orig_bn, orig_mr = ops.bn_act, ops.max_relative

    def bn(*a, **k):
        out = orig_bn(*a, **k)
        hip_acts.append(out.detach().float().cpu())
        return out

    def mr(*a, **k):
        out = orig_mr(*a, **k)
        hip_acts.append(out.detach().float().cpu())
        return out
    ops.bn_act, ops.max_relative = bn, mr
    try:
        with RecordedGraphs() as rg, torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            h_i, h_j, z_i, z_j = model(xi.to(dev), xj.to(dev))
    finally:
        ops.bn_act, ops.max_relative = orig_bn, orig_mr
    z_hip = torch.cat([z_i, z_j]).float().cpu()
    h_hip = torch.cat([h_i, h_j]).float().cpu()

    taps = []
    with torch.no_grad():
        o = om.simclr_forward({k: v.clone() for k, v in sd.items()}, xi, xj, True, idx_fn=rg.replay_fn(),
                              q=om.round_bf16, tap=lambda n, t: (taps.append((n, t.clone())), t)[1])
        f = om.simclr_forward({k: v.clone() for k, v in sd.items()}, xi, xj, True, idx_fn=rg.replay_fn())
    z_emu, h_emu = torch.cat([o[2], o[3]]), torch.cat([o[0], o[1]])
    z_f32 = torch.cat([f[2], f[3]])
    nl = len(taps) // 2
    assert len(hip_acts) == nl, (len(hip_acts), nl)
    print(f"{'layer':58s} {'rel-L2':>10s} {'frac!=':>9s} {'max ulp':>8s}")
    for li in range(nl):
        name = taps[li][0]
        emu = torch.cat([taps[li][1], taps[nl + li][1]], dim=0).squeeze(-1).permute(1, 0, 2)     # (C, 2B, N)
        got = hip_acts[li]
        d = (got - emu)
        rel = float(d.norm() / emu.norm())
        neq = float((d != 0).float().mean())
        ulp = (d.abs() / (emu.abs().clamp_min(1e-30) * 2.0 ** -8)).max()
        print(f"{name:58s} {rel:10.3e} {neq:9.2e} {float(ulp):8.2f}")

    def rl(a, b):
        return float((torch.linalg.norm(a - b, dim=1) / torch.linalg.norm(b, dim=1)).max())
    print("z: hip-bf16 vs emu-bf16 (max per-row rel-L2):", rl(z_hip, z_emu))
    print("h: hip-bf16 vs emu-bf16:", rl(h_hip, h_emu))
    print("z: emu-bf16 vs oracle-f32 (same edges):", rl(z_emu, z_f32))
    print("z: hip-bf16 vs oracle-f32 (same edges):", rl(z_hip, z_f32))
    print("near-tie flips avoided by replay:", rg.flips)


if __name__ == "__main__":
    main()
