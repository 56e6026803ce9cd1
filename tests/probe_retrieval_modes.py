"""Measurement script (not a test): retrieval hit rates of the SAME briefly trained model with fingerprints generated
three ways -- HIP bf16, HIP f32, CPU oracle f32 -- on one synthetic corpus (north_star's last criterion: top-1 hit rate
within 0.5 pt of the reference).  Decides the sizes / bars of tests/test_gpu_bf16.py::test_hit_rates_*.

    python tests/probe_retrieval_modes.py [n_tracks] [train_steps]
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]

from _retrieval_case import build_case, fingerprints_hip, fingerprints_oracle, hit_rates  # noqa: E402


def main():
    n_tracks = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dev = torch.device("cuda:0")
    t0 = time.time()
    case = build_case(dev, n_tracks=n_tracks, train_steps=steps, snrs=(0, 10), n_test=200)
    print(f"threads {torch.get_num_threads()}; case: {case['db'].shape[0]} db / {case['dummy'].shape[0]} dummy segments, built in {time.time() - t0:.1f} s")
    cache = {}

    def fps(mode, key):
        if (mode, key) not in cache:
            t1 = time.time()
            fp = fingerprints_oracle if mode == "oracle" else (lambda m, s: fingerprints_hip(m, s, dev, mode))
            cache[(mode, key)] = fp(case["model"], case[key])
            print(f"   fingerprints {mode}/{key}: {time.time() - t1:.1f} s", flush=True)
        return cache[(mode, key)]
    for snr in case["snrs"]:
        print(f"--- query SNR {snr} dB", flush=True)
        rows = {}
        for mode in ("bf16", "f32", "oracle"):
            t0 = time.time()
            d, u, q = (fps(mode, k) for k in ("db", "dummy", f"query{snr}"))
            rows[mode] = hit_rates(q, d, u, case["test_ids"], case["lens"])
            print(f"{mode:7s} top-1 exact by length {case['lens']}: {np.round(rows[mode][0], 2)}   top-10: "
                  f"{np.round(rows[mode][3], 2)}   ({time.time() - t0:.1f} s)")
        print("bf16 - oracle:", np.round(rows["bf16"][0] - rows["oracle"][0], 2), " f32 - oracle:",
              np.round(rows["f32"][0] - rows["oracle"][0], 2))


if __name__ == "__main__":
    main()
