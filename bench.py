#!/usr/bin/env python3
"""bench.py -- GraFPrint contrastive training step on N MI355X GPUs of one node (one process per GPU).

    python bench.py [--gpus 1] [--steps 20] [--warmup 5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = zero-grad -> log-mel of both views -> SimCLR forward (peak extractor, 12 Grapher+FFN blocks with a
k-NN graph rebuilt in each, projector) for both views -> NT-Xent over the GLOBAL batch -> backward -> gradient
all-reduce -> Adam.  Inputs are synthetic 1 s clips already resident in HBM; weights
are randomly initialised (the architecture and shapes are config/grafp.yaml's).  Rank 0 prints ONE JSON line.

The headline workload is BASELINE.json's metric: the contrastive step at GLOBAL batch 1024 (1024 / N pairs per GPU,
global negatives; N = 8 is BASELINE config 3 and the batch fits one GPU, so N = 1 runs the same global batch:
`scaling: strong`).  Beside it the line carries BASELINE config 2 (256 pairs on one GPU: `config2_batch256`, also as
one HIP graph and in f32), the weak-scaling variant at N > 1 (256 pairs per GPU), the per-kernel table (HIP events,
collected in a SEPARATE pass so the headline loop runs without event records), `roofline` for the kernel family with
the largest measured time, a step-level bytes / time figure, and the retrieval / fingerprinting / augmentation legs.
"""
import argparse
import contextlib
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

if "--local-device" in sys.argv and int(os.environ.get("WORLD_SIZE", "1")) > 1 and "RANK" in os.environ:
    # test hook only: the ranks share ONE device -> disjoint CU sets, set before the HIP runtime exists in this process
    # (grafp_amd.dist.shared_device_cu_mask says why; restated here because that module imports torch)
    _per = 256 // int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("ROC_GLOBAL_CU_MASK", hex(((1 << _per) - 1) << (_per * int(os.environ["RANK"]))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, = f32 vector peak
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s achievable)


def host_threads():
    """ONE policy for both CPU baselines of a bench line (the oracle's train step and the exact CPU search): every host
    core up to 32 -- beyond 32 threads both legs get SLOWER on the GPU boxes' 128-core hosts (B = 32 step: 5 x slower at
    128 threads; the search is memory-bound) -- and `cores` reports exactly the number of threads that ran."""
    return max(1, min(32, os.cpu_count() or 1))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--global-batch", type=int, default=1024,
                    help="positive pairs per step over all GPUs (BASELINE metric: batch 1024; N = 8 is config 3)")
    ap.add_argument("--batch-per-gpu", type=int, default=None,
                    help="override: pairs per GPU and step (then global batch = this x N)")
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16", help="GEMM compute dtype (autocast)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-retrieval", action="store_true")
    ap.add_argument("--no-f32-probe", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="skip the HIP-graph replay of the step")
    ap.add_argument("--no-augment", action="store_true", help="skip the device-side augmentation leg")
    ap.add_argument("--no-config2", action="store_true", help="skip the 256-pair legs (eager, HIP graph, f32)")
    ap.add_argument("--kernel-steps", type=int, default=3, help="steps of the separate per-kernel timing pass")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=25.0)
    ap.add_argument("--backend", default=None, help="test hook: process-group backend (default nccl = RCCL)")
    ap.add_argument("--local-device", type=int, default=None, help="test hook: device index of every rank (ranks share a GPU)")
    return ap.parse_args()


def kernel_sources_sha16():
    """Fingerprint of the HIP sources (csrc/*.hip, *.h): tools/pmc_summary.py stamps it into profiles/pmc_*.json, and a
    counter summary whose stamp differs from the sources of this run is NOT reported as `roofline.traffic`."""
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "grafp_amd", "csrc", "*.hip")) +
                       glob.glob(os.path.join(ROOT, "grafp_amd", "csrc", "*.h"))):
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def knn_flops(meta):
    B, C, N, _k = meta
    return 2.0 * N * N * C * B


def mrconv_bytes(meta, esize):
    B, C, N, K = meta
    return (esize * C * N + 8.0 * K * N + 2.0 * esize * C * N) * B   # x + idx + interleaved (2C rows); bwd same order


PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA


def summarise_kernels(timed, esize=4):
    """name -> dict(calls, total_ms, avg_us, achieved, unit, peak, frac, bound, bytes) from HIP-event records.
    `bytes` = the ALGORITHMIC bytes of the timed launches (operands read once, results written once)."""
    from grafp_amd import ops
    out = {}
    # the batched reduction of the weight gradients' partial sums (one launch per backward pass) belongs to that family
    wgrad_reduce_ms = sum(ops.elapsed_ms(timed.get("conv1x1_wgrad_reduce") or []))
    for name, ev in timed.items():
        if not ev or name == "conv1x1_wgrad_reduce":
            continue
        ms = ops.elapsed_ms(ev)
        tot = sum(ms) + (wgrad_reduce_ms if name == "conv1x1_wgrad" else 0.0)
        row = {"calls": len(ev), "total_ms": round(tot, 4), "avg_us": round(1e3 * tot / len(ev), 2)}
        by = None
        if name == "knn_topk":
            fl = sum(knn_flops(m) for _, _, m in ev)
            by = sum((4.0 * C * N + 4.0 * N + 4.0 * k * N) * B for _, _, (B, C, N, k) in ev)
            row.update(bound="mfma", achieved=round(fl / (tot * 1e-3) / 1e12, 3), unit="TFLOP/s",
                       peak=PEAK_F32_MATRIX_TFLOPS)
        elif name == "knn_split":
            # the certified path (knn_split.hip), all passes of a call: norms, scan, exact recomputation of the
            # uncertified queries.  bf16 inputs: ONE bf16 MFMA per 16 channels of a 32 x 32 tile (2 N^2 C flops) beside
            # 12 VALU instructions per (query, candidate) pair for the key and its sorted insert -- the scan is VALU-bound,
            # so the matrix fraction is small by construction; valu_frac prices the pairs at 12 lane-operations against
            # 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz
            fl = sum(knn_flops(m[:4]) * (1.0 if m[4] == 2 else 3.0) for _, _, m in ev)
            by = sum((e * C * N * 2.0 + 4.0 * k * N) * B for _, _, (B, C, N, k, e) in ev)
            pairs = sum(float(B) * N * N for _, _, (B, C, N, k, e) in ev)
            row.update(bound="mfma", achieved=round(fl / (tot * 1e-3) / 1e12, 3), unit="TFLOP/s",
                       peak=PEAK_BF16_MFMA_TFLOPS,
                       valu_frac=round(pairs * 12.0 / (256 * 4 * 16 * 2.4e9) / (tot * 1e-3), 4),
                       note="VALU-bound scan (integer-key top-k inserts); see valu_frac")
        elif name == "knn_normalize":
            by = sum((esize * C * N + 4.0 * C * N + 4.0 * N) * B for _, _, (B, C, N, k) in ev)
        elif name in ("mrconv_fwd", "mrconv_bwd"):
            by = sum(mrconv_bytes(m, esize) for _, _, m in ev)
        elif name in ("bn_fwd", "bn_bwd"):
            # algorithmic passes: fwd reads x, writes z (+ residual read, not counted); bwd reads x and dz, writes dx
            per = 2.0 if name == "bn_fwd" else 3.0
            by = sum(per * C * M * e for _, _, (C, M, e) in ev)
        elif name == "bn_affine":
            # read y, write z, and read the shortcut where the layer closes a residual block
            by = sum((3.0 if m[2] else 2.0) * m[0] * m[1] * 2 for _, _, m in ev)
        elif name == "conv1x1_wgrad":
            by = sum((co + ci) * M * 2.0 for _, _, (co, ci, g, M) in ev)
            row["tflops"] = round(sum(2.0 * co * (ci // g) * M for _, _, (co, ci, g, M) in ev) / (tot * 1e-3) / 1e12, 1)
            ideal = sum(max((co + ci) * M * 2.0 / (PEAK_HBM_GBS * 1e9), 2.0 * co * (ci // g) * M / (PEAK_BF16_MFMA_TFLOPS * 1e12))
                        for _, _, (co, ci, g, M) in ev)
            row["two_ceiling_frac"] = round(ideal / (tot * 1e-3), 4)
        elif name == "conv1x1_gemm":
            # forward and data-gradient products of every 1x1 convolution: read W (small) and X (K rows), write Y (R rows)
            # (the concatenated-operand data gradients carry a fifth entry: operand rows that meet the identity block of
            # [W^T | I] -- they are bytes, not useful flops)
            shapes = [(m[0], m[1], m[2], m[3], m[4] if len(m) > 4 else 0) for _, _, m in ev]
            by = sum((R + K) * M * 2.0 for R, K, g, M, ident in shapes)
            fl = sum(2.0 * R * ((K - ident) // g) * M for R, K, g, M, ident in shapes)
            row["tflops"] = round(fl / (tot * 1e-3) / 1e12, 1)
            row["mfma_frac"] = round(fl / (tot * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)
            # the family mixes HBM-bound and matrix-bound shapes: per launch the roofline time is the LARGER of
            # bytes / HBM peak and flops / matrix peak; their sum over the measured time is the two-ceiling fraction
            ideal = sum(max((R + K) * M * 2.0 / (PEAK_HBM_GBS * 1e9),
                            2.0 * R * ((K - ident) // g) * M / (PEAK_BF16_MFMA_TFLOPS * 1e12))
                        for R, K, g, M, ident in shapes)
            row["two_ceiling_frac"] = round(ideal / (tot * 1e-3), 4)
        elif name == "ntxent":
            # S = Z Z^T tiles by exact-f32 MFMA, forward (log-sum-exp) + backward (dZ += W^T Z, two products): 3 x
            # 2 (2 n_local)(2 B_all) D flops; 2 x 2 B D x 4 bytes in, the same out.  Latency-bound (one launch of a few
            # hundred microseconds per step): the fraction says so
            fl = sum(3.0 * 2.0 * (2 * nl) * (2 * Ba) * D for _, _, (Ba, nl, D) in ev)
            by = sum(2.0 * (2 * Ba) * D * 4.0 + 2.0 * (2 * nl) * D * 4.0 for _, _, (Ba, nl, D) in ev)
            row.update(bound="mfma", achieved=round(fl / (tot * 1e-3) / 1e12, 3), unit="TFLOP/s",
                       peak=PEAK_F32_MATRIX_TFLOPS, tflops=round(fl / (tot * 1e-3) / 1e12, 3),
                       note="exact-f32 MFMA; launch/latency-bound at these sizes")
        elif name == "logmel":
            # SURVEY 8d prices K1 at 72 192 B and 1.64 M (FFT: 5 n log2 n per 1024-point frame) + 2.10 M (mel: 513 x 64
            # multiply-adds per frame) flops per clip-view: at 8 TB/s and 157 TFLOP/s the ARITHMETIC is the higher ceiling
            # (24 ns against 9 ns per clip-view), so the fraction of the HBM peak alone undersells the kernel:
            # two_ceiling_frac = max(bytes / HBM peak, flops / f32 peak) / measured
            by = sum(B * (4.0 * T + 4.0 * 64 * (1 + T // 512)) for _, _, (B, T) in ev)
            fl = sum(B * (1 + T // 512) * (5.0 * 1024 * 10 + 2.0 * 513 * 64) for _, _, (B, T) in ev)
            row["fp32_tflops"] = round(fl / (tot * 1e-3) / 1e12, 2)
            row["two_ceiling_frac"] = round(max(by / (PEAK_HBM_GBS * 1e9), fl / (PEAK_F32_MATRIX_TFLOPS * 1e12)) / (tot * 1e-3), 4)
        elif name in ("peak_extract_fwd", "peak_extract_bwd"):
            # 3 x 7 x 7 taps x 8 filters x 1024 positions multiply-adds per clip-view either way (2.41 MFLOP); forward moves
            # 8 192 + 32 768 B, backward reads the clip, the output and its gradient (73 728 B) and writes 1 184 sums
            per = 8192.0 + 32768.0 if name == "peak_extract_fwd" else 8192.0 + 2 * 32768.0
            by = sum(B * per for _, _, (B,) in ev)
            fl = sum(B * 2.0 * 147 * 8 * 1024 for _, _, (B,) in ev)
            row["fp32_tflops"] = round(fl / (tot * 1e-3) / 1e12, 2)
            row["two_ceiling_frac"] = round(max(by / (PEAK_HBM_GBS * 1e9), fl / (PEAK_F32_MATRIX_TFLOPS * 1e12)) / (tot * 1e-3), 4)
        if by is not None:
            row["bytes"] = by
            if "achieved" not in row:
                row.update(bound="hbm", achieved=round(by / (tot * 1e-3) / 1e9, 1), unit="GB/s", peak=PEAK_HBM_GBS)
        if "achieved" in row:
            row["frac"] = round(row["achieved"] / row["peak"], 4)
        out[name] = row
    return out


KERNEL_NAMES = ("conv1x1_gemm", "conv1x1_wgrad", "conv1x1_wgrad_reduce", "bn_bwd", "bn_affine", "bn_fwd", "knn_split", "knn_topk", "knn_normalize",
                "mrconv_fwd", "mrconv_bwd", "ntxent", "logmel", "peak_extract_fwd", "peak_extract_bwd")


def cpu_baseline(cfg, seconds):
    """The oracle's CPU train step (test infrastructure used ONLY as the reported baseline): B = 32 pairs,
    log-mel -> peak extractor -> GraphEncoder -> projector -> NT-Xent, fwd + bwd + Adam, f32, on `host_threads()`
    threads.  1 warm-up step, then timed steps until >= 2 steps or `seconds` have elapsed."""
    from grafp_amd.train import build_model
    from oracle import model as om
    B = 32
    torch.set_num_threads(host_threads())
    torch.manual_seed(0)
    sd = {k: v.clone() for k, v in build_model(dict(cfg, bsz_train=B)).state_dict().items()}
    for k, v in sd.items():
        if v.is_floating_point() and k.rsplit(".", 1)[-1] not in ("running_mean", "running_var", "relative_pos"):
            v.requires_grad_(True)
    params = list(om.trainable(sd).values())
    opt = torch.optim.Adam(params, lr=cfg["lr"])
    gen = torch.Generator().manual_seed(0)
    x_i = 0.1 * torch.randn(B, 16000, generator=gen)
    x_j = x_i + 0.03 * torch.randn(B, 16000, generator=torch.Generator().manual_seed(1))

    def step():
        with torch.no_grad():
            S_i, S_j = om.logmel(x_i, cfg), om.logmel(x_j, cfg)
        return om.train_step(sd, opt, S_i, S_j, cfg["tau"])
    step()
    n, t0 = 0, time.perf_counter()
    while n < 2 or (time.perf_counter() - t0 < seconds and n < 50):
        step()
        n += 1
        if time.perf_counter() - t0 > 2 * seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 3), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle CPU step, B=32 pairs (64 clip-views), fwd+bwd+Adam, f32, 1 warm-up + {n} timed "
                      f"steps in {dt:.1f} s"}


# per-dimension noise of the planted queries: chosen so that the exact top-1 hit rate on 1M random unit vectors lands
# between 80 % and 95 % -- informative, unlike a sigma at which every query trivially hits
QUERY_SIGMA = 0.15


def retrieval_probe(device, cpu_check=True):
    """Secondary metric of BASELINE.json: exact top-20 search QPS on a 1 000 000 x 128 resident database
    (L2-normalised randn, seed 2), planted noisy queries; batch sizes 1, 41, 4096."""
    from grafp_amd import ops
    gen = torch.Generator(device=device).manual_seed(2)
    n = 1_000_000
    db = torch.nn.functional.normalize(torch.randn(n, 128, generator=gen, device=device), dim=1)
    rows = torch.randint(0, n, (4096,), generator=gen, device=device)
    q = torch.nn.functional.normalize(db[rows] + QUERY_SIGMA * torch.randn(4096, 128, generator=gen, device=device), dim=1)
    sq = ops.row_sqnorm(db)
    dbh = ops.rows_to_bf16(db)        # the index's pre-filter copy (same results, see knn_search.hip)
    res = {"db": "1000000x128 f32 resident (+ bf16 pre-filter copy)", "k": 20, "query_sigma": QUERY_SIGMA}
    small = {}
    for nq, reps, warm in ((1, 300, 30), (41, 300, 30), (4096, 20, 3)):
        # small batches: a call is tens of microseconds, so 20 calls after 3 warm-ups measure the clock ramp and the
        # allocator, not the kernels (round 4: the driver saw 0.097 ms where a longer probe saw 0.075).  >= 30 untimed
        # calls, then 300 timed ones: the reported latency is wall time over the whole run (back-to-back calls, what a
        # serving loop sees); min and median of per-call HIP-event times are printed beside it.
        for _ in range(warm):
            ops.search_l2(db, sq, q[:nq], 20, db_bf16=dbh)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            D_nq, I = ops.search_l2(db, sq, q[:nq], 20, db_bf16=dbh)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        if nq <= 41:
            small[nq] = (D_nq.clone(), I.clone())
            evs = []
            for _ in range(100):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                ops.search_l2(db, sq, q[:nq], 20, db_bf16=dbh)
                b.record()
                evs.append((a, b))
            torch.cuda.synchronize()
            per = sorted(a.elapsed_time(b) for a, b in evs)
            res[f"ms_per_batch_nq{nq}_events"] = {"min": round(per[0], 4), "median": round(per[len(per) // 2], 4),
                                                   "reps": len(per)}
        res[f"qps_nq{nq}"] = round(nq / dt, 1)
        res[f"ms_per_batch_nq{nq}"] = round(dt * 1e3, 4)
        if nq == 4096:
            res["top1_hit_rate"] = round(float((I[:, 0] == rows).float().mean().item()), 4)
            # algorithmic rate 2*128*n*nq / t: above the 157 TFLOP/s exact-f32 matrix peak because the scan runs in
            # bf16 and only the surviving rows are rescored in f32
            res["tflops_nq4096"] = round(2.0 * 128 * n * nq / dt / 1e12, 2)
        if nq == 1:
            res["db_stream_GBps_nq1"] = round(n * 260.0 / dt / 1e9, 1)      # bf16 rows + norms
        # roofline of one batched search: the pre-filter scan streams the bf16 rows + f32 norms once
        # (n * 260 B) + queries and results; its 2*128*n*nq flops run on the bf16 matrix cores.  Small batches are
        # bound by the stream (and by the launch chain around it), large ones by the matrix/epilogue side.
        by = n * 260.0 + nq * 512.0 + nq * 20 * 12.0
        fl = 2.0 * 128 * n * nq
        t_hbm, t_mfma = by / (PEAK_HBM_GBS * 1e9), fl / (PEAK_BF16_MFMA_TFLOPS * 1e12)
        if t_hbm >= t_mfma:
            roof = {"bound": "hbm", "achieved": round(by / dt / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s"}
        else:
            roof = {"bound": "mfma", "achieved": round(fl / dt / 1e12, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s"}
        roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
        roof["floor_us"] = round(max(t_hbm, t_mfma) * 1e6, 1)
        res.setdefault("roofline", {})[f"nq{nq}"] = roof
    # BASELINE config 4 end to end: 2000 test ids x 41-segment runs = 82 000 query segments, one batched search, then
    # ONE rerank launch over the 8000 (test id, length) items for lengths 1/11/21/41 (eval.py:262-301)
    n_ids, lens = 2000, (1, 11, 21, 41)
    starts = torch.randint(0, n - 41, (n_ids,), generator=gen, device=device)
    seg = (starts[:, None] + torch.arange(41, device=device)[None, :]).reshape(-1)
    qs = torch.nn.functional.normalize(db[seg] + QUERY_SIGMA * torch.randn(seg.numel(), 128, generator=gen, device=device),
                                       dim=1)
    item_row = (torch.arange(n_ids, device=device) * 41).repeat_interleave(len(lens))
    item_len = torch.tensor(lens, dtype=torch.int32, device=device).repeat(n_ids)
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, I_seg = ops.search_l2(db, sq, qs, 20, db_bf16=dbh)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pred, _ = ops.seq_rerank(db, qs, I_seg, item_row, item_len, top=10, max_len=max(lens))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    hit = (pred[:, 0].reshape(n_ids, len(lens)) == starts[:, None]).float().mean(0)
    res["eval_pipeline"] = {"segments": int(seg.numel()), "items": int(item_row.numel()), "lengths": list(lens),
                            "search_ms": round((t1 - t0) * 1e3, 2), "rerank_ms": round((t2 - t1) * 1e3, 3),
                            "segment_qps": round(seg.numel() / (t1 - t0), 1),
                            "items_per_s": round(item_row.numel() / (t2 - t1), 1),
                            "top1_hit_rate_by_length": [round(float(h), 4) for h in hit]}
    if cpu_check:
        # CPU exact search beside it (oracle/csrc/flat_search.c, OpenMP over database rows): ALL 4096 queries once --
        # ids and distances must be equal bit for bit -- and the nq = 41 batch repeatedly for the reported rate
        from oracle import native
        # the same thread policy as the train-step baseline; torch has loaded libgomp long before this point, so the
        # environment variable would come too late: set the runtime's thread count directly, and report what it says
        import ctypes
        threads = host_threads()
        try:
            gomp = ctypes.CDLL("libgomp.so.1")
            gomp.omp_set_num_threads(threads)
            threads = int(gomp.omp_get_max_threads())
        except OSError:
            os.environ["OMP_NUM_THREADS"] = str(threads)
        D, I = ops.search_l2(db, sq, q, 20, db_bf16=dbh)
        db_h, q_h = db.cpu().numpy(), q.cpu().numpy()
        native.flat_search_l2(db_h[:1000], q_h[:1], 20)                 # builds / loads the library outside the timing
        t0 = time.perf_counter()
        wd, wi = native.flat_search_l2(db_h, q_h, 20)
        dt_all = time.perf_counter() - t0
        t0, reps = time.perf_counter(), 0
        while reps < 3 or (time.perf_counter() - t0 < 4.0 and reps < 50):
            native.flat_search_l2(db_h, q_h[:41], 20)
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        res["cpu_baseline"] = {"value": round(41 / dt, 2), "unit": "queries/s", "cores": threads, "kind": "port",
                               "sample": f"the nq=41 batch against the full 1M x 128 database, exact search "
                                         f"(oracle/csrc/flat_search.c, OpenMP over database rows), {reps} passes of "
                                         f"{dt:.2f} s; all 4096 queries once in {dt_all:.1f} s"}
        res["ids_equal_cpu_exact"] = bool((I.cpu().numpy() == wi).all())
        res["dist_equal_cpu_exact"] = bool((D.cpu().numpy() == wd).all())
        # the SMALL-batch launches take another path through the library (one or two query sets per wave, histogram
        # select): their results -- the ones timed above -- against the same CPU rows
        for nq_s, (D_s, I_s) in small.items():
            res[f"ids_equal_cpu_exact_nq{nq_s}"] = bool((I_s.cpu().numpy() == wi[:nq_s]).all())
            res[f"dist_equal_cpu_exact_nq{nq_s}"] = bool((D_s.cpu().numpy() == wd[:nq_s]).all())
        res["cpu_checked_queries"] = int(q_h.shape[0])
        res["top1_hit_rate_cpu_exact"] = round(float((wi[:, 0] == rows.cpu().numpy()).mean()), 4)
    return res


def timed_steps(fn, steps, barrier):
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    barrier()
    return time.perf_counter() - t0, out


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves.

    Runs BEFORE anything touches the GPU (`torch.cuda.device_count()` does not initialise it on this image) and starts
    `python -m torch.distributed.run ... bench.py <same arguments>` as a CHILD process -- never an exec -- whose
    stdout (rank 0's JSON line) and exit code are forwarded.  Fewer visible devices than ranks is an error, not a
    one-rank run that prints `n_gpus: 1` (the --local-device test hook, where the ranks share a device, is exempt)."""
    import socket
    import subprocess
    if args.local_device is None:
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible; refusing to run fewer ranks "
                             "than asked for\n")
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    # stdout carries the ONE JSON line and nothing else: everything a library prints on file descriptor 1 while the
    # bench runs (RCCL's version banner at the first communicator, progress lines of the retrieval legs) goes to stderr
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
    from grafp_amd import dist as gdist
    from grafp_amd import ops
    from grafp_amd.train import Trainer, build_model, synthetic_batch
    from grafp_amd.util import load_config

    rank, world, device = gdist.init_from_env(backend=args.backend, local_device=args.local_device)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (the hot path has no CPU fallback)"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.backends.cudnn.benchmark = False
    cfg = load_config()
    if args.batch_per_gpu is None:
        assert args.global_batch % world == 0, f"global batch {args.global_batch} not divisible by {world} GPUs"
        B = args.global_batch // world
    else:
        B = args.batch_per_gpu
    cfg["bsz_train"] = B * world

    torch.manual_seed(1234)                                   # identical initial weights on every rank
    model = build_model(cfg, device=device)
    amp = torch.bfloat16 if args.dtype == "bf16" else None
    trainer = Trainer(cfg, model, device, amp_dtype=amp)
    x_i, x_j = synthetic_batch(B, seed=100 + rank, device=device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        t = torch.tensor([seconds], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- headline: K steps at the global batch, no per-kernel event records inside the timed region ----
    for _ in range(args.warmup):
        trainer.step(x_i, x_j)
    elapsed, loss = timed_steps(lambda: trainer.step(x_i, x_j), args.steps, barrier)
    elapsed = max_over_ranks(elapsed)
    if world > 1:
        loss = loss.clone()
        dist.all_reduce(loss)
    # ---- separate pass: HIP events around every hand-written kernel (every rank runs it: the collectives stay aligned)
    with ops.time_kernels(*KERNEL_NAMES) as timed:
        t_k, _ = timed_steps(lambda: trainer.step(x_i, x_j), args.kernel_steps, barrier)
        kernels = summarise_kernels(timed, 2 if args.dtype == "bf16" else 4)
    ms_k = 1e3 * max_over_ranks(t_k) / max(1, args.kernel_steps)

    weak = None
    if world > 1 and B != 256 and args.batch_per_gpu is None:
        # the weak-scaling variant beside the headline: 256 pairs per GPU (BASELINE config 2 on every GPU)
        cfg_w = dict(cfg, bsz_train=256 * world)
        trainer.sync.close()                                   # its gradient hooks must not fire in the next trainer's steps
        tw = Trainer(cfg_w, model, device, amp_dtype=amp)
        xw_i, xw_j = synthetic_batch(256, seed=300 + rank, device=device)
        for _ in range(2):
            tw.step(xw_i, xw_j)
        dt, _ = timed_steps(lambda: tw.step(xw_i, xw_j), max(3, args.steps // 2), barrier)
        dt = max_over_ranks(dt) / max(3, args.steps // 2)
        weak = {"value": round(256 * world / dt, 2), "unit": "clips/s", "ms_per_step": round(dt * 1e3, 3),
                "global_batch": 256 * world, "scaling": "weak", "note": "256 pairs per GPU, global negatives"}
        tw.sync.close()
        trainer.sync.open()                                    # the headline trainer listens again (graph leg below)
        del tw, xw_i, xw_j

    sharded = None
    if world > 1 and not args.no_retrieval:
        # BASELINE config 5 shape: the fingerprint database sharded across the GPUs (1.25 M x 128 rows per GPU, built
        # on the device; every rank plants its share of the queries in its own shard), one batched search =
        # local exact top-20 + all-gather of the (nq, 20) lists + merge.  Every rank runs the same collectives.
        n_local, nq_loc = 1_250_000, 4096 // world
        gen = torch.Generator(device=device).manual_seed(2 + rank)
        rows = torch.nn.functional.normalize(torch.randn(n_local, 128, generator=gen, device=device), dim=1)
        index = gdist.ShardedFlatL2Index(128)
        index.add_local(rows, rank * n_local, world * n_local)
        pick = torch.randint(0, n_local, (nq_loc,), generator=gen, device=device)
        q_loc = torch.nn.functional.normalize(rows[pick] + QUERY_SIGMA * torch.randn(nq_loc, 128, generator=gen, device=device), dim=1)
        q_all = torch.empty((world * nq_loc, 128), dtype=torch.float32, device=device)
        want = torch.empty((world * nq_loc,), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(q_all, q_loc.contiguous())
        dist.all_gather_into_tensor(want, (pick + rank * n_local).contiguous())
        index.search(q_all, 20)
        barrier()
        t0, reps = time.perf_counter(), 5
        for _ in range(reps):
            _, ids = index.search(q_all, 20)
        barrier()
        dt = max_over_ranks((time.perf_counter() - t0) / reps)
        sharded = {"db": f"{world} x {n_local} x 128 f32 shards resident (+ bf16 pre-filter copies)", "k": 20,
                   "nq": world * nq_loc, "ms_per_batch": round(dt * 1e3, 4), "qps": round(world * nq_loc / dt, 1),
                   "top1_hit_rate": float((ids[:, 0] == want).float().mean().item()), "n_gpus": world,
                   "query_sigma": QUERY_SIGMA}
        del index, rows

    # ---- what RCCL itself costs at this world size: the two collectives of a step, timed alone (HIP events, 20 calls) ----
    collectives = None
    if world > 1:
        def timed_coll(fn, reps=20):
            fn()
            barrier()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                fn()
            e.record()
            torch.cuda.synchronize()
            return max_over_ranks(s.elapsed_time(e) / reps * 1e-3) * 1e6        # microseconds, slowest rank
        mine = torch.randn(2, B, 128, device=device)
        gathered = torch.empty((2 * world, B, 128), device=device)
        sync = trainer.sync
        collectives = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                       "all_gather_z_bytes_per_rank": int(mine.numel() * 4),
                       "all_gather_z_us": round(timed_coll(lambda: dist.all_gather_into_tensor(gathered, mine)), 1),
                       "bucket_bytes": [int((hi - lo) * 4) for lo, hi in sync.bounds],
                       "bucket_all_reduce_us": [round(timed_coll(lambda lo=lo, hi=hi: dist.all_reduce(sync.flat[lo:hi])), 1)
                                                for lo, hi in sync.bounds],
                       "note": "each collective alone on an idle GPU, max over ranks; in a step the bucket all-reduces "
                               "overlap backward (eager) or follow it (step_graph)"}
        sync.flat.zero_()

    # ---- the same data-parallel step replayed from HIP graphs (forward | loss + backward + pack | Adam, the two
    #      collectives eager between them): every rank takes the same path, so the collectives stay aligned ----
    # (last of the legs: whatever happens here, everything above is already measured)
    hip_graph = None
    if world > 1 and not args.no_graph:
        try:
            err = None
            try:
                trainer.step_graph(x_i, x_j)
            except Exception as exc:      # noqa: BLE001
                err = exc
            # a rank that failed to capture must not leave the others waiting in the first collective of the timed loop
            ok = torch.tensor([0.0 if err is not None else 1.0], device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0:
                raise err if err is not None else RuntimeError("step_graph failed on another rank")
            dt, _ = timed_steps(lambda: trainer.step_graph(x_i, x_j), args.steps, barrier)
            dt = max_over_ranks(dt)
            hip_graph = {"value": round(B * world * args.steps / dt, 2), "unit": "clips/s",
                         "ms_per_step": round(1e3 * dt / args.steps, 3), "steps": args.steps,
                         "backward_graphs": len(trainer._graph[1][1]), "gradient_buckets": len(trainer.sync.bounds),
                         "bucket_bytes": [int((hi - lo) * 4) for lo, hi in trainer.sync.bounds],
                         "note": "forward graph | all-gather of (z_i, z_j) | one backward graph per gradient bucket, bucket "
                                 "b's all-reduce launched right behind graph b (it runs under the graphs that follow; the "
                                 "last bucket is the small tail bucket) | Adam graph (Trainer.step_graph); `value` above "
                                 "is the eager step"}
        except Exception as exc:          # noqa: BLE001 -- report, do not fail the bench line
            hip_graph = {"error": f"{type(exc).__name__}: {exc}"[:200]}

    if rank == 0:
        clips = B * world * args.steps
        # the kernel FAMILY with the largest measured time is the roofline's subject, whichever it is
        dom_name = max(kernels, key=lambda n: kernels[n]["total_ms"]) if kernels else None
        dom = kernels.get(dom_name, {})
        roof = {"kernel": dom_name, "bound": dom.get("bound"), "achieved": dom.get("achieved"), "peak": dom.get("peak"),
                "unit": dom.get("unit"), "frac": dom.get("frac"), "traffic": None, "avg_launch_us": dom.get("avg_us"),
                "launches": dom.get("calls"),
                "share_of_step": round(dom.get("total_ms", 0.0) / max(1, args.kernel_steps) / ms_k, 4),
                "algorithmic_bytes_per_launch": (round(dom["bytes"] / dom["calls"]) if dom.get("bytes") else None),
                "note": "the hand-written kernel family with the largest total time in the per-kernel pass; achieved = "
                        "algorithmic bytes (operands read once, results written once; flops for the exact-f32 k-NN) of "
                        "its launches / their HIP-event time on the launch stream"}
        if dom_name == "conv1x1_gemm":
            roof["mfma_frac"] = dom.get("mfma_frac")
            roof["tflops"] = dom.get("tflops")
            roof["two_ceiling_frac"] = dom.get("two_ceiling_frac")
            roof["note"] += "; the family mixes HBM-bound (stages 0-1) and matrix-bound (stages 2-3) shapes: both fractions are given, and two_ceiling_frac = sum over launches of max(bytes / 8 TB/s, flops / 2.5 PFLOP/s) / measured time"
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes over this same command
        # (FETCH_SIZE and WRITE_SIZE cannot share a pass); their committed summary is read back here.
        pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"pmc_{dom_name}.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pmc = json.load(f)
            if pmc.get("kernel_sources_sha16") != kernel_sources_sha16():
                roof["traffic_note"] = ("profiles/pmc_%s.json was collected on other kernel sources (stamp %s): not "
                                        "reported" % (dom_name, pmc.get("kernel_sources_sha16")))
            elif pmc.get("batch_per_gpu") == B and pmc.get("dtype") == args.dtype:
                # gfx950: FETCH_SIZE tallies the 128-B requests of 16 B/lane streams at 64 B -> doubled
                roof["traffic"] = int((2 * pmc["fetch_kb_per_launch"] + pmc["write_kb_per_launch"]) * 1024)
                roof["traffic_unit"] = "bytes/launch"
                roof["traffic_source"] = pmc.get("source")
        step_bytes = sum(v.get("bytes", 0.0) for v in kernels.values()) / max(1, args.kernel_steps)
        line = {
            "metric": "clips/sec contrastive step @ batch 1024", "value": round(clips / elapsed, 2), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "strong" if args.batch_per_gpu is None else "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"contrastive step at global batch {B * world} ({B} pairs/GPU x {world} GPU, global "
                                   "negatives), 1 s clips @16 kHz, mel->kNN-graph->GNN->NT-Xent, fwd+bwd+Adam, "
                                   "random-init GraphEncoder-t (18.4M params)",
                       "global_batch": B * world, "parallelism": f"dp{world}", "k": 3},
            "loss": round(float(loss.item()), 5),
            "parity": "bf16 mode: every stored activation and the embedding <= 1e-3 relative L2 vs the oracle's "
                      "bf16-storage restatement with inputs held equal, hit rates within 0.5 pt of f32 "
                      "(tests/test_gpu_bf16.py); f32 mode: <= 1e-4 vs the oracle (tests/test_gpu_model.py)",
            "roofline": roof,
            "step_roofline": {"bytes_per_step": round(step_bytes), "ms_per_step": round(ms_k, 3),
                              "achieved": round(step_bytes / (ms_k * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                              "frac": round(step_bytes / (ms_k * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                              "note": "sum of the algorithmic bytes of every timed hand-written kernel / wall time of a "
                                      "step of the per-kernel pass (event records add ~2 % to that step)"},
            "kernels": {k: {kk: vv for kk, vv in v.items() if kk != "bytes"} for k, v in kernels.items()},
        }
        line["step_impl"] = "eager (Trainer.step)"
        if hip_graph is not None:
            line["hip_graph"] = hip_graph
            # Data parallel, a rank's share of the batch is small and the eager step is bound by its ~700 launches
            # (128 pairs on one GPU: 16.7-18.3 ms eager, 15.8 replayed from graphs); the two are the SAME step (equal
            # parameters after a step: tests/test_gpu_model.py, tests/test_gpu_dist.py), both timed over K steps between
            # the same barriers, so the headline is the faster one and the other stays beside it.
            if isinstance(hip_graph.get("value"), (int, float)) and hip_graph["value"] > line["value"]:
                line["eager"] = {"value": line["value"], "unit": "clips/s", "ms_per_step": line["ms_per_step"],
                                 "steps": args.steps}
                line["value"], line["ms_per_step"] = hip_graph["value"], hip_graph["ms_per_step"]
                line["step_impl"] = ("HIP graphs (Trainer.step_graph: forward | one loss + backward graph per gradient bucket | "
                                     "Adam, the all-gather and the bucket all-reduces eager between them, each all-reduce "
                                     "under the backward graphs that follow it); the eager step of the same run is under "
                                     "`eager`, the per-kernel figures come from the eager pass")
        if collectives is not None:
            line["collectives"] = collectives
        if weak is not None:
            line["weak_scaling_256_per_gpu"] = weak
        if world == 1 and not args.no_config2:
            # BASELINE config 3's PER-GPU shape on this one GPU (128 pairs, no collectives): the compute part of one
            # rank of the 8-GPU run, eager and as one HIP graph.  8-GPU scaling >= 6x over the N = 1 line needs this
            # + the z all-gather + the 73.5 MB gradient all-reduce <= ms_per_step / 6.
            B3 = 128
            x3_i, x3_j = synthetic_batch(B3, seed=100, device=device)
            t3 = Trainer(dict(cfg, bsz_train=B3), model, device, amp_dtype=amp)
            for _ in range(3):
                t3.step(x3_i, x3_j)
            dt, _ = timed_steps(lambda: t3.step(x3_i, x3_j), args.steps, barrier)
            c3 = {"pairs_per_gpu": B3, "ms_per_step": round(1e3 * dt / args.steps, 3),
                  "clips_per_s_per_gpu": round(B3 * args.steps / dt, 2), "steps": args.steps,
                  "budget_ms_for_6x_at_8_gpus": round(1e3 * elapsed / args.steps / 6.0, 3),
                  "note": "one rank's compute at global batch 1024 on 8 GPUs, no collectives (measured on one GPU)"}
            if not args.no_graph:
                try:
                    t3.step_graph(x3_i, x3_j)
                    dt, _ = timed_steps(lambda: t3.step_graph(x3_i, x3_j), args.steps, barrier)
                    c3["hip_graph_ms_per_step"] = round(1e3 * dt / args.steps, 3)
                except Exception as exc:      # noqa: BLE001 -- report, do not fail the bench line
                    c3["hip_graph_error"] = f"{type(exc).__name__}: {exc}"[:200]
            del t3
            if not args.no_graph and not dist.is_initialized():
                # The same 128-pair step in its DATA-PARALLEL form on a ONE-RANK RCCL group (the most a one-GPU box can run):
                # forward graph -> RCCL all-gather -> one backward graph per gradient bucket with that bucket's RCCL all-reduce
                # launched behind it -> Adam graph.  Beside it the two collectives alone (the all-gather of the stacked
                # embeddings and the bucketed all-reduce of the 18.4 M f32 gradients), so that
                # "graph step + collectives <= budget" can be read off this line.  At one rank RCCL moves no bytes over xGMI:
                # these are launch + local-copy costs, the floor of what N = 8 adds.
                try:
                    import socket
                    with socket.socket() as sk:
                        sk.bind(("127.0.0.1", 0))
                        port = sk.getsockname()[1]
                    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                            device_id=device)
                    t4 = Trainer(dict(cfg, bsz_train=B3), model, device, amp_dtype=amp, data_parallel_graphs=True)
                    t4.step_graph(x3_i, x3_j)
                    dt, _ = timed_steps(lambda: t4.step_graph(x3_i, x3_j), args.steps, barrier)
                    c3["dp_graphs_rccl_one_rank_ms_per_step"] = round(1e3 * dt / args.steps, 3)
                    c3["dp_graphs"] = len(t4._graph[1][1]) + 2
                    mine = torch.zeros((2, B3, 128), device=device)
                    gathered = torch.empty_like(mine)
                    flat, bounds = t4.sync.flat, t4.sync.bounds

                    def collectives_once():
                        dist.all_gather_into_tensor(gathered, mine)
                        hs = [dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True) for lo, hi in bounds]
                        for h in hs:
                            h.wait()
                    for _ in range(5):
                        collectives_once()
                    dt, _ = timed_steps(collectives_once, 50, barrier)
                    c3["collectives_alone_rccl_one_rank_ms"] = round(1e3 * dt / 50, 3)
                    c3["gradient_bucket_bytes"] = [4 * (hi - lo) for lo, hi in bounds]
                    t4.sync.close()
                    del t4
                except Exception as exc:      # noqa: BLE001 -- report, do not fail the bench line
                    c3["dp_graphs_error"] = f"{type(exc).__name__}: {exc}"[:200]
                finally:
                    if dist.is_initialized():
                        dist.destroy_process_group()
            # What the one-GPU figures above CANNOT contain: at N = 8 a rank's loss runs its 128 local rows against the 2 048
            # gathered columns (first pass over all rows), where the single-process step above sees 256.  Timed alone
            # (HIP events around the launches) so that it can be added to the budget arithmetic.
            try:
                gen3 = torch.Generator(device=device).manual_seed(7)
                za = torch.nn.functional.normalize(torch.randn(1024, 128, generator=gen3, device=device), dim=1)
                zb = torch.nn.functional.normalize(za + 0.3 * torch.randn(1024, 128, generator=gen3, device=device), dim=1)
                la, lb = za[:B3].clone().requires_grad_(True), zb[:B3].clone().requires_grad_(True)
                sa, sb = za[:B3].clone().requires_grad_(True), zb[:B3].clone().requires_grad_(True)

                def loss_us(fn):
                    for _ in range(3):
                        fn().backward()
                    with ops.time_kernels("ntxent") as timed3:
                        for _ in range(20):
                            fn().backward()
                        torch.cuda.synchronize()
                        return round(1e3 * sum(a.elapsed_time(b) for a, b, _ in timed3["ntxent"]) / 20, 1)
                c3["loss_128_local_x_2048_global_us"] = loss_us(lambda: ops.ntxent(la, lb, cfg["tau"], za, zb, 0))
                c3["loss_128_x_256_single_process_us"] = loss_us(lambda: ops.ntxent(sa, sb, cfg["tau"]))
            except Exception as exc:          # noqa: BLE001 -- report, do not fail the bench line
                c3["loss_probe_error"] = f"{type(exc).__name__}: {exc}"[:200]
            line["config3_per_gpu_128"] = c3
            # measured parity figures of THIS model and build (numbers, not prose): the free-running distance of the bf16
            # mode from the f32 mode on 256 clip-views (eval mode, random-init weights) -- see tests/test_gpu_bf16.py for
            # the teacher-forced <= 1e-3 bars against the oracle's bf16-storage restatement
            keep = {k: v.clone() for k, v in model.state_dict().items()}       # the two passes advance the running statistics
            with torch.no_grad():
                S_i, S_j = trainer.augment(x_i[:128], x_j[:128])
                z32 = torch.cat(model(S_i, S_j)[2:]).float()
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    z16 = torch.cat(model(S_i, S_j)[2:]).float()
            model.load_state_dict(keep)
            segs_p = torch.cat((S_i, S_j))
            rel = torch.linalg.norm(z16 - z32, dim=1) / torch.linalg.norm(z32, dim=1)
            line["parity_measured"] = {"bf16_vs_f32_embedding_rel_l2_max": round(float(rel.max()), 5),
                                       "bf16_vs_f32_embedding_rel_l2_mean": round(float(rel.mean()), 5),
                                       "clip_views": int(segs_p.shape[0]),
                                       "note": "free-running bf16 mode vs f32 mode on the bench's own inputs: WHITE-NOISE clips through "
                                               "a random-init network, where every node's neighbours are nearly equidistant "
                                               "and the two modes build different k-NN graphs -- the worst case, not 1e-3.  On "
                                               "a briefly trained model and structured audio: mean 0.09, max 0.23 "
                                               "(tests/test_gpu_bf16.py, where the teacher-forced per-layer bars -- <= 1e-3 "
                                               "forward, <= 5e-3 backward against the oracle -- and the hit-rate bar over ten "
                                               "trained models also live).  Keeping the residual stream / the k-NN features / "
                                               "every activation in f32 with bf16 product operands was measured "
                                               "(tools/mixed_mode_probe.py, profiles/r04_mixed_mode_probe.txt): 0.104 against 0.107 "
                                               "-- the drift is the operand rounding of the products (DESIGN 10.4)"}
        if world == 1 and not args.no_config2:
            # BASELINE config 2: 256 pairs on one GPU -- eager, replayed from ONE HIP graph, and in f32
            B2 = 256
            cfg2 = dict(cfg, bsz_train=B2)
            x2_i, x2_j = synthetic_batch(B2, seed=100, device=device)
            t2 = Trainer(cfg2, model, device, amp_dtype=amp)
            for _ in range(3):
                t2.step(x2_i, x2_j)
            dt, _ = timed_steps(lambda: t2.step(x2_i, x2_j), args.steps, barrier)
            c2 = {"value": round(B2 * args.steps / dt, 2), "unit": "clips/s", "ms_per_step": round(1e3 * dt / args.steps, 3),
                  "steps": args.steps, "global_batch": B2}
            if not args.no_graph:
                try:
                    t2.step_graph(x2_i, x2_j)
                    dt, _ = timed_steps(lambda: t2.step_graph(x2_i, x2_j), args.steps, barrier)
                    c2["hip_graph"] = {"value": round(B2 * args.steps / dt, 2), "unit": "clips/s",
                                       "ms_per_step": round(1e3 * dt / args.steps, 3), "steps": args.steps,
                                       "note": "whole step (augment, forward, loss, backward, Adam) captured once, "
                                               "replayed per step; single process"}
                except Exception as exc:      # noqa: BLE001 -- report, do not fail the bench line
                    c2["hip_graph"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
            del t2
            if args.dtype == "bf16" and not args.no_f32_probe:
                t32 = Trainer(cfg2, model, device, amp_dtype=None)
                for _ in range(2):
                    t32.step(x2_i, x2_j)
                dt, _ = timed_steps(lambda: t32.step(x2_i, x2_j), args.steps, barrier)
                c2["f32_parity_mode"] = {"value": round(B2 * args.steps / dt, 2), "unit": "clips/s",
                                         "ms_per_step": round(1e3 * dt / args.steps, 3), "steps": args.steps,
                                         "note": "f32 GEMMs and f32 activations; embeddings <= 1e-4 relative L2 vs the "
                                                 "oracle with the k-NN edges held equal"}
                del t32
            line["config2_batch256"] = c2
            if args.dtype == "bf16" and not args.no_f32_probe:
                # the f32 ("parity") mode at the metric's own batch: the only mode whose embeddings meet the stated 1e-3
                # bar end to end (f32 activations, library f32 GEMMs forward / data gradient, split-bf16 weight gradient)
                t32 = Trainer(cfg, model, device, amp_dtype=None)
                for _ in range(2):
                    t32.step(x_i, x_j)
                n32 = max(3, args.steps // 4)
                dt, _ = timed_steps(lambda: t32.step(x_i, x_j), n32, barrier)
                line["f32_parity_mode_batch1024"] = {"value": round(B * n32 / dt, 2), "unit": "clips/s",
                                                     "ms_per_step": round(1e3 * dt / n32, 3), "steps": n32,
                                                     "global_batch": B}
                del t32
        if world == 1 and not args.no_retrieval:
            # BASELINE config 3/4, generation side: fingerprinting throughput of the forward pass alone (eval mode,
            # log-mel already computed), as generate.py / test_fp.py drive it, 1024 one-second segments per call
            model.eval()
            segs = trainer.augment(x_i, x_j)[0]
            segs = segs.repeat((1023 + segs.shape[0]) // segs.shape[0], 1, 1)[:1024]
            fp = {}
            for tag, amp_dt in (("f32", None), ("bf16", torch.bfloat16)):
                ctx = torch.autocast("cuda", dtype=amp_dt) if amp_dt is not None else contextlib.nullcontext()
                with torch.no_grad(), ctx:
                    model.embed(segs)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        model.embed(segs)
                    torch.cuda.synchronize()
                fp[f"segments_per_s_{tag}"] = round(3 * segs.shape[0] / (time.perf_counter() - t0), 1)
            # ... and END TO END as create_dummy_db drives it: 64 thirty-second tracks resident in HBM ->
            # log-mel -> overlapping 1 s segments -> packed model calls -> HBM -> (pinned) pages of the output memmap
            import shutil
            import tempfile
            from grafp_amd import fpdb
            from grafp_amd.modules.transformations import GPUTransformNeuralfp
            aug_eval = GPUTransformNeuralfp(cfg, None, None, train=False)
            gen = torch.Generator(device=device).manual_seed(11)
            tracks = 0.1 * torch.randn(64, 1, 30 * cfg["fs"], generator=gen, device=device)
            out_dir = tempfile.mkdtemp(prefix="grafp_fp_")
            try:
                for tag, amp_dt in (("f32", None), ("bf16", torch.bfloat16)):
                    ctx = torch.autocast("cuda", dtype=amp_dt) if amp_dt is not None else contextlib.nullcontext()
                    with ctx, contextlib.redirect_stdout(sys.stderr):
                        fpdb.create_dummy_db(list(tracks[:4]), aug_eval, model, out_dir, fname="warm", verbose=False,
                                             max_segments=1024)
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        fpdb.create_dummy_db(list(tracks), aug_eval, model, out_dir, fname=f"db_{tag}", verbose=False,
                                             max_segments=1024)
                        dt = time.perf_counter() - t0
                    n_seg = os.path.getsize(os.path.join(out_dir, f"db_{tag}.mm")) // 512
                    fp[f"end_to_end_segments_per_s_{tag}"] = round(n_seg / dt, 1)
                fp["end_to_end"] = (f"{n_seg} segments of 64 x 30 s tracks: waveform in HBM -> log-mel -> segments -> "
                                    "model (eval, 1024-segment calls) -> direct DMA into the output memmap, file closed")
            finally:
                shutil.rmtree(out_dir, ignore_errors=True)
            line["fingerprinting"] = fp
            model.train()
            line["retrieval"] = retrieval_probe(device, cpu_check=not args.no_cpu_baseline)
        if world == 1 and not args.no_augment:
            # the second view augmented on the device (impulse response + background noise for every
            # clip: ir_prob = noise_prob = 1 as in config/grafp.yaml), synthetic banks: 8 one-second decaying-noise
            # responses, 16 ten-second noise recordings; at 256 clips
            Ba = 256
            xa_i, xa_j = synthetic_batch(Ba, seed=100, device=device)
            gen = torch.Generator(device=device).manual_seed(5)
            irs = torch.randn(8, 16000, generator=gen, device=device) * \
                torch.exp(-torch.arange(16000, device=device, dtype=torch.float32) / 3000.0)
            noise = torch.randn(16, 160000, generator=gen, device=device)
            taug = Trainer(dict(cfg, bsz_train=Ba), model, device, amp_dtype=amp, ir_dir=irs, noise_dir=noise, aug_seed=0)
            tf = taug.augment
            pick = torch.randint(0, 8, (Ba,), generator=gen, device=device)
            off = torch.randint(0, 160000, (Ba,), generator=gen, device=device)
            snr = 20.0 * torch.rand(Ba, generator=gen, device=device)

            def timed(fn, reps):
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps
            dt_ir = timed(lambda: ops.ir_convolve(xa_j, tf.ir_bank, tf.ir_len, pick, tf.ir_start), 5)
            dt_mx = timed(lambda: ops.mix_snr(xa_j, tf.noise_bank, tf.noise_len, pick, off, snr, tf.noise_start), 20)
            Tn, Ln = xa_j.shape[1], 16000
            useful = 2.0 * Ba * (Tn * Ln - Ln * (Ln - 1) / 2.0)
            dt_step = timed(lambda: taug.step(xa_i, xa_j), max(3, args.steps // 2))
            line["augmentation"] = {
                "ir_convolve_ms": round(dt_ir * 1e3, 3), "ir_convolve_tflops": round(useful / dt_ir / 1e12, 1),
                "ir_convolve_peak_tflops": PEAK_F32_MATRIX_TFLOPS,
                "mix_snr_us": round(dt_mx * 1e6, 1), "mix_snr_gbs": round(3.0 * 4 * Ba * Tn / dt_mx / 1e9, 1),
                "step_ms_with_augmentation": round(dt_step * 1e3, 3),
                "clips_per_s_with_augmentation": round(Ba / dt_step, 2),
                "note": f"{Ba} one-second clips, one-second responses (useful flops 2*sum_t min(t+1, L)), x/noise/out "
                        "once for the mix; step = the 256-pair eager step with both transforms on every clip of view j"}
            del taug
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, args.cpu_baseline_seconds)
        if sharded is not None:
            line["retrieval_sharded"] = sharded
        # LAST key of the line (the driver keeps only the final ~2 000 characters of stdout): every kernel family as
        # [average launch in us, fraction of its roofline] and the handful of step figures the verdict's gates name
        def pick(obj, *path):
            for k in path:
                obj = obj.get(k) if isinstance(obj, dict) else None
            return obj
        line["kernels_digest"] = {
            "kernels_avg_us_frac": {k: [v.get("avg_us"), v.get("frac")] for k, v in kernels.items()},
            "ms_1024": line["ms_per_step"], "bytes_per_step_gb": round(step_bytes / 1e9, 1),
            "ms_256": pick(line, "config2_batch256", "ms_per_step"),
            "ms_256_graph": pick(line, "config2_batch256", "hip_graph", "ms_per_step"),
            "ms_128": pick(line, "config3_per_gpu_128", "ms_per_step"),
            "ms_128_graph": pick(line, "config3_per_gpu_128", "hip_graph_ms_per_step"),
            "ms_128_dp_graphs_rccl1": pick(line, "config3_per_gpu_128", "dp_graphs_rccl_one_rank_ms_per_step"),
            "retrieval_frac_nq1_nq41_nq4096": [pick(line, "retrieval", "roofline", n, "frac") for n in ("nq1", "nq41", "nq4096")],
            "retrieval_ms_nq4096": pick(line, "retrieval", "ms_per_batch_nq4096"),
        }
        sys.stdout.flush()
        os.write(line_fd, (json.dumps(line) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
