"""Data-parallel path with the real HIP kernels: two ranks share cuda:0 over gloo (the GPU box has one GPU, so RCCL
itself is out of reach here; everything around the collectives -- shard, all-gather of (z_i, z_j), local rows x
global columns NT-Xent, flat-buffer bucketed gradient all-reduce from autograd hooks, sharded search + merge -- is
the production code).  Compared with ONE process that pushes the two shards through the model one after the other
(per-replica BatchNorm statistics, as under the reference's DataParallel, train.py:165-168) and takes the loss over
the concatenated batch."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _shared_device_env(rank, world):
    """Ranks that share ONE device run on disjoint CU sets (grafp_amd.dist.shared_device_cu_mask says why; two unmasked
    processes repeating the 128-pair step: 7 differing iterations of 118, profiles/r06_contention_single.txt)."""
    from grafp_amd.dist import shared_device_cu_mask
    cus = torch.cuda.get_device_properties(0).multi_processor_count        # 256 on an MI355X in its default partition mode
    return {"ROC_GLOBAL_CU_MASK": shared_device_cu_mask(rank, world, cus)} if world > 1 else {}


def _launch(world, out, B, extra=()):
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29653", HSA_ENABLE_IPC_MODE_LEGACY="0", **_shared_device_env(r, world))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_gpu_worker.py"), out, str(B), *extra],
                                      env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        log, _ = p.communicate(timeout=600)
        assert p.returncode == 0, log[-3000:]


def test_two_ranks_match_one_process(tmp_path):
    from grafp_amd import ops
    B = 8
    out = str(tmp_path / "w2")
    from _common import eval_case, golden, write_eval_case
    case = eval_case()
    evdir = tmp_path / "ev"
    evdir.mkdir()
    write_eval_case(str(evdir), case)
    import numpy as np
    np.save(evdir / "ids.npy", case["test_ids"])
    _launch(2, out, B, extra=(str(evdir),))
    got = [torch.load(f"{out}.{r}.pt", weights_only=False) for r in range(2)]
    # eval_faiss over the two-rank sharded index == the table the reference's own eval_faiss produced
    g = golden("eval_faiss.npz")
    for r in range(2):
        np.testing.assert_array_equal(got[r]["rates"], g["hit_rates"])

    device = torch.device("cuda:0")
    _compare_with_one_process(out, B, 2, got)

    # sharded search == unsharded search, on both ranks
    gen = torch.Generator().manual_seed(3)
    db = torch.nn.functional.normalize(torch.randn(5000, 128, generator=gen), dim=1)
    q = torch.nn.functional.normalize(db[::97][:20] + 0.05 * torch.randn(20, 128, generator=gen), dim=1)
    dbd = db.to(device)
    D, I = ops.search_l2(dbd, ops.row_sqnorm(dbd), q.to(device), 10)
    for r in range(2):
        assert torch.equal(got[r]["I"], I.cpu()) and torch.equal(got[r]["D"], D.cpu())
    # sharded sequence rerank (rows + halo per rank, per-shard top lists merged by score) == unsharded rerank
    starts = torch.tensor([40, 2495, 3100, 4989])
    qs = torch.cat([db[s0:s0 + 11] for s0 in starts.tolist()])
    qs = torch.nn.functional.normalize(qs + 0.05 * torch.randn(qs.shape, generator=gen), dim=1).to(device)
    _, Iq = ops.search_l2(dbd, ops.row_sqnorm(dbd), qs, 10)
    item_row = (torch.arange(4).repeat_interleave(3) * 11).to(device)
    item_len = torch.tensor([1, 5, 11] * 4, dtype=torch.int32, device=device)
    wid, wsc = ops.seq_rerank(dbd, qs, Iq, item_row, item_len, top=10)
    for r in range(2):
        assert torch.equal(got[r]["rid"], wid.cpu()) and torch.equal(got[r]["rsc"], wsc.cpu())
    assert (wid[:, 0].cpu().reshape(4, 3) == starts[:, None]).all()          # planted runs found, across the boundary


def _compare_with_one_process(out, B, world, got):
    """The ranks' embeddings, loss shares, reduced gradients and gradient norm against ONE process that pushes the `world`
    shards through the model one after the other (tests/_dist_gpu_worker.py reference_mode).  That side runs in a process
    of its own under a CU mask of the ranks' width: the f32 mode's library GEMMs choose their splits by the number of
    CUs they see."""
    env = dict(os.environ, **_shared_device_env(0, world))
    p = subprocess.run([sys.executable, os.path.join(HERE, "_dist_gpu_worker.py"), out, str(B), "reference", str(world)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    ref = torch.load(f"{out}.ref.pt", weights_only=False)
    per = B // world
    for r in range(world):
        assert got[r]["z_i"].shape == (per, ref["z_i"][r].shape[1])
        assert torch.allclose(got[r]["z_i"], ref["z_i"][r], rtol=0, atol=1e-6), r
    loss_v = ref["loss"]
    assert abs(sum(g["loss_share"] for g in got) - loss_v) <= 1e-5 * max(1.0, abs(loss_v))
    rel = {}
    for name, g in got[0]["grads"].items():
        want = ref["grads"][name]
        rel[name] = float((g - want).norm() / want.norm().clamp_min(1e-12))
        for r in range(1, world):
            assert torch.equal(g, got[r]["grads"][name]), (name, r)    # every rank holds the same reduced gradient
    # (all names in one message: a mismatch that grows towards the first layers points at one backward kernel, the same
    #  factor everywhere at the loss / the reduction)
    assert max(rel.values()) < 2e-4, rel
    want_norm = ref["grad_norm"]
    for r in range(world):
        assert abs(got[r]["grad_norm"] - want_norm) <= 2e-4 * want_norm


def _ranks_against_one_process(tmp_path, world, B, tag):
    """`world` ranks on cuda:0 over gloo, B // world pairs each, against ONE process pushing the shards through the model
    one after the other (per-replica BatchNorm statistics, as under the reference's DataParallel, train.py:165-168) with
    the loss over the concatenated batch (train.py:69-71)."""
    out = str(tmp_path / tag)
    _launch(world, out, B)
    got = [torch.load(f"{out}.{r}.pt", weights_only=False) for r in range(world)]
    device = torch.device("cuda:0")
    _compare_with_one_process(out, B, world, got)
    return got, device


def test_two_ranks_at_128_pairs_per_rank(tmp_path):
    """The per-GPU shape of BASELINE config 3 (128 pairs per rank) on two ranks: global-negative loss shares, summed
    gradients and their norm against ONE process pushing the two shards through the model one after the other."""
    _ranks_against_one_process(tmp_path, 2, 256, "w2big")


def test_eight_ranks_at_128_pairs_per_rank(tmp_path):
    """BASELINE config 3 exactly -- global batch 1024 on EIGHT ranks of 128 pairs -- with all eight processes on cuda:0
    over gloo (288 GB holds eight replicas; RCCL needs eight devices, everything around the collectives is the production
    code): the rank-major all-gather of 8 x (2, 128, 128) embeddings, row_begin = rank * 128 against 2046 global
    negatives per row, the SUM of eight flat gradient buffers in arrival-order buckets, and 8-way shard_range + merge +
    halo rerank on the retrieval side (/root/reference/train.py:69-71,165-168; eval.py:269-290)."""
    from grafp_amd import ops
    got, device = _ranks_against_one_process(tmp_path, 8, 1024, "w8")
    gen = torch.Generator().manual_seed(3)
    db = torch.nn.functional.normalize(torch.randn(5000, 128, generator=gen), dim=1)
    q = torch.nn.functional.normalize(db[::97][:20] + 0.05 * torch.randn(20, 128, generator=gen), dim=1)
    dbd = db.to(device)
    D, I = ops.search_l2(dbd, ops.row_sqnorm(dbd), q.to(device), 10)
    starts = torch.tensor([40, 2495, 3100, 4989])                       # 2495..2505 straddles the shard boundary at 2500
    qs = torch.cat([db[s0:s0 + 11] for s0 in starts.tolist()])
    qs = torch.nn.functional.normalize(qs + 0.05 * torch.randn(qs.shape, generator=gen), dim=1).to(device)
    _, Iq = ops.search_l2(dbd, ops.row_sqnorm(dbd), qs, 10)
    item_row = (torch.arange(4).repeat_interleave(3) * 11).to(device)
    item_len = torch.tensor([1, 5, 11] * 4, dtype=torch.int32, device=device)
    wid, wsc = ops.seq_rerank(dbd, qs, Iq, item_row, item_len, top=10)
    for r in range(8):
        assert torch.equal(got[r]["I"], I.cpu()) and torch.equal(got[r]["D"], D.cpu()), r
        assert torch.equal(got[r]["rid"], wid.cpu()) and torch.equal(got[r]["rsc"], wsc.cpu()), r


def test_step_graph_data_parallel(tmp_path):
    """Trainer.step_graph on two ranks (forward graph -> all-gather -> loss/backward/pack graph -> bucket all-reduces ->
    Adam graph): from the same state a replayed step returns the eager step's loss share and moves the parameters
    where the eager step moves them, and both ranks end with the same parameters."""
    out = str(tmp_path / "g2")
    _launch(2, out, 32, extra=("graph",))
    got = [torch.load(f"{out}.{r}.pt", weights_only=False) for r in range(2)]
    for r in range(2):
        for step in got[r]:
            assert abs(step["loss_g"] - step["loss_e"]) <= 2e-3 * max(1.0, abs(step["loss_e"])), step   # bf16 step
            assert step["d_e"] > 0 and step["d_diff"] < 0.05 * step["d_e"], step
    for a, b in zip(got[0], got[1]):
        assert a["p_sum"] == b["p_sum"]                 # replicas stay in step (same reduced gradients, same update)


@pytest.mark.parametrize("mode", ["graph_rccl1", "graph_rccl1_single"])
def test_step_graph_data_parallel_over_rccl_one_rank(tmp_path, mode):
    """The same comparison on the RCCL backend (a one-rank process group: the most a one-GPU box can do): the all-gather
    and the bucket all-reduces are RCCL launches between the replayed graphs -- by default one backward graph per gradient
    bucket with bucket b's all-reduce launched behind graph b (GradSync.begin_capture), or, with
    Trainer(overlap_graph_allreduce=False), one backward graph and every bucket behind it."""
    out = str(tmp_path / "g1")
    _launch(1, out, 16, extra=(mode,))
    for step in torch.load(f"{out}.0.pt", weights_only=False):
        assert abs(step["loss_g"] - step["loss_e"]) <= 2e-3 * max(1.0, abs(step["loss_e"])), step
        assert step["d_e"] > 0 and step["d_diff"] < 0.05 * step["d_e"], step
        # the replayed step really issued one RCCL all-reduce per gradient bucket (a one-rank group reduces too when the
        # flat layout is forced: GradSync._reduce)
        assert step["n_allreduce"] == len(step["bucket_numel"]) >= 4, step


def test_bucket_all_reduces_are_released_before_backward_ends(tmp_path):
    """in graph mode the gradient all-reduces overlap backward BY DEFAULT.  Backward is replayed as one
    graph per bucket; a stream that waits for the main stream where bucket b's all-reduce is launched (what RCCL's stream
    does) is released when graph b has run, i.e. before the backward graphs behind it finish: its marker completes a
    measurable time before the end of the last backward graph, for every bucket but the last.  The last bucket is the small
    tail bucket (the parameters whose gradients arrive last), so the exposed all-reduce is a few MB."""
    out = str(tmp_path / "tl")
    _launch(1, out, 64, extra=("graph_rccl1_timeline",))
    for step in torch.load(f"{out}.0.pt", weights_only=False):
        assert abs(step["loss_g"] - step["loss_e"]) <= 2e-3 * max(1.0, abs(step["loss_e"])), step
        tl = step["ms_before_backward_end"]
        sizes = step["bucket_numel"]
        assert step["n_allreduce"] == len(sizes), step                    # RCCL all-reduces interleaved with the graphs
        print("graphs per step:", step["n_graphs"], "bucket sizes:", sizes, "markers (buckets, ms before backward's end):", tl)
        assert len(tl) == len(sizes) and [b for b, _ in tl] == [[i] for i in range(len(sizes))]      # in completion order
        assert sizes[-1] <= (1 << 20) < max(sizes)                                      # the tail bucket is the small one
        lead = [ms for _, ms in tl]
        assert all(a > b for a, b in zip(lead, lead[1:]))                                # released one after the other
        assert all(ms > 0.05 for ms in lead[:-1]), lead                                  # ... and BEFORE backward ends
        # everything but the tail bucket is released with most of backward's run time still ahead
        assert lead[len(lead) - 2] > 0.2 * lead[0], lead


def test_bench_two_ranks_one_gpu(tmp_path):
    """bench.py's N > 1 path end to end (barriers, MAX over ranks, rank-0 JSON line, sharded retrieval leg) with both
    ranks on cuda:0 -- started in the PLAIN form `python bench.py --gpus 2 ...` with no launcher around it and no
    WORLD_SIZE in the environment: bench.py starts its own ranks (a child `python -m torch.distributed.run`, before
    anything touches the GPU) and forwards rank 0's line and the exit code."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--global-batch", "32", "--kernel-steps", "1",
                        "--no-cpu-baseline", "--backend", "gloo", "--local-device", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    logs = [p.stdout, ""]
    lines = [ln for ln in logs[0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not any(ln.startswith("{") for ln in logs[1].splitlines())
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0      # the global batch is split
    assert line["config"]["global_batch"] == 32 and line["roofline"]["frac"] > 0
    assert line["hip_graph"].get("value", 0) > 0, line["hip_graph"]                        # the graph-replayed data-parallel step
    assert line["hip_graph"]["backward_graphs"] == line["hip_graph"]["gradient_buckets"] >= 4           # overlap by default
    assert line["hip_graph"]["bucket_bytes"][-1] <= 4 << 20                                  # ... and a small tail bucket
    # the headline is the faster of the two implementations of the step, the other one stays beside it
    if line["step_impl"].startswith("HIP graphs"):
        assert line["value"] == line["hip_graph"]["value"] and line["ms_per_step"] == line["hip_graph"]["ms_per_step"]
        assert 0 < line["eager"]["value"] <= line["value"]
    else:
        assert line["step_impl"].startswith("eager") and "eager" not in line and line["value"] >= line["hip_graph"]["value"]
    w = line["weak_scaling_256_per_gpu"]                                                   # ... and 256 pairs per rank beside it
    assert w["global_batch"] == 512 and w["scaling"] == "weak" and w["value"] > 0
    rs = line["retrieval_sharded"]                 # config-5 shape: one 1.25 M-row shard per rank, planted queries
    assert rs["n_gpus"] == 2 and rs["nq"] == 4096 and 0.5 < rs["top1_hit_rate"] < 1.0 and rs["qps"] > 0   # informative sigma


def test_step_is_bit_stable_next_to_other_processes_on_disjoint_cus():
    """Four processes share cuda:0, each on its own quarter of the CUs, and repeat the same 128-pair bf16 forward +
    backward six times: every log-mel spectrogram, embedding, loss and gradient must be the same bits each time, while
    the other three processes load HBM, the Infinity Cache and the queues, and with 64 CUs per process the single-pass
    BatchNorm rendezvous regularly takes its recompute path (a row's workgroups no longer fit at once).  WITHOUT the
    disjoint CU sets this does not hold on this platform (tools/contention/step_stress.py)."""
    root = os.path.dirname(HERE)
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "contention", "step_stress.py"), "4", "6", "bf16",
                        "default", "--disjoint-cus"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if "iterations, BAD" in ln]
    assert p.returncode == 0 and len(lines) == 4, p.stdout[-3000:]
    assert all(ln.rstrip().endswith("BAD 0") for ln in lines), p.stdout[-3000:]
