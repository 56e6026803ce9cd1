"""Worker of tests/test_gpu_dist.py: one of WORLD_SIZE ranks sharing cuda:0 over gloo
(init_from_env(backend='gloo', local_device=0)).  Runs one data-parallel training step on its shard of a fixed batch with the real HIP
kernels and writes its loss share, a few all-reduced gradients and its sharded-search result to OUT.rank.pt."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from grafp_amd import dist as gdist                                    # noqa: E402
from grafp_amd.train import Trainer, build_model, synthetic_batch      # noqa: E402
from grafp_amd.util import load_config                                 # noqa: E402


def graph_mode(out, B, rccl_one_rank=False, overlap=True, timeline=False):
    """Trainer.step_graph under data parallelism (forward graph, one backward graph per gradient bucket, Adam graph, eager
    collectives between) against Trainer.step from the same weights and optimizer state, on this rank's shard: losses
    and parameter updates.
    rccl_one_rank: a ONE-rank process group on the `nccl` (= RCCL) backend with the data-parallel graph form forced --
    the only way to run RCCL's launches between the replayed graphs on a one-GPU box;
    overlap=False: ONE backward graph with every bucket reduced behind it (Trainer(overlap_graph_allreduce=False));
    timeline: also record WHEN work ordered behind each bucket's graph (where its all-reduce goes) completes relative to
    the end of the last backward graph."""
    if rccl_one_rank:
        rank, world, device = 0, 1, torch.device("cuda", 0)
        torch.cuda.set_device(0)
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    else:
        rank, world, device = gdist.init_from_env(backend="gloo", local_device=0)
    cfg = load_config()
    cfg["bsz_train"] = B
    # every later allocation of this process comes out of blocks full of NaNs (the caching allocator keeps them): a
    # kernel that reads what nobody wrote -- fine on a fresh box where memory is zero -- shows up here
    junk = torch.full((1 << 30,), float("nan"), device=device)
    del junk
    torch.manual_seed(1234)
    model = build_model(cfg, device=device)
    tr = Trainer(cfg, model, device, amp_dtype=torch.bfloat16, data_parallel_graphs=True if rccl_one_rank else None,
                 overlap_graph_allreduce=overlap)
    assert tr._dp_graphs
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    x_i, x_j = synthetic_batch(B, 7, device)
    tr.step_graph(x_i[sl], x_j[sl])                                # warm-up, capture, first replay
    n_parts = len(tr._graph[1][1])
    assert n_parts == (len(tr.sync.bounds) if overlap else 1), (n_parts, len(tr.sync.bounds))
    marks = []
    if timeline:
        # where reduce_buckets() launches bucket b's all-reduce, ALSO put a marker on a side stream that waits for the main
        # stream's position at that moment (what RCCL's own stream does): a small kernel + an event.  After the step:
        # how long before the end of the last backward graph did each marker complete?
        side = torch.cuda.Stream()
        scratch = torch.zeros(1 << 16, device=device)
        orig_reduce, orig_wait = tr.sync.reduce_buckets, tr.sync.wait_reduced

        def reduce_and_mark(buckets):
            orig_reduce(buckets)
            here = torch.cuda.Event()
            here.record()
            side.wait_event(here)
            with torch.cuda.stream(side):
                scratch.add_(1.0)
                done = torch.cuda.Event(enable_timing=True)
                done.record()
            marks[-1].append((list(buckets), done))

        def wait_and_mark():
            end = torch.cuda.Event(enable_timing=True)
            end.record()                                             # behind the last backward graph
            marks[-1].append(("end", end))
            orig_wait()
        tr.sync.reduce_buckets, tr.sync.wait_reduced = reduce_and_mark, wait_and_mark

    def snapshot():
        return ([p.detach().clone() for p in model.parameters()], [b.detach().clone() for b in model.buffers()],
                [{k: v.detach().clone() for k, v in st.items() if torch.is_tensor(v)} for st in tr.opt.state.values()])

    def restore(snap):
        with torch.no_grad():
            for p, v in zip(model.parameters(), snap[0]):
                p.copy_(v)
            for b, v in zip(model.buffers(), snap[1]):
                b.copy_(v)
            for st, sv in zip(tr.opt.state.values(), snap[2]):
                for k, v in sv.items():
                    st[k].copy_(v)

    # count the collectives a replayed step really issues (at one rank reduce_buckets() used to issue none)
    calls = {"all_reduce": 0}
    orig_all_reduce = torch.distributed.all_reduce

    def counted_all_reduce(*a, **k):
        calls["all_reduce"] += 1
        return orig_all_reduce(*a, **k)
    torch.distributed.all_reduce = counted_all_reduce

    res = []
    for seed in (51, 52):
        y_i, y_j = synthetic_batch(B, seed, device)
        snap = snapshot()
        loss_e = float(tr.step(y_i[sl], y_j[sl]))
        p_e = torch.cat([p.detach().flatten() for p in model.parameters()]).clone()
        restore(snap)
        marks.append([])
        calls["all_reduce"] = 0
        loss_g = float(tr.step_graph(y_i[sl], y_j[sl]))
        n_allreduce = calls["all_reduce"]
        p_g = torch.cat([p.detach().flatten() for p in model.parameters()]).clone()
        p_0 = torch.cat([v.flatten() for v in snap[0]])
        res.append({"loss_e": loss_e, "loss_g": loss_g, "d_e": float((p_e - p_0).norm()),
                    "d_diff": float((p_g - p_e).norm()), "p_sum": float(p_g.double().sum()), "n_graphs": n_parts + 2,
                    "n_allreduce": n_allreduce, "bucket_numel": [hi - lo for lo, hi in tr.sync.bounds]})
        if timeline:
            torch.cuda.synchronize()
            end = [e for tag, e in marks[-1] if tag == "end"][0]
            res[-1]["ms_before_backward_end"] = [(b, e.elapsed_time(end)) for b, e in marks[-1] if b != "end"]
    torch.cuda.synchronize()
    torch.save(res, f"{out}.{rank}.pt")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def reference_mode(out, B, world):
    """ONE process pushes the `world` shards of the batch through the model one after the other (per-replica BatchNorm
    statistics, as under the reference's DataParallel, train.py:165-168) and takes the loss over the concatenated batch
    (train.py:69-71): the embeddings, the loss, every gradient and their norm -> OUT.ref.pt.  A process of its own so
    that it can run under the same ROC_GLOBAL_CU_MASK width as the ranks: the library GEMMs of the f32 mode pick their
    splits by the number of CUs they see, i.e. their low bits follow it."""
    from grafp_amd.simclr.ntxent import ntxent_loss
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    cfg = load_config()
    cfg["bsz_train"] = B
    torch.manual_seed(1234)
    model = build_model(cfg, device=device)
    trainer = Trainer(cfg, model, device, amp_dtype=None)
    x_i, x_j = synthetic_batch(B, 7, device)
    model.train()
    per = B // world
    zs_i, zs_j = [], []
    for r in range(world):
        with torch.no_grad():
            X_i, X_j = trainer.augment(x_i[r * per:(r + 1) * per], x_j[r * per:(r + 1) * per])
        _, _, z_i, z_j = model(X_i, X_j)
        zs_i.append(z_i); zs_j.append(z_j)
    loss = ntxent_loss(torch.cat(zs_i), torch.cat(zs_j), cfg)
    loss.backward()
    grads = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}
    norm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).item()
    torch.save({"z_i": [z.detach().cpu() for z in zs_i], "loss": float(loss.detach()), "grads": grads, "grad_norm": norm},
               f"{out}.ref.pt")


def main():
    out, B = sys.argv[1], int(sys.argv[2])
    if len(sys.argv) > 4 and sys.argv[3] == "reference":
        return reference_mode(out, B, int(sys.argv[4]))
    if len(sys.argv) > 3 and sys.argv[3] in ("graph", "graph_rccl1", "graph_rccl1_single", "graph_rccl1_timeline"):
        return graph_mode(out, B, rccl_one_rank=sys.argv[3] != "graph", overlap=sys.argv[3] != "graph_rccl1_single",
                          timeline=sys.argv[3] == "graph_rccl1_timeline")
    rank, world, device = gdist.init_from_env(backend="gloo", local_device=0)
    cfg = load_config()
    cfg["bsz_train"] = B
    torch.manual_seed(1234)
    model = build_model(cfg, device=device)
    trainer = Trainer(cfg, model, device, amp_dtype=None)
    x_i, x_j = synthetic_batch(B, 7, device)
    per = B // world
    lo = rank * per
    # the step of Trainer.step without the optimizer update, so that the gradients can be inspected
    model.train()
    trainer.sync.zero()
    with torch.no_grad():
        X_i, X_j = trainer.augment(x_i[lo:lo + per], x_j[lo:lo + per])
    _, _, z_i, z_j = model(X_i, X_j)
    loss = gdist.ntxent_global(z_i, z_j, cfg["tau"])
    loss.backward()
    trainer.sync.finish()
    torch.cuda.synchronize()
    names = ["encoder.stem.0.weight", "encoder.backbone.0.0.fc1.0.weight", "encoder.backbone.7.1.fc2.0.weight",
             "encoder.proj.bias", "projector.0.weight", "peak_extractor.convs.0.weight"]
    params = dict(model.named_parameters())
    grads = {n: params[n].grad.detach().float().cpu().clone() for n in names if n in params}
    total = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in model.parameters() if p.grad is not None)).item()

    # sharded exact search on the same ranks
    gen = torch.Generator().manual_seed(3)
    db = torch.nn.functional.normalize(torch.randn(5000, 128, generator=gen), dim=1)
    q = torch.nn.functional.normalize(db[::97][:20] + 0.05 * torch.randn(20, 128, generator=gen), dim=1)
    index = gdist.ShardedFlatL2Index(128)
    index.add_global(db.numpy())
    D, I = index.search(q.numpy(), 10)
    # sharded sequence rerank: planted 11-segment runs, one of them straddling the shard boundary (rows 2495..2505)
    starts = torch.tensor([40, 2495, 3100, 4989])
    qs = torch.cat([db[s0:s0 + 11] for s0 in starts.tolist()])
    qs = torch.nn.functional.normalize(qs + 0.05 * torch.randn(qs.shape, generator=gen), dim=1)
    _, Iq = index.search(qs.numpy(), 10)
    item_row = torch.arange(4).repeat_interleave(3) * 11
    item_len = torch.tensor([1, 5, 11] * 4, dtype=torch.int32)
    rid, rsc = index.rerank(qs, torch.as_tensor(Iq), item_row, item_len, top=10)
    rates = None
    if len(sys.argv) > 3:                       # eval_faiss over the sharded index on the files of sys.argv[3]
        from grafp_amd.eval import eval_faiss_sharded
        rates = eval_faiss_sharded(sys.argv[3], index_type="l2", test_ids=os.path.join(sys.argv[3], "ids.npy"),
                                   test_seq_len="1 3 5 9", k_probe=20)
    torch.save({"loss_share": float(loss), "grads": grads, "grad_norm": total, "z_i": z_i.detach().cpu(),
                "D": torch.as_tensor(D).cpu(), "I": torch.as_tensor(I).cpu(), "rid": rid.cpu(), "rsc": rsc.cpu(),
                "rates": rates}, f"{out}.{rank}.pt")
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
