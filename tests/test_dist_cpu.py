"""World-size-2 and world-size-8 gloo tests (CPU) of the data-parallel plumbing.  The compute step is injected (the
oracle), because the product's kernels have no CPU path: what is under test is the collective logic of grafp_amd.dist.
World 8 is the layout of BASELINE config 3 / config 5 (/root/reference/train.py:69-71,165-168: eight replicas, global
negatives; eval.py:285: candidate sequences across shard boundaries)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _common import hash_normalish


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _oracle_local_loss(z_i, z_j, tau, zi_all, zj_all, row_begin):
    """Same contract as ops.ntxent's data-parallel form: value = local share of the global mean loss,
    gradient = d(global loss)/d(local z)."""
    from oracle import model as om
    n = z_i.shape[0]
    ai, aj = zi_all.clone(), zj_all.clone()
    ai[row_begin:row_begin + n] = z_i
    aj[row_begin:row_begin + n] = z_j
    full = om.ntxent(ai, aj, tau)
    M = 2 * ai.shape[0]
    z = torch.cat([ai, aj]).detach()
    s = z @ z.T / tau
    s.fill_diagonal_(float("-inf"))
    rows = torch.cat([torch.arange(row_begin, row_begin + n), torch.arange(row_begin, row_begin + n) + ai.shape[0]])
    share = (torch.logsumexp(s[rows], 1) - s[rows, (rows + ai.shape[0]) % M]).sum() / M
    return share.detach() + (full - full.detach())


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from grafp_amd import dist as gdist
    from oracle import model as om, native
    r, w, dev = gdist.init_from_env(backend="gloo")
    assert (r, w, dev.type) == (rank, world, "cpu")
    torch.manual_seed(0)
    torch.set_num_threads(1)
    res = {}
    # ---- global-negative NT-Xent: shares add up, local gradients are slices of the global gradient
    B, D = (12 if world <= 4 else 3 * world), 32          # equal shares per rank (world 8: 24 pairs, 3 per rank)
    zi = torch.nn.functional.normalize(torch.from_numpy(hash_normalish("dist:zi", (B, D))), dim=1)
    zj = torch.nn.functional.normalize(torch.from_numpy(hash_normalish("dist:zj", (B, D))), dim=1)
    lo, hi = rank * B // world, (rank + 1) * B // world
    a = zi[lo:hi].clone().requires_grad_(True); b = zj[lo:hi].clone().requires_grad_(True)
    gi, gj = gdist.all_gather_embeddings(a, b)
    assert torch.equal(gi, zi) and torch.equal(gj, zj) and not gi.requires_grad
    share = gdist.ntxent_global(a, b, 0.1, loss_fn=_oracle_local_loss)
    share.backward()
    tot = share.detach().clone(); dist.all_reduce(tot)
    fa = zi.clone().requires_grad_(True); fb = zj.clone().requires_grad_(True)
    full = om.ntxent(fa, fb, 0.1); full.backward()
    res["loss_ok"] = bool(torch.allclose(tot, full.detach(), rtol=1e-5))
    res["grad_ok"] = bool(torch.allclose(a.grad, fa.grad[lo:hi], rtol=1e-4, atol=1e-7) and
                          torch.allclose(b.grad, fb.grad[lo:hi], rtol=1e-4, atol=1e-7))
    # ---- flat-buffer gradient sync: SUM over ranks, .grad stays a view, buckets fire from hooks
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 2))
    frozen = torch.nn.Parameter(torch.ones(3), requires_grad=False)
    sync = gdist.GradSync(list(net.parameters()) + [frozen], n_buckets=3)
    assert len(sync.bounds) >= 2 and sync.flat.numel() == sum(p.numel() for p in net.parameters())
    import copy
    shadow = copy.deepcopy(net)            # un-synchronised twin: the bucket all-reduces start DURING backward
    for it in range(2):
        sync.zero()
        x = torch.from_numpy(hash_normalish(f"dist:x{rank}.{it}", (5, 8)))
        net(x).square().sum().backward()
        shadow.zero_grad()
        shadow(x).square().sum().backward()
        local = [p.grad.clone() for p in shadow.parameters()]
        sync.finish()
        ok = True
        for p, g in zip(net.parameters(), local):
            want = g.clone(); dist.all_reduce(want)
            ok &= bool(torch.allclose(p.grad, want, rtol=1e-6, atol=1e-7))
            ok &= p.grad.data_ptr() >= sync.flat.data_ptr() and p.grad.data_ptr() < sync.flat.data_ptr() + 4 * sync.flat.numel()
        res[f"gradsync_{it}"] = ok
    # ---- a step that dies mid-backward (every rank: same place) must not poison the next one: zero() resets the hook
    #      counters and waits for the collectives that already fired
    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t): return t.clone()
        @staticmethod
        def backward(ctx, g): raise RuntimeError("boom")
    sync.zero()
    x = torch.from_numpy(hash_normalish(f"dist:xb{rank}", (5, 8)))
    try:
        net[3](Boom.apply(net[2](net[1](net[0](x))))).square().sum().backward()    # the last layer's bucket fires first
        raised = False
    except RuntimeError:
        raised = True
    sync.zero()
    net(x).square().sum().backward()
    shadow.zero_grad(); shadow(x).square().sum().backward()
    local = [p.grad.clone() for p in shadow.parameters()]
    sync.finish()
    ok = raised
    for p, g in zip(net.parameters(), local):
        want = g.clone(); dist.all_reduce(want)
        ok &= bool(torch.allclose(p.grad, want, rtol=1e-6, atol=1e-7))
    res["gradsync_after_failed_backward"] = ok
    # ---- a parameter that takes no part in a step contributes zero, and ONLY its slice is cleared: the other members
    #      of its bucket keep what the flat buffer already holds for them (accumulation without zero())
    sync2 = gdist.GradSync(list(net.parameters()), n_buckets=1, overlap=False)
    for p in net.parameters():
        p.grad = None
    net[0](x).square().sum().backward()                                             # only the first layer gets gradients
    g0 = [net[0].weight.grad.clone(), net[0].bias.grad.clone()]
    sync2.finish()
    ok = True
    for p, g in zip(net[0].parameters(), g0):
        want = g.clone(); dist.all_reduce(want)
        ok &= bool(torch.allclose(p.grad, want, rtol=1e-6, atol=1e-7))
    ok &= all(float(p.grad.abs().max()) == 0.0 for p in list(net[2].parameters()) + list(net[3].parameters()))
    net[0](x).square().sum().backward()          # accumulates INTO the flat views; the untouched layers stay as they are
    held = [p.grad.clone() for p in net[0].parameters()]
    for p in list(net[2].parameters()) + list(net[3].parameters()):
        p.grad = None
    sync2.finish()
    for p, h in zip(net[0].parameters(), held):
        want = h.clone(); dist.all_reduce(want)
        ok &= bool(torch.allclose(p.grad, want, rtol=1e-6, atol=1e-7))
    res["gradsync_partial_step"] = ok
    # ---- sharded exact search: local top-k + all-gather + merge == unsharded
    db = hash_normalish("dist:db", (1001, 128)); q = db[5:30] + 0.05 * hash_normalish("dist:q", (25, 128))

    class OracleIndex:
        def __init__(self, id_base): self.id_base, self.x = id_base, None
        def add(self, x): self.x = np.ascontiguousarray(x)
        def search(self, qq, k): return native.flat_search_l2(self.x, qq, k, id_base=self.id_base)

    def merge(gd, gi):
        d, i = native.merge_topk(gd.numpy(), gi.numpy())
        return torch.from_numpy(d), torch.from_numpy(i)
    idx = gdist.ShardedFlatL2Index(128, local_index_factory=OracleIndex, merge_fn=merge)
    idx.add_global(db)
    D, I = idx.search(q, 20)
    wd, wi = native.flat_search_l2(db, q, 20)
    idx2 = gdist.ShardedFlatL2Index(128, local_index_factory=OracleIndex, merge_fn=merge)
    lo, hi = gdist.shard_range(len(db), rank, world)
    idx2.add_local(db[lo:hi], lo, len(db))                 # shard-aware loading: only the own rows
    D2, I2 = idx2.search(q, 20)
    res["search_local_ok"] = bool(np.array_equal(np.asarray(I2), wi) and np.array_equal(np.asarray(D2), wd))
    res["shard_rows"] = gdist.shard_range(1001, rank, world)
    res["search_ok"] = bool(np.array_equal(np.asarray(I), wi) and np.array_equal(np.asarray(D), wd))
    # ---- the buckets follow the gradients' ARRIVAL order (rank 0's, broadcast) at this world size too: a module
    #      registered last and applied first ends up in the last bucket on every rank, and the sums stay right
    torch.manual_seed(7)
    body = [torch.nn.Linear(16, 16) for _ in range(4)]
    head, first = torch.nn.Linear(16, 16), torch.nn.Linear(16, 16)
    mods = body + [head, first]
    for m in mods:                                      # replicas start equal (seeded), as after a broadcast
        for q_ in m.parameters():
            dist.broadcast(q_.data, src=0)
    prm = [q_ for m in mods for q_ in m.parameters()]

    def fwd(ms, x):
        x = ms[5](x)
        for m in ms[:4]:
            x = x + torch.relu(m(x))
        return ms[4](x).square().sum()
    sync3 = gdist.GradSync(prm, n_buckets=3, tail_numel=0)
    twin = copy.deepcopy(torch.nn.ModuleList(mods))
    ok, was_first = True, sync3._bucket_of[id(first.weight)]
    for it in range(3):
        sync3.zero()
        x = torch.from_numpy(hash_normalish(f"dist:relay{rank}.{it}", (4, 16)))
        fwd(mods, x).backward()
        twin.zero_grad(); fwd(list(twin), x).backward()
        sync3.finish()
        for q_, t_ in zip(prm, twin.parameters()):
            want = t_.grad.clone(); dist.all_reduce(want)
            mag = t_.grad.abs(); dist.all_reduce(mag)             # gloo's ring adds the ranks' terms in another order
            ok &= bool(((q_.grad - want).abs() <= 2e-6 * mag + 1e-12).all())         # per slice: a few ulps of sum |g_r|
    layout = torch.tensor([sync3._bucket_of[id(q_)] for q_ in prm])
    lay0 = layout.clone(); dist.broadcast(lay0, src=0)
    res["relayout_ok"] = bool(ok and sync3._relaid and was_first == 0 and torch.equal(layout, lay0)
                              and sync3._bucket_of[id(first.weight)] == len(sync3.bounds) - 1
                              and sync3._bucket_of[id(head.weight)] == 0)
    # ---- sharded sequence rerank (eval.py:272-290 over shard_range + halo): planted runs, one straddling EVERY shard
    #      boundary; per-shard top lists merged == the unsharded rerank, ids and scores bit for bit
    def oracle_rerank(rows, q_rows, topk_ids, item_row, item_len, top=10, shard=None, max_len=None):
        row_base, n_total, id_lo, id_hi = shard
        rows_np, ids_out, sc_out = rows.numpy(), [], []
        assert row_base + len(rows_np) <= n_total
        for r0, ql in zip(item_row.tolist(), item_len.tolist()):
            tk = topk_ids[r0:r0 + ql].clone()
            start = tk - torch.arange(ql)[:, None]
            tk = torch.where((tk >= 0) & (start >= id_lo) & (start < id_hi), tk - row_base, torch.full_like(tk, -1))
            own = start[(tk >= 0) & (start >= id_lo) & (start < id_hi)]
            assert own.numel() == 0 or id_hi == n_total or int(own.max()) + ql <= row_base + len(rows_np)   # halo suffices
            oi, os_ = native.seq_rerank(rows_np, q_rows[r0:r0 + ql].numpy(), tk.numpy(), [0], [ql], top=top)
            ids_out.append(np.where(oi >= 0, oi + row_base, -1)); sc_out.append(os_)
        return torch.from_numpy(np.concatenate(ids_out)), torch.from_numpy(np.concatenate(sc_out))

    class OracleRows(OracleIndex):
        device = torch.device("cpu")
        def rows(self): return torch.from_numpy(self.x)
    per = (len(db) + world - 1) // world
    starts = sorted({3} | {min(len(db) - 11, b * per - 4) for b in range(1, world)} | {len(db) - 6})
    qs = np.concatenate([np.concatenate([db[s0:s0 + 11], np.zeros((max(0, s0 + 11 - len(db)), 128), np.float32)])[:11]
                         for s0 in starts]).astype(np.float32)
    qs = qs + 0.05 * hash_normalish("dist:qs", qs.shape)
    idx3 = gdist.ShardedFlatL2Index(128, local_index_factory=OracleRows, merge_fn=merge, halo=10,
                                    rerank_fn=oracle_rerank)
    idx3.add_global(db)
    _, Iq = idx3.search(qs, 10)
    item_row = np.repeat(np.arange(len(starts)) * 11, 3)
    item_len = np.tile(np.array([1, 5, 11], np.int32), len(starts))
    rid, rsc = idx3.rerank(qs, Iq, item_row, item_len, top=10)
    _, wIq = native.flat_search_l2(db, qs, 10)
    wid, wsc = native.seq_rerank(db, qs, wIq, item_row, item_len, top=10)
    res["rerank_ok"] = bool(np.array_equal(rid.numpy(), wid) and np.array_equal(rsc.numpy(), wsc)
                            and np.array_equal(np.asarray(Iq), wIq)
                            and np.array_equal(wid[2::3, 0], np.asarray(starts)))       # the planted 11-runs found
    try:
        idx3.halo = 9                                   # 11-segment items need 10 halo rows
        idx3.rerank(qs, Iq, item_row, item_len, top=10)
        res["rerank_halo_guard"] = world == 1
    except ValueError:
        res["rerank_halo_guard"] = True
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert set(res) == {0, 1}
    for rank, r in res.items():
        assert r["loss_ok"] and r["grad_ok"] and r["gradsync_0"] and r["gradsync_1"] and r["search_ok"] and r["search_local_ok"] and \
            r["gradsync_after_failed_backward"] and r["gradsync_partial_step"] and r["relayout_ok"] and r["rerank_ok"] and \
            r["rerank_halo_guard"], (rank, r)
    assert res[0]["shard_rows"] == (0, 501) and res[1]["shard_rows"] == (501, 1001)


def test_world_size_8_gloo():
    """The 8-rank layout of BASELINE configs 3 and 5, once, on gloo: rank-major all-gather of the stacked (z_i, z_j) and
    row_begin = rank * B_loc (shares add up, local gradients are slices of the global gradient); the flat-buffer SUM
    with buckets fired from hooks; the arrival-order re-layout broadcast from rank 0; 8-way shard_range + merge ==
    unsharded search; sequence rerank with a planted run across every one of the 7 shard boundaries
    (/root/reference/train.py:69-71,165-168; eval.py:269-290)."""
    world, port = 8, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert set(res) == set(range(8))
    for rank, r in res.items():
        assert all(r[k] for k in ("loss_ok", "grad_ok", "gradsync_0", "gradsync_1", "search_ok", "search_local_ok",
                                  "gradsync_after_failed_backward", "gradsync_partial_step", "relayout_ok", "rerank_ok",
                                  "rerank_halo_guard")), (rank, {k: v for k, v in r.items() if v is not True})
    assert [res[r]["shard_rows"] for r in range(8)] == [(126 * r, min(1001, 126 * (r + 1))) for r in range(8)]


def test_single_process_paths():
    from grafp_amd import dist as gdist
    assert gdist.world_size() == 1 and gdist.rank_of() == 0
    a, b = torch.randn(4, 8), torch.randn(4, 8)
    gi, gj = gdist.all_gather_embeddings(a, b)
    assert torch.equal(gi, a) and torch.equal(gj, b)
    assert [gdist.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    net = torch.nn.Linear(3, 2)
    sync = gdist.GradSync(net.parameters())
    sync.zero(); net(torch.ones(1, 3)).sum().backward(); sync.finish()
    assert torch.allclose(net.weight.grad, torch.ones(2, 3))


def test_grad_buckets_tail_and_graph_cuts():
    """GradSync's bucket layout and the bookkeeping Trainer.step_graph's per-bucket backward graphs rest on (no process
    group needed: force_flat).  A small tail bucket holds the parameters whose gradients arrive last; during a recorded
    backward pass every bucket but the last calls `cut` right after it was packed, in completion order, and end_capture()
    reports which buckets are complete behind which graph -- also when some parameters get no gradient."""
    from grafp_amd.dist import GradSync
    torch.manual_seed(0)
    # registration order = forward order: two small early layers (their gradients arrive LAST), three large late ones
    sizes = [8, 8, 400, 400, 400]
    layers = [torch.nn.Linear(16, s) for s in sizes]
    heads = [torch.nn.Linear(s, 16) for s in sizes]

    def forward(x, skip=()):
        for i, (a, b) in enumerate(zip(layers, heads)):
            if i not in skip:
                x = x + b(torch.relu(a(x)))
        return x.sum()
    params = [p for a, b in zip(layers, heads) for p in list(a.parameters()) + list(b.parameters())]
    total = sum(p.numel() for p in params)
    sync = GradSync(params, n_buckets=2, force_flat=True, tail_numel=600)
    sizes_b = [hi - lo for lo, hi in sync.bounds]
    assert sum(sizes_b) == total and sync.bounds[0][0] == 0 and all(a[1] == b[0] for a, b in zip(sync.bounds, sync.bounds[1:]))
    assert len(sizes_b) == 3 and sizes_b[-1] <= 600 < min(sizes_b[:-1])          # two large buckets + the small tail
    tail = {id(p) for p in sync._members[-1]}
    assert tail == {id(p) for i in (0, 1) for m in (layers[i], heads[i]) for p in m.parameters()}
    # a recorded backward pass: cuts after bucket 0 and bucket 1, none after the last
    cuts = []
    sync.zero()
    loss = forward(torch.randn(4, 16))
    sync.begin_capture(cuts.append)
    loss.backward()
    ready = sync.end_capture()
    assert cuts == [0, 1] and ready == [[0], [1], [2]]
    want = torch.cat([p.grad.flatten() for p in reversed(params)])
    assert torch.equal(sync.flat, want)                                           # every bucket was packed
    sync.reduce_buckets([0, 1, 2]); sync.wait_reduced()                           # world 1: nothing to reduce
    assert all(p.grad.data_ptr() == sync._view[id(p)].data_ptr() for p in params)
    # half of a bucket's parameters take no part: that bucket never completes inside backward -> packed by end_capture
    cuts.clear()
    sync.zero()
    loss = forward(torch.randn(4, 16), skip=(3,))
    sync.begin_capture(cuts.append)
    loss.backward()
    ready = sync.end_capture()
    b3 = sync._bucket_of[id(layers[3].weight)]
    assert b3 not in cuts and ready[-1][-1] == b3 or b3 in ready[-1]
    assert sorted(b for r in ready for b in r) == [0, 1, 2] and len(ready) == len(cuts) + 1
    for p in list(layers[3].parameters()) + list(heads[3].parameters()):
        assert float(sync._view[id(p)].abs().sum()) == 0.0                       # its slice was cleared, nothing else
    # without a cut function: one graph, every bucket behind it
    sync.zero()
    loss = forward(torch.randn(4, 16))
    sync.begin_capture(None)
    loss.backward()
    assert sync.end_capture() == [[0, 1, 2]]


def test_bench_refuses_fewer_devices_than_ranks():
    """`python bench.py --gpus N` without a launcher starts its own ranks -- or, when fewer than N devices are visible,
    exits non-zero with a message instead of silently running one rank (missing #1).  No GPU here: N = 2
    must be refused before anything else happens."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.device_count() >= 2:
        return
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and "refusing to run fewer ranks" in p.stderr and not p.stdout.strip()


def test_grad_buckets_follow_the_observed_arrival_order():
    """A module registered LATE but used FIRST (SimCLR registers the peak extractor after the encoder; its gradient is
    the last of the pass) must not hold an early bucket open until backward ends: after the first complete backward pass
    the buffer is laid out in the order the gradients really arrived, once."""
    from grafp_amd.dist import GradSync
    torch.manual_seed(0)
    body = [torch.nn.Linear(16, 16) for _ in range(4)]
    head = torch.nn.Linear(16, 16)
    first = torch.nn.Linear(16, 16)                       # registered last, applied first
    params = [p for m in body + [head, first] for p in m.parameters()]

    def forward(x):
        x = first(x)
        for m in body:
            x = x + torch.relu(m(x))
        return head(x).sum()
    sync = GradSync(params, n_buckets=3, force_flat=True, tail_numel=0)
    assert sync._bucket_of[id(first.weight)] == 0                                  # the registration-order guess
    for step in range(3):
        sync.zero()
        forward(torch.randn(4, 16)).backward()
        sync.finish()
        want = {id(p): p.grad.clone() for p in params}
        assert all(p.grad.data_ptr() == sync._view[id(p)].data_ptr() for p in params)
        if step >= 1:
            assert sync._relaid and sync._bucket_of[id(first.weight)] == len(sync.bounds) - 1   # now in the LAST bucket
            assert sync._bucket_of[id(head.weight)] == 0
    # the layout is fixed after the one re-layout, every element of the buffer belongs to exactly one parameter
    covered = torch.zeros(sync.flat.numel(), dtype=torch.int32)
    for p in params:
        off = sync._view[id(p)].data_ptr() - sync.flat.data_ptr()
        covered[off // 4: off // 4 + p.numel()] += 1
        assert torch.equal(sync._view[id(p)], want[id(p)])
    assert bool((covered == 1).all())


def test_shared_device_cu_masks_are_disjoint_and_cover_the_device():
    """grafp_amd.dist.shared_device_cu_mask: the CU slices of ranks that share one device (test layouts only; DESIGN.md
    section 12.7b) are disjoint, equally wide and together cover the 256 CUs."""
    from grafp_amd.dist import shared_device_cu_mask
    for world in (2, 3, 4, 8):
        masks = [int(shared_device_cu_mask(r, world), 16) for r in range(world)]
        width = 256 // world
        for i, m in enumerate(masks):
            assert bin(m).count("1") == width and m >> 256 == 0
            for n in masks[i + 1:]:
                assert m & n == 0
        if 256 % world == 0:
            assert sum(masks) == (1 << 256) - 1
