"""World-size-2 gloo tests (CPU) of the data-parallel plumbing.  The compute step is injected (the oracle),
because the product's kernels have no CPU path: what is under test is the collective logic of grafp_amd.dist."""
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
    res = {}
    # ---- global-negative NT-Xent: shares add up, local gradients are slices of the global gradient
    B, D = 12, 32
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
            r["gradsync_after_failed_backward"] and r["gradsync_partial_step"], (rank, r)
    assert res[0]["shard_rows"] == (0, 501) and res[1]["shard_rows"] == (501, 1001)


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
