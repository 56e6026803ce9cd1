"""Data parallelism: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) / xGMI.

The reference's only multi-GPU mechanism is nn.DataParallel (train.py:165-168): per step it broadcasts
the 82.7 MB state dict, scatters the batch, gathers h/z to GPU 0 (which makes the NT-Xent negatives
global, train.py:69-71) and reduces 73.5 MB of gradients to GPU 0.  Here:
  * parameters are replicated once (DDP style), BatchNorm statistics stay per replica as under DataParallel;
  * ONE all-gather of the stacked (z_i, z_j) per step (2 * B/R * 128 f32 per rank: latency-bound, so both
    views travel in a single call); each rank then evaluates its own rows against the global columns with
    ops.ntxent, whose backward already returns d(global loss)/d(local z): no collective in backward;
  * gradients are packed bucket by bucket (one multi-tensor copy each) into ONE flat f32 buffer and summed
    with a few large all-reduces launched from autograd hooks as soon as a bucket is complete (overlap
    with the rest of backward); afterwards every .grad is a view of the reduced buffer.  xGMI is point-to-point, ring collectives are per-link bound (~153 GB/s):
    few, large buckets (default 4 x ~18 MB) keep the ring busy without paying per-call latency 100+ times;
  * the fingerprint database is sharded by contiguous row ranges; a search is a local top-k, one
    all-gather of (nq, k) candidates and a local merge.
Everything that touches a collective takes the compute step as an argument, so the plumbing is testable
on CPU with the gloo backend (tests/test_dist_cpu.py) without any CPU fallback in the product path.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def shared_device_cu_mask(rank, world, cus=256):
    """ROC_GLOBAL_CU_MASK of rank `rank` when `world` processes share ONE device: disjoint, equal slices of its CUs.
    Only the test layouts do that (production is one process per GPU).  Why they need it: on MI355X / ROCm 7.2 a
    packed-f32 instruction whose low lane reads the high register of a pair (`v_pk_add_f32 ... op_sel:[0,1]`; hipcc emits it
    in the log-mel, single-pass BatchNorm and peak-extractor kernels) returns wrong values on lanes 48-63 while bf16 MFMA
    waves of ANOTHER kernel -- another process's or stream's GEMM -- run on the same SIMD (DESIGN.md section 12.7b,
    tools/contention/two_stream.py); with disjoint CU sets the step is bit-reproducible (0 of 240 iterations against 20-34
    of 120).  NOTE the f32 mode's library GEMMs
    pick their splits by the number of CUs they see: results under a mask differ in the low bits (1e-4 on an embedding)
    from those on the whole device -- compare like with like (tests/test_gpu_dist.py runs its one-process side under a
    mask of the same width)."""
    per = max(1, cus // max(1, world))
    return hex(((1 << per) - 1) << (per * (rank % max(1, cus // per))))


def init_from_env(backend=None, local_device=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as set by torch.distributed.run.  Returns (rank, world, device).
    backend / local_device: for a box with fewer GPUs than ranks (the tests run two ranks on cuda:0 over gloo: same
    code path, no RCCL); production leaves both None (one GPU per rank = LOCAL_RANK, RCCL)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if local_device is not None and world > 1:
        # ranks sharing a device (test layouts): disjoint CU sets, see shared_device_cu_mask; a launcher that set the
        # variable itself wins.  Measured: the HIP runtime reads it at its first call, not when torch is imported, so
        # this is early enough
        os.environ.setdefault("ROC_GLOBAL_CU_MASK", shared_device_cu_mask(rank, world))
    if torch.cuda.is_available():
        if local_device is not None:
            local = int(local_device)
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or ("nccl" if device.type == "cuda" else "gloo")
        kw = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, device


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank_of(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def all_gather_embeddings(z_i, z_j, group=None):
    """(B_loc, D) x 2 -> (R*B_loc, D) x 2, rank-major, detached.  One collective for both views."""
    R = world_size(group)
    zi, zj = z_i.detach().float().contiguous(), z_j.detach().float().contiguous()
    if R == 1:
        return zi, zj
    mine = torch.stack((zi, zj), dim=0)                                   # (2, B_loc, D)
    out = torch.empty((R * 2,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)                   # concatenated along dim 0
    out = out.reshape(R, 2, *mine.shape[1:]).permute(1, 0, 2, 3)          # (2, R, B_loc, D)
    return out[0].reshape(-1, zi.shape[1]).contiguous(), out[1].reshape(-1, zi.shape[1]).contiguous()


def ntxent_global(z_i, z_j, tau, group=None, loss_fn=None):
    """NT-Xent with negatives from every rank.  Returns this rank's SHARE of the global mean loss (the
    shares sum to the loss the reference computes on the gathered batch); backward gives the gradient of the
    GLOBAL loss w.r.t. the local embeddings.  loss_fn(z_i, z_j, tau, zi_all, zj_all, row_begin)."""
    if loss_fn is None:
        from . import ops
        loss_fn = ops.ntxent
    zi_all, zj_all = all_gather_embeddings(z_i, z_j, group)
    return loss_fn(z_i, z_j, tau, zi_all, zj_all, rank_of(group) * z_i.shape[0])


class GradSync:
    """Flat-buffer gradient all-reduce (SUM) with bucketed launch from autograd hooks.

    The sum (not the mean) is the right reduction: each rank back-propagates d(global loss)/d(z_local),
    so the per-rank parameter gradients are the disjoint terms of the full gradient, exactly what
    DataParallel's reduce-to-GPU-0 adds up (train.py:165-168)."""

    def __init__(self, params, group=None, n_buckets=4, overlap=True, force_flat=False, tail_numel=1 << 20):
        """n_buckets near-equal buckets in the order the gradients become ready, plus -- when the last of them would hold
        more than `tail_numel` elements -- a small TAIL bucket for the parameters whose gradients arrive last (the stem and
        stages 0-1: 0.6 M of the 18.4 M parameters, but half of backward's run time): the one all-reduce that nothing is
        left to overlap with is then a few MB (latency-bound), not a quarter of the gradient."""
        self.group = group
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, total = self.params[0].device, sum(p.numel() for p in self.params)
        self.world = world_size(group)
        # collectives are issued whenever there is something to reduce -- and also on a ONE-rank process group when the
        # caller forces the flat layout (the one-GPU tests of the data-parallel graph step: the RCCL all-reduce of every
        # bucket then really runs between the replayed backward graphs, on the shared memory pool, as it does at N > 1)
        self._reduce = self.world > 1 or (force_flat and dist.is_available() and dist.is_initialized())
        self.flat = None
        if self.world == 1 and not force_flat:
            # nothing to reduce: let autograd ASSIGN fresh gradients (zero() drops them) instead of launching one
            # tiny accumulate kernel per parameter into pre-existing views
            self.bounds, self._hooks, self._handles = [], [], []
            return
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self._n_buckets, self._tail_numel = n_buckets, tail_numel
        # gradients become ready ROUGHLY in reverse registration order: start with that layout.  The first complete
        # backward pass records the order in which they really arrive, and the next zero() re-lays the buffer out in
        # it (as DDP rebuilds its buckets after the first iteration): a module registered late but used first -- the
        # peak extractor sits between the projector and the encoder in SimCLR's registration order, and its gradient is
        # the LAST of the whole pass -- would otherwise keep an early bucket open until backward ends.
        self._layout(list(reversed(self.params)))
        self._arrival, self._arrived, self._relaid = [], set(), False
        self._hooks = []
        self._cap_cut = None
        self.paused = False            # Trainer.step_graph: backward is being captured, the buckets are reduced afterwards
        self.capturing = False         # ... and each completed bucket ends one backward graph (begin_capture)
        self._overlap = bool(overlap and (self.world > 1 or force_flat))
        self.open()

    def _layout(self, order):
        """Cut the flat buffer into buckets for the parameters in `order` (expected arrival order of their gradients)."""
        total, n_buckets, tail_numel = self.flat.numel(), self._n_buckets, self._tail_numel
        per_bucket = (total + n_buckets - 1) // max(1, n_buckets)
        self.bounds, self._bucket_of, self._view, self._members, off, b_lo = [], {}, {}, [[]], 0, 0
        tail_cut = False
        for p in order:
            n = p.numel()
            if (not tail_cut and tail_numel and off > b_lo and total - off <= tail_numel
                    and total - b_lo > tail_numel):
                # everything from here on fits the tail bucket: close the running bucket in front of it
                self.bounds.append((b_lo, off))
                self._members.append([])
                b_lo, tail_cut = off, True
            self._view[id(p)] = self.flat[off:off + n].view_as(p)
            self._bucket_of[id(p)] = len(self.bounds)
            self._members[-1].append(p)
            off += n
            if off - b_lo >= per_bucket:
                self.bounds.append((b_lo, off))
                self._members.append([])
                b_lo = off
        if b_lo < off:
            self.bounds.append((b_lo, off))
        self._members = self._members[:len(self.bounds)]
        self._size = [len(m) for m in self._members]
        self._left, self._handles, self._fired = list(self._size), [], [False] * len(self.bounds)

    def _maybe_relayout(self):
        """Once, at the start of the step that follows the first COMPLETE backward pass (every parameter's gradient
        arrived through the hooks): buckets in the observed arrival order.  Every rank runs the same autograd graph, so
        the orders agree; rank 0's is broadcast anyway (a layout that differed between ranks would add up the wrong
        slices), and nothing is in flight here (zero() has just waited)."""
        if self._relaid or len(self._arrival) != len(self.params):
            self._arrival, self._arrived = [], set()
            return
        index = {id(p): i for i, p in enumerate(self.params)}
        order = torch.tensor([index[i] for i in self._arrival], dtype=torch.int64)
        if self.world > 1:
            dev_order = order.to(self.flat.device)
            dist.broadcast(dev_order, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                           group=self.group)
            order = dev_order.cpu()
        self._layout([self.params[i] for i in order.tolist()])
        self._relaid = True
        self._arrival, self._arrived = [], set()

    def zero(self):
        """Use instead of optimizer.zero_grad(): autograd then ASSIGNS fresh gradients (no per-parameter accumulate
        kernels); finish() leaves every .grad a view of the flat buffer.  Also the start of a step for the hook state:
        a backward pass that raised, or a finish() that was skipped (out of memory, a NaN guard), must not leave
        half-counted buckets or un-waited collectives behind -- every rank then resets here, in the same place."""
        for p in self.params:
            p.grad = None
        if self.flat is not None:
            for h in self._handles:
                h.wait()
            self._handles, self._left, self._fired = [], list(self._size), [False] * len(self.bounds)
            if not self._relaid and not self.capturing:
                self._maybe_relayout()

    def open(self):
        """Re-attach the autograd hooks after close() (bench.py lends the model to a second Trainer in between)."""
        if self.flat is not None and not self._hooks and self._overlap:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        if self.flat is not None and self.flat.is_cuda:
            from . import ops
            ops._GRAD_TARGET_OF = self._target_of       # weight gradients are produced in place (ops._wgrad_bf16)

    def close(self):
        """Detach the autograd hooks (a second GradSync over the same parameters -- another Trainer on the same model --
        must not find this one still listening) and wait for anything in flight."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for h in self._handles:
            h.wait()
        self._handles = []
        if self.flat is not None and self.flat.is_cuda:
            from . import ops
            if getattr(ops._GRAD_TARGET_OF, "__self__", None) is self:
                ops._GRAD_TARGET_OF = None

    def _target_of(self, p):
        """The slice of the flat buffer a kernel may write parameter p's gradient into directly -- only while p has no
        gradient yet (autograd then takes the tensor over; an existing .grad would be ADDED to, i.e. to itself)."""
        return self._view.get(id(p)) if (p is not None and p.grad is None) else None

    def _pack(self, b):
        """Bucket b's gradients -> its slice of the flat buffer: one multi-tensor copy.  A parameter that took no part
        in this step (grad None) contributes zero: ONLY its own slice is cleared -- members whose .grad already is
        their flat view (gradient accumulation without zero()) keep what they hold."""
        members = self._members[b]
        if self.flat.is_cuda:                         # gradients whose reduction is still queued (ops.defer_wgrad_reduce)
            from . import ops
            ops.flush_wgrad_reduce()
        missing = [p for p in members if p.grad is None]
        if missing:
            torch._foreach_zero_([self._view[id(p)] for p in missing])
        have = [p for p in members if p.grad is not None and p.grad.data_ptr() != self._view[id(p)].data_ptr()]
        if have:
            torch._foreach_copy_([self._view[id(p)] for p in have], [p.grad for p in have])

    def _launch(self, b):
        self._pack(b)
        lo, hi = self.bounds[b]
        if self._reduce:
            self._handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))
        self._fired[b] = True

    def begin_capture(self, cut=None):
        """Backward is about to be recorded into HIP graphs (Trainer.step_graph).  The hooks then launch no collective;
        when a bucket's last gradient arrives they record the bucket's pack (one multi-tensor copy into the flat buffer)
        and -- except for the last bucket -- call `cut(b)`, which ends the graph being recorded and begins the next one:
        backward becomes one graph per bucket, and on replay bucket b's all-reduce is launched between graph b and graph
        b + 1, i.e. it runs under the rest of backward exactly like the eager step's (no RCCL call is captured, no device-
        side polling)."""
        self._cap_left, self._cap_done = list(self._size), [False] * len(self.bounds)
        self._cap_cut = cut
        self._cap_order = []
        self.capturing = True

    def _cap_complete(self, b):
        self._pack(b)
        self._cap_done[b] = True
        self._cap_order.append(b)
        if self._cap_cut is not None and not all(self._cap_done):
            self._cap_cut(b)

    def end_capture(self):
        """Still inside the capture, after backward: buckets whose parameters did not all receive a gradient are packed
        into the last graph.  Returns, per recorded graph, the buckets whose packed gradients are complete when that
        graph has run (the last entry may hold several)."""
        cuts = [[b] for b in self._cap_order]
        tail = []
        for b in range(len(self.bounds)):
            if not self._cap_done[b]:
                self._pack(b)
                self._cap_done[b] = True
                tail.append(b)
        if self._cap_cut is None:
            cuts = [[b for c in cuts for b in c] + tail]
        elif cuts and len(cuts) == len(self.bounds) and not tail:
            pass                                     # the last bucket completed inside the last graph: no cut behind it
        else:
            cuts.append(tail)
        self.capturing, self._cap_cut = False, None
        return cuts

    def reduce_buckets(self, buckets):
        """Launch (do not wait for) the all-reduce of the given buckets' slices of the flat buffer: the replay side of
        begin_capture(), called right after the graph that packed them was enqueued.  RCCL orders the collective behind
        everything enqueued on the current stream so far and runs it on its own stream, under whatever is enqueued next."""
        if self._reduce:
            for b in buckets:
                lo, hi = self.bounds[b]
                self._handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group,
                                                     async_op=True))

    def wait_reduced(self):
        """The current stream waits for every all-reduce launched by reduce_buckets(); .grad of every parameter becomes
        its view of the reduced buffer."""
        for h in self._handles:
            h.wait()
        self._handles = []
        for p in self.params:
            p.grad = self._view[id(p)]

    def _on_grad(self, p):
        if not self._relaid and id(p) not in self._arrived:
            self._arrived.add(id(p))
            self._arrival.append(id(p))
        if self.capturing:
            b = self._bucket_of[id(p)]
            self._cap_left[b] -= 1
            if self._cap_left[b] == 0 and not self._cap_done[b]:
                self._cap_complete(b)
            return
        if self.paused:
            return
        b = self._bucket_of[id(p)]
        self._left[b] -= 1
        if self._left[b] == 0 and not self._fired[b]:
            self._launch(b)

    def pack_all(self):
        """Every bucket's gradients -> the flat buffer (a few multi-tensor copies, no collective): the tail of a CAPTURED
        backward pass (Trainer.step_graph); reduce_all() then runs outside the graph."""
        for b in range(len(self.bounds)):
            self._pack(b)

    def reduce_all(self):
        """All-reduce (SUM) of the packed flat buffer, bucket by bucket, and wait; .grad of every parameter becomes its
        view of the reduced buffer.  No autograd involvement: the eager piece between two replayed graphs."""
        if self.flat is None:
            return
        if self._reduce:
            hs = [dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                  for lo, hi in self.bounds]
            for h in hs:
                h.wait()
        for p in self.params:
            p.grad = self._view[id(p)]

    def finish(self):
        """Call after backward(): launches whatever has not fired, waits for all buckets, and points every .grad at
        its (reduced) slice of the flat buffer."""
        if self.flat is None:
            return
        for b in range(len(self.bounds)):
            if not self._fired[b]:
                self._launch(b)
        for h in self._handles:
            h.wait()
        for p in self.params:
            p.grad = self._view[id(p)]
        self._handles, self._left, self._fired = [], list(self._size), [False] * len(self.bounds)


def shard_range(n, rank, world):
    """Contiguous row range [lo, hi) of rank `rank` out of `world` for an n-row database."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


class ShardedFlatL2Index:
    """Row-sharded exact search: every rank holds rows shard_range(n, rank, world) of the database.
    search(q, k): local top-k (ids offset by the shard start) -> all-gather of (nq, k) dist+ids -> merge.
    `local_index_factory(id_base)`, `merge_fn(part_d, part_i)` and `rerank_fn` (signature of ops.seq_rerank) default to
    the HIP implementations; the gloo tests inject the oracle's, as for the loss."""

    def __init__(self, d=128, group=None, local_index_factory=None, merge_fn=None, halo=63, rerank_fn=None):
        self.d, self.group = d, group
        self.rank, self.world = rank_of(group), world_size(group)
        self._factory, self._merge, self._rerank = local_index_factory, merge_fn, rerank_fn
        self.local = None
        self.ntotal = 0
        self.halo = int(halo)          # rows of the NEXT shard kept after the own rows: sequences stay local (rerank)
        self.lo = self.hi = 0
        self._halo_rows = None

    def add_global(self, x):
        """Every rank passes the SAME full (n, d) array (host or memmap); each keeps only its rows (searched) plus
        `halo` rows of the next shard (read by the sequence rerank only).  One call per index."""
        n = len(x)
        lo, hi = shard_range(n, self.rank, self.world)
        if self._factory is None:
            from .ops import FlatL2Index
            self._factory = lambda id_base: FlatL2Index(self.d, id_base=id_base)
        self.local = self._factory(lo + self.ntotal)
        self.local.add(x[lo:hi])
        self.lo, self.hi = lo + self.ntotal, hi + self.ntotal
        self._halo_rows = x[hi:min(n, hi + self.halo)]
        self.ntotal += n

    def add_local(self, rows, lo, n_total, halo_rows=None):
        """Shard-aware loading: this rank passes ONLY the rows it owns, global ids [lo, lo + len(rows)) of an n_total-row
        index (e.g. read from its own slice of a memmap, or generated on the device), plus optionally the first `halo`
        rows of the next shard for the sequence rerank.  Nothing is replicated on the host."""
        if self._factory is None:
            from .ops import FlatL2Index
            self._factory = lambda id_base: FlatL2Index(self.d, id_base=id_base)
        self.local = self._factory(int(lo))
        self.local.add(rows)
        self.lo, self.hi = int(lo), int(lo) + len(rows)
        self._halo_rows = halo_rows
        self.ntotal = int(n_total)

    def rerank(self, q_rows, topk_ids, item_row, item_len, top=10):
        """Sequence-level rerank (eval.py:262-290) over the sharded index: every rank scores the candidates whose start
        row it owns (its rows + halo make those sequences local), then one all-gather of the (n_items, top) lists
        and a merge by (score descending, id ascending).  q_rows / topk_ids are the replicated query segments and the
        merged GLOBAL search results; returns (ids, scores) like ops.seq_rerank."""
        if self._rerank is None:
            from . import ops
            self._rerank = ops.seq_rerank
        dev = self.local.device
        rows = self.local.rows()
        if self._halo_rows is not None and len(self._halo_rows):
            halo = torch.as_tensor(np.ascontiguousarray(self._halo_rows)).to(dev, dtype=torch.float32)
            rows = torch.cat([rows, halo], dim=0)
        as_t = lambda a, dt: torch.as_tensor(a).to(dev, dtype=dt)
        item_len_t = as_t(item_len, torch.int32)
        # host-side arrays (what eval.py passes) give the length bound without a device round trip
        max_len = int(np.max(item_len)) if not torch.is_tensor(item_len) else int(item_len_t.max().item())
        if max_len - 1 > self.halo and self.world > 1:
            raise ValueError(f"sequences of {max_len} segments need halo >= that minus one")
        known = not torch.is_tensor(item_len) and not torch.is_tensor(item_row)
        if known and (int(np.max(np.asarray(item_row) + np.asarray(item_len))) > len(q_rows) or int(np.min(item_row)) < 0):
            raise ValueError("seq_rerank: an item reaches outside q_rows")
        ids, sc = self._rerank(rows, as_t(q_rows, torch.float32), as_t(topk_ids, torch.int64),
                               as_t(item_row, torch.int64), item_len_t, top=top,
                               shard=(self.lo, self.ntotal, self.lo, self.hi), max_len=max_len if known else None)
        if self.world == 1:
            return ids, sc
        n_items = ids.shape[0]
        gi = torch.empty((self.world * n_items, top), dtype=ids.dtype, device=dev)
        gs = torch.empty((self.world * n_items, top), dtype=sc.dtype, device=dev)
        dist.all_gather_into_tensor(gi, ids.contiguous(), group=self.group)
        dist.all_gather_into_tensor(gs, sc.contiguous(), group=self.group)
        gi = gi.reshape(self.world, n_items, top).permute(1, 0, 2).reshape(n_items, -1)
        gs = gs.reshape(self.world, n_items, top).permute(1, 0, 2).reshape(n_items, -1)
        # (score descending, id ascending); empty slots (id -1, score -inf) sink to the end
        key_id = torch.where(gi < 0, torch.full_like(gi, torch.iinfo(torch.int64).max), gi)
        o1 = torch.argsort(key_id, dim=1, stable=True)
        gs1, gi1 = torch.gather(gs, 1, o1), torch.gather(gi, 1, o1)
        o2 = torch.argsort(gs1, dim=1, descending=True, stable=True)
        return torch.gather(gi1, 1, o2)[:, :top].contiguous(), torch.gather(gs1, 1, o2)[:, :top].contiguous()

    def search(self, q, k):
        """numpy in -> numpy out, tensors in -> tensors out (like FlatL2Index.search).  The local search, the
        all-gather and the merge all stay on the local index's device."""
        as_numpy = not torch.is_tensor(q)
        dev = getattr(self.local, "device", None)
        if dev is not None:
            q = torch.as_tensor(q).to(dev, dtype=torch.float32)
        D, I = self.local.search(q, k)
        if self.world > 1:
            D, I = self._gather_merge(torch.as_tensor(D), torch.as_tensor(I))
        if as_numpy and torch.is_tensor(D):
            return D.cpu().numpy(), I.cpu().numpy()
        return D, I

    def _gather_merge(self, D, I):
        nq, kk = D.shape
        gd = torch.empty((self.world * nq, kk), dtype=D.dtype, device=D.device)
        gi = torch.empty((self.world * nq, kk), dtype=I.dtype, device=I.device)
        dist.all_gather_into_tensor(gd, D.contiguous(), group=self.group)
        dist.all_gather_into_tensor(gi, I.contiguous(), group=self.group)
        gd, gi = gd.reshape(self.world, nq, kk), gi.reshape(self.world, nq, kk)
        if self._merge is None:
            from .ops import merge_topk
            self._merge = merge_topk
        return self._merge(gd, gi)
