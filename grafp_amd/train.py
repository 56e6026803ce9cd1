"""Training-step harness (counterpart of train.py:56-82, 164-177 for synthetic or pre-loaded waveforms).

Step semantics reproduced: zero grads -> (no_grad) log-mel of both views -> SimCLR forward, the two views
sequentially -> NT-Xent on the whole (global) batch -> backward -> Adam step; fp32 parameters, optional bf16
autocast for the GEMMs (the graph build, gather and loss kernels always run in f32).  The reference's
per-step `loss.item()` host sync (train.py:80) is NOT reproduced: the loss stays on the device.
"""
import contextlib

import torch

from . import dist as gdist
from .encoder.graph_encoder import GraphEncoder
from .modules.transformations import GPUTransformNeuralfp
from .simclr.ntxent import ntxent_loss
from .simclr.simclr import SimCLR


def build_model(cfg, k=3, device=None):
    model = SimCLR(cfg, encoder=GraphEncoder(cfg=cfg, in_channels=cfg["n_filters"], k=k))   # == train.py:164
    return model.to(device) if device is not None else model


class Trainer:
    def __init__(self, cfg, model, device, amp_dtype=None, group=None, lr=None, n_buckets=4, ir_dir=None,
                 noise_dir=None, aug_seed=None, data_parallel_graphs=None, overlap_graph_allreduce=True):
        self.cfg, self.model, self.device, self.group = cfg, model, device, group
        self.amp_dtype = amp_dtype
        self.world = gdist.world_size(group)
        # step_graph's data-parallel form (three graphs, eager collectives): by default whenever there is more than one
        # rank; True forces it for a one-rank process group too (the tests run RCCL that way on a one-GPU box)
        self._dp_graphs = self.world > 1 if data_parallel_graphs is None else bool(data_parallel_graphs)
        # step_graph under data parallelism: True (default) = backward is recorded as one graph per gradient bucket and
        # bucket b's all-reduce is launched between graph b and graph b + 1 (it runs under the rest of backward);
        # False = one backward graph, all buckets reduced behind it
        self._overlap_graph_allreduce = bool(overlap_graph_allreduce)
        # ir_dir / noise_dir: recordings for the batched device-side augmentation of the second view (train.py:150-151)
        self.augment = GPUTransformNeuralfp(dict(cfg, aug_seed=aug_seed), ir_dir, noise_dir, train=True).to(device)
        # train.py:174 (same Adam, defaults); on the GPU the update of all 271 parameter tensors is one fused launch
        # with device-side step counters instead of ~35 multi-tensor launches and 271 host-side counter bumps
        on_gpu = torch.device(device).type == "cuda"
        # capturable: the step counters live on the device, so the whole step can be recorded into a HIP graph -- and
        # so does the learning rate (a 0-d device tensor the scheduler updates in place): a captured Adam reads it at
        # REPLAY time, so CosineAnnealingLR (train.py:175,224: scheduler.step() once per epoch) keeps working under
        # step_graph instead of being frozen at its value at capture time
        lr0 = lr or cfg["lr"]
        # the tensor belongs to the Trainer, not to the optimizer's dict: load_state_dict REPLACES param_group['lr'] with
        # whatever the checkpoint holds (a float from a reference-format file), see load_checkpoint()
        self._lr = torch.tensor(float(lr0), dtype=torch.float32, device=device) if on_gpu else None
        if on_gpu:
            # round 6: the update itself is the hand-written multi-tensor kernel (csrc/adam.hip: 5 launches at the rate of
            # a streaming kernel instead of torch's 12 at 1.6 TB/s -- 0.2 ms of EVERY step, 1.4 % at 128 pairs per GPU);
            # the object is a torch.optim.Adam in every other respect (state keys, state_dict, scheduler)
            from .optim import Adam
            self.opt = Adam(model.parameters(), lr=self._lr)
        else:
            self.opt = torch.optim.Adam(model.parameters(), lr=lr0)
        self._graph = None                     # (hipGraph, static x_i, static x_j, static loss) once captured
        # modules whose forward depends on .training (BatchNorm statistics, dropout): the short list step() looks at
        self._mode_sensitive = [m for m in model.modules()
                                if isinstance(m, (torch.nn.modules.batchnorm._BatchNorm, torch.nn.modules.dropout._DropoutNd))]
        self.sched = torch.optim.lr_scheduler.CosineAnnealingLR(self.opt, T_max=cfg["T_max"], eta_min=cfg["min_lr"])
        self.sync = gdist.GradSync(model.parameters(), group=group, n_buckets=n_buckets,
                                   force_flat=self._dp_graphs and self.world == 1)

    def _autocast(self):
        if self.amp_dtype is None:
            return contextlib.nullcontext()
        return torch.autocast(device_type="cuda", dtype=self.amp_dtype)

    def step(self, x_i, x_j):
        """x_i, x_j: (B_local, T) waveforms already on the device.  Returns this rank's share of the loss
        (a 0-d device tensor; the shares sum to the global mean loss)."""
        self._rebind_lr()                      # a util.load_ckp(optimizer=trainer.opt) in between swapped the lr object
        # (Module.train() walks and re-assigns ~340 modules: 1.7 ms of host time per call -- only when something IS in eval
        #  mode: the root, or one of the mode-sensitive modules under it, e.g. after model.encoder.eval() for fingerprinting)
        if not self.model.training or not all(m.training for m in self._mode_sensitive):
            self.model.train()
        self.sync.zero()
        with torch.no_grad():
            X_i, X_j = self.augment(x_i, x_j)
        with self._autocast():
            _, _, z_i, z_j = self.model(X_i, X_j)
        if self.world > 1:
            loss = gdist.ntxent_global(z_i, z_j, self.cfg["tau"], self.group)
        else:
            loss = ntxent_loss(z_i, z_j, self.cfg)
        from . import ops
        with ops.defer_wgrad_reduce():            # the 64 weight gradients' partial sums: reduced together, not one by one
            loss.backward()
        self.sync.finish()
        self.opt.step()
        return loss.detach()

    def _snapshot(self):
        with torch.no_grad():
            keep = [t.detach().clone() for t in list(self.model.parameters()) + list(self.model.buffers())]
            opt_keep = {p: {k: v.detach().clone() for k, v in st.items() if torch.is_tensor(v)}
                        for p, st in self.opt.state.items()}
        return keep, opt_keep

    def _restore(self, keep, opt_keep):
        with torch.no_grad():
            for t, v in zip(list(self.model.parameters()) + list(self.model.buffers()), keep):
                t.copy_(v)
            for p, st in self.opt.state.items():     # Adam moments and step counters: as before the warm-up
                for k, v in st.items():
                    if torch.is_tensor(v):
                        v.copy_(opt_keep[p][k]) if p in opt_keep else v.zero_()

    def step_graph(self, x_i, x_j):
        """Same step, replayed from HIP graphs.  The first call runs three eager steps on a side stream (allocator and
        library warm-up, as torch.cuda.graphs asks), records the fourth and replays it; later calls copy the batch
        into the static input buffers and replay.  Shapes must not change between calls.
          * one process: ONE graph (augment, forward, loss, backward, Adam), ~700 launches as one graph launch;
          * data parallel: graphs with the collectives between them, eager, exactly where step() has them --
            [augment + forward] -> all-gather of (z_i, z_j) -> [global-negative loss + backward up to the point where
            gradient bucket 0 is complete and packed] -> all-reduce of bucket 0 (launched, not waited for) -> [backward up
            to bucket 1] -> ... -> [Adam].  The collectives are not captured (RCCL kernels inside a graph are untested
            on this stack): backward is cut into one graph per bucket (GradSync.begin_capture), so every bucket's
            all-reduce runs under the backward graphs that follow it -- the overlap of the eager step, with the ~700
            launches of a step folded into 2 + (number of buckets) graph launches."""
        self._rebind_lr()
        if self._graph is None:
            if getattr(self.augment, "seed", None) is not None:
                raise RuntimeError("step_graph: a private augmentation generator (aug_seed) is eager-only -- its draws "
                                   "would be frozen into the graph; use the default device generator")
            sx_i, sx_j = x_i.clone(), x_j.clone()
            # the warm-up steps (allocator / library warm-up, as torch.cuda.graphs asks) and the capture itself must
            # not count as training: weights, BatchNorm statistics and optimizer state are put back afterwards, so the
            # first call is ONE step like every later one
            keep, opt_keep = self._snapshot()
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        self.step(sx_i, sx_j)
                torch.cuda.current_stream().wait_stream(side)
                if not self._dp_graphs:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        loss = self.step(sx_i, sx_j)
                    captured = ("single", graph, sx_i, sx_j, loss)
                else:
                    captured = self._capture_data_parallel(sx_i, sx_j)
            finally:
                # also when the warm-up or the capture raised: the caller's model is as it was before the call
                self._restore(keep, opt_keep)
            self._graph = captured
            return self._replay()                # capture only records: this runs the step on the batch
        sx_i, sx_j = self._graph[2], self._graph[3]
        if sx_i.shape != x_i.shape or sx_j.shape != x_j.shape:
            raise ValueError("step_graph: batch shape changed since capture")
        sx_i.copy_(x_i)
        sx_j.copy_(x_j)
        return self._replay()

    def _capture_data_parallel(self, sx_i, sx_j):
        import gc

        import torch.distributed as tdist
        from . import ops
        R, rank = self.world, gdist.rank_of(self.group)
        self.model.train()
        self.sync.zero()
        self.sync.paused = True                      # no collective from the autograd hooks while the graphs are recorded
        cap = torch.cuda.Stream()                    # ONE capture stream for all graphs: backward runs where forward ran
        mode = "thread_local"                        # the process group's watchdog thread may query events meanwhile
        try:
            g_fwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_fwd, stream=cap, capture_error_mode=mode):
                with torch.no_grad():
                    X_i, X_j = self.augment(sx_i, sx_j)
                with self._autocast():
                    _, _, z_i, z_j = self.model(X_i, X_j)
                mine = torch.stack((z_i.detach().float(), z_j.detach().float()), dim=0).contiguous()   # (2, B_loc, D)
            gathered = torch.empty((R * 2,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
            tdist.all_gather_into_tensor(gathered, mine, group=self.group)
            # Backward as ONE GRAPH PER GRADIENT BUCKET: when a bucket's last gradient arrives its hook packs the bucket and
            # ends the graph being recorded; the next one begins on the same stream and memory pool.  Autograd runs on THIS
            # thread meanwhile (set_multithreading_enabled(False)): a capture must end on the thread that began it.
            parts = []
            pool = g_fwd.pool()
            torch.cuda.synchronize()
            gc.collect()
            with torch.cuda.stream(cap):
                cur = [torch.cuda.CUDAGraph()]
                cur[0].capture_begin(pool=pool, capture_error_mode=mode)
                try:
                    def cut(_b):
                        cur[0].capture_end()
                        parts.append(cur[0])
                        cur[0] = None
                        nxt = torch.cuda.CUDAGraph()
                        nxt.capture_begin(pool=pool, capture_error_mode=mode)
                        cur[0] = nxt
                    both = gathered.reshape(R, 2, *mine.shape[1:]).permute(1, 0, 2, 3)
                    zi_all = both[0].reshape(-1, mine.shape[2]).contiguous()
                    zj_all = both[1].reshape(-1, mine.shape[2]).contiguous()
                    loss = ops.ntxent(z_i, z_j, self.cfg["tau"], zi_all, zj_all, rank * z_i.shape[0])
                    self.sync.begin_capture(cut if self._overlap_graph_allreduce else None)
                    with torch.autograd.set_multithreading_enabled(False), ops.defer_wgrad_reduce():
                        loss.backward()
                    ready = self.sync.end_capture()       # per graph: the buckets complete once it has run
                    loss = loss.detach()
                finally:
                    if cur[0] is not None:
                        cur[0].capture_end()
                        parts.append(cur[0])
            assert len(ready) == len(parts), (len(ready), len(parts))
            self.sync.reduce_all()
            g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_opt, pool=pool, stream=cap, capture_error_mode=mode):
                self.opt.step()
        finally:
            self.sync.paused = False
            self.sync.capturing = False
        return ("dp", (g_fwd, parts, g_opt), sx_i, sx_j, loss, mine, gathered, ready)

    def _replay(self):
        if self._graph[0] == "single":
            self._graph[1].replay()
            return self._graph[4].clone()
        import torch.distributed as tdist
        (g_fwd, parts, g_opt), loss, mine, gathered = self._graph[1], self._graph[4], self._graph[5], self._graph[6]
        ready = self._graph[7]
        # RCCL collectives are stream-ordered behind the replayed graph.  A host-staged backend (gloo: the two-ranks-on-
        # one-GPU tests) synchronises with the stream from its own thread, and doing that while a ~700-node graph
        # launch is still being enqueued costs SECONDS per step on this stack (measured: 1-11 s against 47 ms with the
        # explicit synchronize below) -- so for those backends the stream is drained first.
        host_staged = tdist.get_backend(self.group) != "nccl"
        g_fwd.replay()
        if host_staged:
            torch.cuda.current_stream().synchronize()
        tdist.all_gather_into_tensor(gathered, mine, group=self.group)
        if host_staged:
            for g in parts:
                g.replay()
            torch.cuda.current_stream().synchronize()
            self.sync.reduce_all()
        else:
            # bucket b's all-reduce is enqueued between backward graph b and b + 1: RCCL orders it behind graph b and runs
            # it on its own stream under the graphs that follow -- the overlap of the eager step, without a captured
            # collective and without device-side polling
            for g, buckets in zip(parts, ready):
                g.replay()
                self.sync.reduce_buckets(buckets)
            self.sync.wait_reduced()
        g_opt.replay()
        return loss.clone()

    def checkpoint(self, epoch, loss_log, hit_rate_log, hit_rates=None):
        """The dict layout of train.py:212-220; the learning rate is stored as the float the reference's files hold (not
        the device tensor the captured Adam reads)."""
        opt_sd = self.opt.state_dict()
        opt_sd["param_groups"] = [dict(g, lr=float(g["lr"]), **({"initial_lr": float(g["initial_lr"])}
                                                               if "initial_lr" in g else {}))
                                  for g in opt_sd["param_groups"]]
        return {"epoch": epoch, "loss": loss_log, "valid_acc": hit_rate_log, "hit_rate": hit_rates,
                "state_dict": self.model.state_dict(), "optimizer": opt_sd,
                "scheduler": self.sched.state_dict()}

    def _rebind_lr(self):
        """After anything that may have replaced param_group['lr'] (optimizer.load_state_dict does): the loaded VALUE goes
        into the Trainer's own device tensor -- the one a captured Adam graph reads at replay time and the scheduler
        updates in place -- and the tensor goes back into the group."""
        if self._lr is None:
            return
        for g in self.opt.param_groups:
            if g["lr"] is not self._lr:
                self._lr.fill_(float(g["lr"]))
                g["lr"] = self._lr

    def load_checkpoint(self, ckp):
        """Resume from a checkpoint dict (train.py:212-220 layout; `ckp` may also be a path): model weights, Adam state,
        scheduler.  Works before and after step_graph() has captured."""
        if not isinstance(ckp, dict):
            ckp = torch.load(ckp, map_location=self.device, weights_only=False)
        from .util import strip_module_prefix
        self.model.load_state_dict(strip_module_prefix(ckp["state_dict"]))
        if self._graph is not None:
            # a captured Adam updates the state tensors it was recorded with: copy INTO them instead of replacing them
            loaded = torch.optim.Adam(self.model.parameters(), lr=float(self._lr), fused=True, capturable=True)
            loaded.load_state_dict(ckp["optimizer"])
            with torch.no_grad():
                for p, st in self.opt.state.items():
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            v.copy_(loaded.state[p][k])
            for g, gl in zip(self.opt.param_groups, loaded.param_groups):
                self._lr.fill_(float(gl["lr"]))
                if "initial_lr" in gl:
                    g["initial_lr"] = float(gl["initial_lr"])
        else:
            self.opt.load_state_dict(ckp["optimizer"])
        self._rebind_lr()
        if ckp.get("scheduler") is not None:
            self.sched.load_state_dict(ckp["scheduler"])
        return ckp.get("epoch"), ckp.get("loss"), ckp.get("valid_acc")


def synthetic_batch(batch, seed, device, n_samples=16000):
    """x_i = 0.1*randn, x_j = x_i + 0.03*randn (CPU generator, then copied)."""
    gen = torch.Generator().manual_seed(seed)
    x_i = 0.1 * torch.randn(batch, n_samples, generator=gen)
    x_j = x_i + 0.03 * torch.randn(batch, n_samples, generator=gen)
    return x_i.to(device), x_j.to(device)
