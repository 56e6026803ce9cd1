"""Fingerprint database writers (mirror of test_fp.py:87-158 `create_fp_db` / `create_dummy_db` and
generate.py:34-57 `create_db`): same signatures and on-disk format ({query,db,dummy_db}.mm float32 memmap +
*_shape.npy; fingerprints.npy), same per-track batching (BatchNorm stays in whatever mode the caller left
the model in -- the reference never calls .eval() in these scripts).  The duplicated
second view of `model(x, x)` is not recomputed.  The memmaps are streamed to disk block by block instead of being
concatenated in host memory first, and with `max_segments > 0` in eval mode the segments of consecutive tracks are
packed into large model calls."""
import os

import numpy as np
import torch


def _device_of(model):
    return next(model.parameters()).device


def _embed(model, x):
    core = model.module if hasattr(model, "module") else model
    with torch.no_grad():
        return core.embed(x)[1]


def _write_memmap(path_noext, arr):
    shape = (len(arr), arr.shape[-1])
    mm = np.memmap(path_noext + ".mm", dtype="float32", mode="w+", shape=shape)
    mm[:] = arr[:]
    mm.flush()
    del mm
    np.save(path_noext + "_shape.npy", shape)


FORCE_STAGING = False      # tests: take the pinned-staging path even where hipHostRegister accepts the file pages


class _MemmapAppender:
    """Streams (n_i, d) float32 blocks into `<name>.mm` (the raw row-major file np.memmap reads, eval.py:151-163) and
    writes `<name>_shape.npy` at the end; no second copy of the whole database exists in host memory.

    Device blocks (`append_device`) take the MI355X-first route -- the reference does a pageable `.cpu()` copy, a
    `tobytes()` copy and a `write()` copy, each synchronous:
      * DIRECT: the file is mapped one window (GROW_ROWS rows) at a time and the window's pages are registered with the
        HIP runtime (hipHostRegister accepts file-backed MAP_SHARED pages on this stack: measured 33 GB/s device -> file
        pages, tools/hostreg_probe.py), so a block is ONE asynchronous DMA from HBM into the page cache of the output
        file on a side stream -- no host copy at all, and the next model call is already running;
      * STAGING (registration refused, or `fpdb.FORCE_STAGING = True`): device -> two alternating pinned staging slots on the
        side stream, each drained with one memcpy into the memmap when it comes round again.
    The file grows sparsely window by window and is truncated to the rows written on close()."""

    GROW_ROWS = 1 << 20            # rows per mapped (and, in direct mode, pinned) window: 512 MB at d = 128

    def __init__(self, path_noext, slot_rows=1 << 15):
        self.path, self.rows, self.dim = path_noext, 0, None
        self.slot_rows = int(slot_rows)
        self._win, self._win_lo, self._win_t, self._registered = None, 0, None, False
        self._direct = not FORCE_STAGING
        self._slots, self._next, self._side = None, 0, None
        open(path_noext + ".mm", "wb").close()

    # ---- file side: one mapped window [win_lo, win_lo + GROW_ROWS) at a time -----------------------------------------
    def _release_window(self):
        if self._win is None:
            return
        if self._side is not None:
            self._side.synchronize()                                       # DMAs into this window have landed
        if self._slots is not None:
            for i in (self._next, self._next ^ 1):
                self._drain(i)
        if self._registered:
            torch.cuda.cudart().cudaHostUnregister(self._win_t.data_ptr())
            self._registered = False
        self._win_t = None
        self._win.flush()
        del self._win
        self._win = None

    def _window_for(self, row, dim, on_device):
        """Maps (and in direct mode pins) the window that holds `row`; returns the rows left in it from `row`."""
        if self.dim is None:
            self.dim = int(dim)
        if dim != self.dim:
            raise ValueError(f"fingerprint width changed from {self.dim} to {dim}")
        G = self.GROW_ROWS
        if self._win is None or not (self._win_lo <= row < self._win_lo + G):
            self._release_window()
            self._win_lo = (row // G) * G
            with open(self.path + ".mm", "r+b") as f:
                f.truncate((self._win_lo + G) * self.dim * 4)              # sparse: no blocks until written
            self._win = np.memmap(self.path + ".mm", dtype="float32", mode="r+", offset=self._win_lo * self.dim * 4,
                                  shape=(G, self.dim))
            self._win_t = torch.from_numpy(self._win)
            if on_device and self._direct:
                rc = torch.cuda.cudart().cudaHostRegister(self._win_t.data_ptr(), self._win_t.numel() * 4, 0)
                self._registered = int(rc) == 0
                self._direct = self._registered                            # refused once: staging from here on
        return self._win_lo + G - row

    def append(self, block):
        """A host block (numpy or CPU tensor)."""
        block = np.ascontiguousarray(block, dtype=np.float32)
        if block.size == 0:
            return
        block = block.reshape(-1, block.shape[-1])
        off = 0
        while off < block.shape[0]:
            m = min(block.shape[0] - off, self._window_for(self.rows, block.shape[-1], False))
            lo = self.rows - self._win_lo
            self._win[lo:lo + m] = block[off:off + m]
            self.rows += m
            off += m

    # ---- device side -----------------------------------------------------------------------------------------------
    def _drain(self, i):
        buf, ev, pending = self._slots[i]
        if pending is not None:
            ev.synchronize()
            lo, n = pending                                                # window-relative
            self._win[lo:lo + n] = buf[:n].numpy()                         # pinned staging -> file pages, one memcpy
            self._slots[i] = (buf, ev, None)

    def append_device(self, block):
        """A (n, d) float32 block resident on a HIP device; returns immediately (the copy runs on a side stream)."""
        if not block.is_cuda:
            return self.append(block.detach().float().numpy())
        block = block.detach().float().contiguous()
        n, d = block.shape
        if n == 0:
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=block.device)
        self._side.wait_stream(torch.cuda.current_stream(block.device))
        block.record_stream(self._side)
        off = 0
        while off < n:
            m = min(n - off, self._window_for(self.rows, d, True))
            lo = self.rows - self._win_lo
            if self._direct:
                with torch.cuda.stream(self._side):
                    self._win_t[lo:lo + m].copy_(block[off:off + m], non_blocking=True)    # HBM -> file pages
            else:
                if self._slots is None:
                    self._slots = [(torch.empty((self.slot_rows, d), dtype=torch.float32, pin_memory=True),
                                    torch.cuda.Event(), None) for _ in range(2)]
                m = min(m, self.slot_rows)
                i = self._next
                self._next ^= 1
                self._drain(i)                                            # the copy issued two blocks ago
                buf, ev, _ = self._slots[i]
                with torch.cuda.stream(self._side):
                    buf[:m].copy_(block[off:off + m], non_blocking=True)
                    ev.record(self._side)
                self._slots[i] = (buf, ev, (lo, m))
            self.rows += m
            off += m

    def close(self):
        self._release_window()
        with open(self.path + ".mm", "r+b") as f:
            f.truncate(self.rows * (self.dim or 0) * 4)
        np.save(self.path + "_shape.npy", (self.rows, self.dim if self.dim is not None else 0))


def part_name(fname, rank, world):
    """File stem of rank `rank`'s slice of a database written by `world` ranks (shard-aware layout): the rows of
    `<fname>` are the concatenation of `<fname>.part0 ... part{world-1}` in rank order; `<fname>_parts.npy` holds the
    row counts.  grafp_amd.eval.load_memmap_data reads either layout; dist.ShardedFlatL2Index loads its own slice."""
    return fname if world <= 1 else f"{fname}.part{rank}"


def write_parts_manifest(output_root_dir, fname, world):
    """After every rank has closed its part: the per-part row counts + the global `<fname>_shape.npy` (rank 0)."""
    shapes = [np.load(os.path.join(output_root_dir, part_name(fname, r, world) + "_shape.npy")) for r in range(world)]
    rows = np.array([int(sh[0]) for sh in shapes], dtype=np.int64)
    dim = max(int(sh[1]) for sh in shapes)
    np.save(os.path.join(output_root_dir, fname + "_parts.npy"), rows)
    np.save(os.path.join(output_root_dir, fname + "_shape.npy"), (int(rows.sum()), dim))


def track_range(n_tracks, rank, world):
    """Contiguous range of tracks rank `rank` fingerprints (so the part files concatenate in track order)."""
    per = (n_tracks + world - 1) // world
    lo = min(n_tracks, rank * per)
    return lo, min(n_tracks, lo + per)


def _embed_stream(dataloader, augment, model, chunks_of, max_segments=0, verbose_every=0, on_device=False):
    """Yields (z (n_seg, d) float32 numpy -- or, with on_device, the device tensor --, track index) in track order.
    Train mode (what the reference's scripts run in: they never call .eval(), so BatchNorm uses the statistics of each
    call): one model call per reference chunk of one track -- the numbers depend on that batching and are reproduced.
    Eval mode with max_segments > 0: a clip's fingerprint does not depend on its batch mathematically, so the segments
    of consecutive tracks are packed into calls of up to `max_segments` clips (fewer, larger launches) and split back
    per track.  Numerically the packed GEMMs round differently and a near-tie k-NN neighbour can flip, which moves a
    fingerprint by up to ~1e-3 relative -- hence opt-in; the default reproduces the per-track calls exactly."""
    dev = _device_of(model)
    training = (model.module if hasattr(model, "module") else model).training or max_segments <= 0
    pend, pend_meta, n_pend = [], [], 0

    def flush():
        nonlocal pend, pend_meta, n_pend
        if not pend:
            return
        z = _embed(model, torch.cat(pend, dim=0)).detach().float()
        z = z if on_device else z.cpu().numpy()
        off = 0
        for idx, n in pend_meta:
            yield z[off:off + n], idx
            off += n
        pend, pend_meta, n_pend = [], [], 0

    for idx, audio in enumerate(dataloader):
        x_i, _ = augment(audio.to(dev), None)
        assert x_i.size(1) == 64 and len(x_i.size()) == 3, f"Shape of x_i: {x_i.shape}"
        if training:
            for part in chunks_of(x_i):
                z = _embed(model, part.to(dev)).detach().float()
                yield (z if on_device else z.cpu().numpy()), idx
        else:
            pend.append(x_i)
            pend_meta.append((idx, x_i.shape[0]))
            n_pend += x_i.shape[0]
            if n_pend >= max_segments:
                yield from flush()
        if verbose_every and idx % verbose_every == 0:
            print(f"Step [{idx}/{len(dataloader)}]\t segments: {x_i.shape[0]}")
    yield from flush()


def create_fp_db(dataloader, augment, model, output_root_dir, verbose=True):
    dev = _device_of(model)
    fp_q, fp_db = _MemmapAppender(os.path.join(output_root_dir, "query")), _MemmapAppender(os.path.join(output_root_dir, "db"))
    print("=> Creating query and db fingerprints...")
    for idx, audio in enumerate(dataloader):
        audio = audio.to(dev)
        x_i, x_j = augment(audio, audio)
        core = model.module if hasattr(model, "module") else model
        with torch.no_grad():
            _, _, z_i, z_j = core(x_i.to(dev), x_j.to(dev))
        fp_db.append_device(z_i)
        fp_q.append_device(z_j)
        if verbose and idx % 10 == 0:
            print(f"Step [{idx}/{len(dataloader)}]\t shape: {z_i.shape}")
    fp_q.close()
    fp_db.close()


def create_dummy_db(dataloader, augment, model, output_root_dir, fname="dummy_db", verbose=True, max_segments=0, rank=0,
                    world=1):
    """test_fp.py:121-158.  world > 1 (shard-aware layout): this rank fingerprints tracks track_range(len, rank, world)
    of an indexable `dataloader` into `<fname>.part<rank>.mm`; once all ranks are done, rank 0 calls
    write_parts_manifest.  The blocks go device -> pinned staging -> memmap pages (_MemmapAppender.append_device)."""
    out = _MemmapAppender(os.path.join(output_root_dir, part_name(fname, rank, world)))
    print("=> Creating dummy fingerprints...")
    if world > 1:
        lo, hi = track_range(len(dataloader), rank, world)
        dataloader = [dataloader[i] for i in range(lo, hi)]
    halves = lambda x: [x] if x.size(0) < 256 else list(torch.chunk(x, 2, dim=0))          # test_fp.py:134-138
    for z, _ in _embed_stream(dataloader, augment, model, halves, max_segments, 100 if verbose else 0, on_device=True):
        out.append_device(z)
    out.close()


def create_db(dataloader, model, augment, output_dir, concat=True, max_size=128, max_segments=0):
    print("Computing fingerprints...")
    blocks = [z for z, _ in _embed_stream(dataloader, augment, model, lambda x: torch.split(x, max_size, dim=0),
                                          max_segments, 10)]                                  # generate.py:41
    if concat:
        fp = np.concatenate(blocks, axis=0) if blocks else np.zeros((0, 128), np.float32)
    else:       # song-level separation kept: one array per model call of the reference (per track in eval mode)
        fp = np.empty(len(blocks), dtype=object)
        for i, v in enumerate(blocks):
            fp[i] = v
    np.save(os.path.join(output_dir, "fingerprints.npy"), fp)
