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


class _MemmapAppender:
    """Streams (n_i, d) float32 blocks into `<name>.mm` (the raw row-major file np.memmap reads, eval.py:151-163)
    and writes `<name>_shape.npy` at the end: no second copy of the whole database in host memory."""

    def __init__(self, path_noext):
        self.path, self.rows, self.dim = path_noext, 0, None
        self.f = open(path_noext + ".mm", "wb")

    def append(self, block):
        block = np.ascontiguousarray(block, dtype=np.float32)
        if block.size == 0:
            return
        self.dim = block.shape[-1]
        self.f.write(block.tobytes())
        self.rows += block.shape[0]

    def close(self):
        self.f.close()
        np.save(self.path + "_shape.npy", (self.rows, self.dim if self.dim is not None else 0))


def _embed_stream(dataloader, augment, model, chunks_of, max_segments=0, verbose_every=0):
    """Yields (z (n_seg, d) float32 numpy, track index) in track order.
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
        z = _embed(model, torch.cat(pend, dim=0)).detach().float().cpu().numpy()
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
                yield _embed(model, part.to(dev)).detach().float().cpu().numpy(), idx
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
        fp_db.append(z_i.detach().float().cpu().numpy())
        fp_q.append(z_j.detach().float().cpu().numpy())
        if verbose and idx % 10 == 0:
            print(f"Step [{idx}/{len(dataloader)}]\t shape: {z_i.shape}")
    fp_q.close()
    fp_db.close()


def create_dummy_db(dataloader, augment, model, output_root_dir, fname="dummy_db", verbose=True, max_segments=0):
    out = _MemmapAppender(os.path.join(output_root_dir, fname))
    print("=> Creating dummy fingerprints...")
    halves = lambda x: [x] if x.size(0) < 256 else list(torch.chunk(x, 2, dim=0))          # test_fp.py:134-138
    for z, _ in _embed_stream(dataloader, augment, model, halves, max_segments, 100 if verbose else 0):
        out.append(z)
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
