"""Fingerprint database writers (mirror of test_fp.py:87-158 `create_fp_db` / `create_dummy_db` and
generate.py:34-57 `create_db`): same signatures and on-disk format ({query,db,dummy_db}.mm float32 memmap +
*_shape.npy; fingerprints.npy), same per-track batching (BatchNorm stays in whatever mode the caller left
the model in -- the reference never calls .eval() in these scripts).  The duplicated
second view of `model(x, x)` is not recomputed."""
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


def create_fp_db(dataloader, augment, model, output_root_dir, verbose=True):
    dev = _device_of(model)
    fp_q, fp_db = [], []
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
    _write_memmap(os.path.join(output_root_dir, "query"), np.concatenate(fp_q))
    _write_memmap(os.path.join(output_root_dir, "db"), np.concatenate(fp_db))


def create_dummy_db(dataloader, augment, model, output_root_dir, fname="dummy_db", verbose=True):
    dev = _device_of(model)
    fp = []
    print("=> Creating dummy fingerprints...")
    for idx, audio in enumerate(dataloader):
        x_i, _ = augment(audio.to(dev), None)
        assert x_i.size(1) == 64 and len(x_i.size()) == 3, f"Shape of x_i: {x_i.shape}"
        parts = [x_i] if x_i.size(0) < 256 else list(torch.chunk(x_i, 2, dim=0))   # test_fp.py:134-138
        for part in parts:
            z = _embed(model, part.to(dev))
            fp.append(z.detach().float().cpu().numpy())
        if verbose and idx % 100 == 0:
            print(f"Step [{idx}/{len(dataloader)}]\t shape: {z.shape}")
    _write_memmap(os.path.join(output_root_dir, fname), np.concatenate(fp))


def create_db(dataloader, model, augment, output_dir, concat=True, max_size=128):
    dev = _device_of(model)
    fp = []
    print("Computing fingerprints...")
    for idx, audio in enumerate(dataloader):
        x_i, _ = augment(audio.to(dev), None)
        for part in torch.split(x_i, max_size, dim=0):                                 # generate.py:41
            z = _embed(model, part.to(dev))
            fp.append(z.detach().float().cpu().numpy())
        if idx % 10 == 0:
            print(f"Step [{idx}/{len(dataloader)}]\t shape: {z.shape}")
    fp = np.concatenate(fp, axis=0) if concat else np.array(fp, dtype=object)
    np.save(os.path.join(output_dir, "fingerprints.npy"), fp)
