"""Portable closed-form tensor filler (TEST INFRASTRUCTURE).

Every golden fixture under tests/golden/ is produced from inputs and weights generated here, so the
GPU box can regenerate bit-identical inputs without the reference, without torch's RNG and without
shipping an 82 MB state dict.  value(name, i) = splitmix64(crc32(name) << 32 | i) -> uniform [-1, 1).
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def hash_uniform(name: str, shape) -> np.ndarray:
    """float32 array of `shape`, i.i.d.-looking uniform in [-1, 1), a pure function of (name, index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF)
    idx = np.arange(n, dtype=np.uint64) + (seed << np.uint64(32))
    bits = _splitmix64(idx) >> np.uint64(40)                     # top 24 bits
    u = bits.astype(np.float64) / float(1 << 23) - 1.0           # exact in f32
    return u.astype(np.float32).reshape(shape)


def hash_normalish(name: str, shape) -> np.ndarray:
    """Sum of 4 uniforms, scaled to unit variance: bell-shaped, still closed-form."""
    acc = sum(hash_uniform(f"{name}#{k}", shape).astype(np.float64) for k in range(4))
    return (acc * np.sqrt(3.0 / 4.0)).astype(np.float32)


def hash_ints(name: str, shape, lo: int, hi: int) -> np.ndarray:
    """int32 array uniform in [lo, hi] (inclusive)."""
    u = hash_uniform(name, shape).astype(np.float64)
    return np.clip(np.floor((u + 1.0) * 0.5 * (hi - lo + 1)) + lo, lo, hi).astype(np.int32)


def fill_state_dict(shapes: dict, prefix: str = "w") -> dict:
    """name -> float32/int64 array for every state-dict entry in `shapes` (name -> shape tuple).

    Conv/Linear weights get variance 1/fan_in so activations stay O(1) through 12 blocks; BatchNorm
    affine/running entries get non-trivial values so eval mode is exercised; `relative_pos` (frozen,
    never read by forward: /root/reference/encoder/gcn_lib/torch_vertex.py:190) is left out.
    """
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "relative_pos":
            continue
        if leaf == "num_batches_tracked":
            out[name] = np.zeros((), dtype=np.int64)
            continue
        u = hash_uniform(f"{prefix}:{name}", shape)
        if leaf == "weight" and len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            out[name] = (u * np.float32(np.sqrt(3.0 / fan_in))).astype(np.float32)
        elif leaf == "weight":                    # BatchNorm gamma
            out[name] = (1.0 + 0.1 * u).astype(np.float32)
        elif leaf == "running_var":
            out[name] = (1.0 + 0.5 * np.abs(u)).astype(np.float32)
        else:                                     # bias, BN beta, running_mean
            out[name] = (0.1 * u).astype(np.float32)
    return out
