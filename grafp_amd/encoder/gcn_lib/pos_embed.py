"""2-D sin-cos relative position table.

Only needed so that every Grapher owns a frozen `relative_pos` parameter of the reference's shape (the
state-dict schema); forward never reads it (/root/reference/encoder/gcn_lib/torch_vertex.py:190 passes
relative_pos=None).  Public construction: MoCo-v3 style sin-cos embedding e(p) of each grid cell,
table = 2 e e^T / dim.
"""
import numpy as np


def _axis_embed(dim, coords):
    freq = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
    phase = np.outer(coords.reshape(-1), freq)
    return np.concatenate([np.sin(phase), np.cos(phase)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    assert embed_dim % 4 == 0
    ww, hh = np.meshgrid(np.arange(grid_size, dtype=np.float32), np.arange(grid_size, dtype=np.float32))
    emb = np.concatenate([_axis_embed(embed_dim // 2, ww), _axis_embed(embed_dim // 2, hh)], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


def get_2d_relative_pos_embed(embed_dim, grid_size):
    e = get_2d_sincos_pos_embed(embed_dim, grid_size)
    return 2 * np.matmul(e, e.transpose()) / e.shape[1]
