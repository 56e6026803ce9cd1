"""ctypes bindings for oracle/csrc/*.c.  TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    fl = line.split()
                    return "fma" in fl and "avx2" in fl
    except OSError:
        pass
    return False


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        name = "liboracle_fma.so" if _cpu_has_fma() else "liboracle_generic.so"
        path = os.path.join(_HERE, "_build", name)
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        f32p, i64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)
        L.oracle_knn_graph.argtypes = [f32p] + [ctypes.c_int] * 5 + [i64p, f32p]
        L.oracle_knn_graph.restype = ctypes.c_int
        L.oracle_flat_search_l2.argtypes = [f32p, ctypes.c_int64, f32p, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int64, f32p, i64p]
        L.oracle_flat_search_l2.restype = ctypes.c_int
        L.oracle_merge_topk.argtypes = [f32p, i64p, ctypes.c_int, ctypes.c_int, ctypes.c_int, f32p, i64p]
        L.oracle_merge_topk.restype = None
        L.oracle_row_sqnorm.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p]
        L.oracle_row_sqnorm.restype = None
        i32p = ctypes.POINTER(ctypes.c_int32)
        L.oracle_seq_rerank.argtypes = [f32p, ctypes.c_int64, f32p, i64p, ctypes.c_int, i64p, i32p, ctypes.c_int,
                                        ctypes.c_int, i64p, f32p]
        L.oracle_seq_rerank.restype = ctypes.c_int
        L.oracle_ir_convolve.argtypes = [f32p, ctypes.c_int, ctypes.c_int, f32p, i64p, i32p, i32p, f32p]
        L.oracle_ir_convolve.restype = None
        L.oracle_mix_snr.argtypes = [f32p, ctypes.c_int, ctypes.c_int, f32p, i64p, i32p, i32p, i32p, f32p, f32p]
        L.oracle_mix_snr.restype = None
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.oracle_pq_assign.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, f32p, i32p, f32p, ctypes.c_int, i32p]
        L.oracle_pq_assign.restype = None
        L.oracle_kmeans.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, f32p, i32p, i64p, ctypes.c_int,
                                    ctypes.c_int, f32p]
        L.oracle_kmeans.restype = ctypes.c_int
        L.oracle_ivfpq_probe.argtypes = [f32p, ctypes.c_int, ctypes.c_int, f32p, ctypes.c_int, ctypes.c_int, i32p]
        L.oracle_ivfpq_probe.restype = None
        L.oracle_ivfpq_search.argtypes = [f32p, ctypes.c_int, ctypes.c_int, f32p, f32p, ctypes.c_int, u8p, i64p, i64p,
                                          i32p, ctypes.c_int, ctypes.c_int, f32p, i64p]
        L.oracle_ivfpq_search.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i64(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def knn_graph(x, k, normalize=True, return_dist=False):
    """x (B,C,N) f32 -> idx int64 (B,N,k) with the arithmetic fixed in csrc/knn_graph.c."""
    x, xp = _f32(x)
    B, C, N = x.shape
    idx = np.empty((B, N, k), dtype=np.int64)
    dist = np.empty((B, N, k), dtype=np.float32)
    rc = lib().oracle_knn_graph(xp, B, C, N, k, int(normalize), _i64(idx),
                                dist.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    if rc != 0:
        raise ValueError(f"oracle_knn_graph failed ({rc})")
    return (idx, dist) if return_dist else idx


def flat_search_l2(db, q, k, id_base=0):
    """db (n,d), q (nq,d) -> (dist f32 (nq,k) ascending, ids int64 (nq,k)); csrc/flat_search.c."""
    db, dbp = _f32(db)
    q, qp = _f32(q)
    n, d = db.shape
    nq = q.shape[0]
    od = np.empty((nq, k), dtype=np.float32)
    oi = np.empty((nq, k), dtype=np.int64)
    rc = lib().oracle_flat_search_l2(dbp, n, qp, nq, d, k, id_base,
                                     od.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), _i64(oi))
    if rc != 0:
        raise ValueError(f"oracle_flat_search_l2 failed ({rc})")
    return od, oi


def merge_topk(pd, pi):
    """(P,nq,k) partial lists -> (nq,k)."""
    pd, pdp = _f32(pd)
    pi = np.ascontiguousarray(pi, dtype=np.int64)
    P, nq, k = pd.shape
    od = np.empty((nq, k), dtype=np.float32)
    oi = np.empty((nq, k), dtype=np.int64)
    lib().oracle_merge_topk(pdp, _i64(pi), P, nq, k, od.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), _i64(oi))
    return od, oi


def seq_rerank(index_rows, q_rows, topk_ids, item_row, item_len, top=10):
    """csrc/seq_rerank.c: (out_ids int64 (n_items, top), out_scores f32 (n_items, top))."""
    index_rows, ip = _f32(index_rows)
    q_rows, qp = _f32(q_rows)
    topk_ids = np.ascontiguousarray(topk_ids, dtype=np.int64)
    item_row = np.ascontiguousarray(item_row, dtype=np.int64)
    item_len = np.ascontiguousarray(item_len, dtype=np.int32)
    n_items = len(item_row)
    oi = np.empty((n_items, top), dtype=np.int64)
    os_ = np.empty((n_items, top), dtype=np.float32)
    rc = lib().oracle_seq_rerank(ip, index_rows.shape[0], qp, _i64(topk_ids), topk_ids.shape[1], _i64(item_row),
                                 item_len.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), n_items, top, _i64(oi),
                                 os_.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    if rc != 0:
        raise ValueError(f"oracle_seq_rerank failed ({rc})")
    return oi, os_


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def _ragged(bank, lens, starts):
    bank = np.ascontiguousarray(bank, dtype=np.float32)
    if bank.ndim == 2:                      # padded (n, Lmax) rows -> flat buffer + row starts
        starts = np.arange(bank.shape[0], dtype=np.int64) * bank.shape[1]
        bank = bank.reshape(-1)
    return bank, np.ascontiguousarray(starts, dtype=np.int64), np.ascontiguousarray(lens, dtype=np.int32)


def ir_convolve(x, ir_bank, ir_len, ir_index=None, ir_start=None):
    """x (B,T); impulse responses as a padded (n_ir,Lmax) array or a flat buffer + ir_start; ir_len (n_ir); ir_index (B)
    or None -> (B,T); csrc/augment.c (one fmaf chain per output, taps ascending; full convolution truncated to T)."""
    x, xp = _f32(x)
    bank, st, ln = _ragged(ir_bank, ir_len, ir_start)
    B, T = x.shape
    out = np.empty_like(x)
    ip = None
    if ir_index is not None:
        ix, ip = _i32(ir_index)
    lib().oracle_ir_convolve(xp, B, T, bank.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), _i64(st),
                             ln.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ip,
                             out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def mix_snr(x, noise_bank, noise_len, noise_index, noise_offset, snr_db, noise_start=None):
    """AddBackgroundNoise restated (csrc/augment.c): x + rms(x)/10^(snr/20) * n/(rms(n)+1e-8), n read circularly."""
    x, xp = _f32(x)
    bank, st, ln = _ragged(noise_bank, noise_len, noise_start)
    ni, nip = _i32(noise_index)
    no, nop = _i32(noise_offset)
    sn, sp = _f32(snr_db)
    B, T = x.shape
    out = np.empty_like(x)
    lib().oracle_mix_snr(xp, B, T, bank.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), _i64(st),
                         ln.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), nip, nop, sp,
                         out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


# ---- IVF-PQ (csrc/ivfpq.c) ---------------------------------------------------------------------------------------------
def _pi32(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def _opt_base(base, base_idx):
    if base is None:
        return None, None, None, None
    b, bp = _f32(base)
    bi = np.ascontiguousarray(base_idx, dtype=np.int32)
    return b, bp, bi, _pi32(bi)


def pq_assign(x, G, cent, base=None, base_idx=None):
    """x (n, D), cent (G, k, D/G) -> (n, G) int32: nearest codeword of every (row, sub-space) of the residuals."""
    x, xp = _f32(x)
    c, cp = _f32(cent)
    n, D = x.shape
    k = c.shape[-2]
    _b, bp, _bi, bip = _opt_base(base, base_idx)
    out = np.empty((n, G), dtype=np.int32)
    lib().oracle_pq_assign(xp, n, D, G, bp, bip, cp, k, _pi32(out))
    return out


def kmeans(x, G, k, init_rows, niter, base=None, base_idx=None):
    """Seeded Lloyd iterations with the summation order fixed in csrc/ivfpq.c -> centroids (G, k, D/G) f32."""
    x, xp = _f32(x)
    n, D = x.shape
    init = np.ascontiguousarray(init_rows, dtype=np.int64)
    _b, bp, _bi, bip = _opt_base(base, base_idx)
    cent = np.empty((G, k, D // G), dtype=np.float32)
    rc = lib().oracle_kmeans(xp, n, D, G, bp, bip, _i64(init), k, niter,
                             cent.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    if rc != 0:
        raise MemoryError("oracle_kmeans")
    return cent


def ivfpq_probe(q, cent, nprobe):
    q, qp = _f32(q)
    c, cp = _f32(cent)
    probe = np.empty((q.shape[0], nprobe), dtype=np.int32)
    lib().oracle_ivfpq_probe(qp, q.shape[0], q.shape[1], cp, c.shape[0], nprobe, _pi32(probe))
    return probe


def ivfpq_search(q, cent, books, codes, list_start, ids, probe, k):
    """codes (n, M) uint8 and ids (n) int64 in list order -> (D (nq, k) f32, I (nq, k) int64) by (estimate, id)."""
    q, qp = _f32(q)
    c, cp = _f32(cent)
    b, bp = _f32(books)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    ls = np.ascontiguousarray(list_start, dtype=np.int64)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    probe = np.ascontiguousarray(probe, dtype=np.int32)
    D = np.empty((q.shape[0], k), dtype=np.float32)
    I = np.empty((q.shape[0], k), dtype=np.int64)
    lib().oracle_ivfpq_search(qp, q.shape[0], q.shape[1], cp, bp, b.shape[0],
                              codes.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), _i64(ls), _i64(ids), _pi32(probe),
                              probe.shape[1], k, D.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), _i64(I))
    return D, I
