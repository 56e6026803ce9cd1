"""IVF-PQ index: the index of the published evaluation protocol, /root/reference/eval.py:65-69
`faiss.IndexIVFPQ(IndexFlatL2(d), d, n_centroids=64, code_sz=64, nbits=8)`, `nprobe = 20` (:122) -- the default
`index_type` of test_fp.py:276.  faiss==1.7.2 is not vendored and not installable; its published algorithm is restated
(inverted file over a k-means coarse quantiser, product quantisation of the RESIDUALS with M sub-quantisers of 2^nbits
codewords, asymmetric distance computation at search time) and the two k-means are pinned to a seeded Lloyd iteration
of our own, so results are reproducible but only STATISTICALLY comparable with faiss (its k-means initialisation is not
reproducible without faiss).  Training (both k-means), encoding, the coarse probe and the search (scan of the
probed lists with the top-k fused in) are hand-written kernels of csrc/ivfpq.hip; only the list bookkeeping of `add`
(stable sort of the rows by list, list offsets) uses torch.

Same surface as ops.FlatL2Index / the subset of faiss eval.py uses: d, ntotal, nprobe, train(x), add(x), search(q, k),
plus rows() (the raw vectors, which the sequence rerank reads: the reference keeps them in a memmap, eval.py:214-232).
Exact search remains the faster and more accurate index on this GPU; this one exists so that the
published protocol can be followed to the letter.
"""
import ctypes

import numpy as np
import torch

from ._lib import check, lib

_vp = ctypes.c_void_p


def _stream():
    return _vp(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _ptr(t):
    return _vp(t.data_ptr()) if t is not None else None


def pq_assign(x, G, cent, base=None, base_idx=None, as_codes=False):
    """Nearest centroid of every (row, sub-space) (csrc/ivfpq.hip, grafp_pq_assign_f32).  x (n, D) f32 on the device,
    cent (G, k, D/G); base / base_idx: residuals x[row] - base[base_idx[row]] are quantised instead of the rows.
    -> (n, G) int32, or uint8 codes with as_codes (k <= 256)."""
    n, D = x.shape
    k = cent.shape[-2]
    out = torch.empty((n, G), dtype=torch.uint8 if as_codes else torch.int32, device=x.device)
    check(lib.grafp_pq_assign_f32(_ptr(x), n, D, G, _ptr(base), _ptr(base_idx), _ptr(cent), k,
                                  None if as_codes else _ptr(out), _ptr(out) if as_codes else None, _stream()),
          "pq_assign")
    return out


def kmeans_init_rows(n, k, seed):
    """k distinct training rows (seeded permutation; repeated when there are fewer rows than centroids)."""
    gen = torch.Generator().manual_seed(int(seed))
    perm = torch.randperm(n, generator=gen)[:k]
    if perm.numel() < k:
        perm = perm.repeat((k + perm.numel() - 1) // perm.numel())[:k]
    return perm.to(torch.int64)


def kmeans(x, G, k, niter=25, seed=1234, base=None, base_idx=None):
    """Seeded Lloyd k-means wholly on the device (csrc/ivfpq.hip, grafp_kmeans_f32): x (n, D) f32 split into G
    sub-spaces -> centroids (G, k, D/G).  Initial centroids = k distinct training rows (seeded permutation, the same for
    every sub-space); cluster sums in a fixed order (reproducible bit for bit, restated by oracle/csrc/ivfpq.c); an
    empty cluster keeps its previous centroid."""
    n, D = x.shape
    init = kmeans_init_rows(n, k, seed).to(x.device)
    cent = torch.empty((G, k, D // G), dtype=torch.float32, device=x.device)
    ws_bytes = int(lib.grafp_kmeans_workspace(n, D, G, k))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    check(lib.grafp_kmeans_f32(_ptr(x), n, D, G, _ptr(base), _ptr(base_idx), _ptr(init), k, int(niter), _ptr(cent),
                               _ptr(ws), ws_bytes, _stream()), "kmeans")
    return cent


class IVFPQIndex:
    def __init__(self, d=128, nlist=64, M=64, nbits=8, device=None, seed=1234, niter=25, max_points_per_centroid=256):
        if nbits != 8:
            raise NotImplementedError("IVFPQIndex: 8-bit codes (the reference's setting)")
        if d % M:
            raise ValueError("IVFPQIndex: d must be a multiple of M")
        if not torch.cuda.is_available():
            raise RuntimeError("IVFPQIndex needs a HIP device: there is no CPU fallback")
        self.d, self.nlist, self.M, self.dsub = int(d), int(nlist), int(M), d // M
        self.device = torch.device(device if device is not None else "cuda")
        self.seed, self.niter, self.max_ppc = seed, niter, max_points_per_centroid
        self.nprobe = 1                                   # faiss default; eval.get_index sets 20
        self.is_trained = False
        self.centroids = self.codebooks = None            # (nlist, d), (M, 256, dsub)
        self._raw, self._codes, self._assign = [], [], []
        self._sorted = None
        self.ntotal = 0

    # ---- training: coarse k-means, then one k-means per sub-space on the residuals (by_residual = True) ---------------
    def train(self, x):
        # faiss subsamples its training set to max_points_per_centroid * max(nlist, 256) rows; the selection happens on
        # the HOST side (array or memmap) so that only those rows ever reach the device
        cap = self.max_ppc * max(self.nlist, 256)
        n_rows = (x.numel() if torch.is_tensor(x) else int(np.prod(x.shape))) // self.d
        if n_rows > cap:
            gen = torch.Generator().manual_seed(self.seed + 1)
            sel = torch.sort(torch.randperm(n_rows, generator=gen)[:cap]).values
            x = x.reshape(-1, self.d)[sel.to(x.device)] if torch.is_tensor(x) else np.asarray(x).reshape(-1, self.d)[sel.numpy()]
        x = torch.as_tensor(np.ascontiguousarray(x) if isinstance(x, np.ndarray) else x).to(self.device, torch.float32)
        x = x.reshape(-1, self.d)
        x = x.contiguous()
        self.centroids = kmeans(x, 1, self.nlist, self.niter, self.seed)[0].contiguous()
        a = pq_assign(x, 1, self.centroids[None]).reshape(-1)
        self.codebooks = kmeans(x, self.M, 256, self.niter, self.seed + 2, base=self.centroids, base_idx=a).contiguous()
        self.is_trained = True

    def encode(self, x):
        """x (n, d) on the device -> (list id (n) int64, codes (n, M) uint8)."""
        x = x.contiguous()
        a = pq_assign(x, 1, self.centroids[None]).reshape(-1)
        codes = pq_assign(x, self.M, self.codebooks, base=self.centroids, base_idx=a, as_codes=True)
        return a.to(torch.int64), codes

    def add(self, x, chunk=1 << 16):
        if not self.is_trained:
            raise RuntimeError("IVFPQIndex.add before train")
        x = torch.as_tensor(np.ascontiguousarray(x) if isinstance(x, np.ndarray) else x).reshape(-1, self.d)
        for lo in range(0, x.shape[0], chunk):
            xb = x[lo:lo + chunk].to(self.device, torch.float32)
            a, codes = self.encode(xb)
            self._raw.append(xb)
            self._assign.append(a)
            self._codes.append(codes)
        self.ntotal += x.shape[0]
        self._sorted = None

    def _materialise(self):
        if self._sorted is None:
            raw = torch.cat(self._raw) if len(self._raw) != 1 else self._raw[0]
            a, codes = torch.cat(self._assign), torch.cat(self._codes)
            order = torch.argsort(a, stable=True)                                     # insertion order inside a list
            counts = torch.bincount(a, minlength=self.nlist)
            start = torch.zeros(self.nlist + 1, dtype=torch.int64, device=self.device)
            start[1:] = torch.cumsum(counts, 0)
            self._raw, self._assign, self._codes = [raw], [a], [codes]
            self._sorted = (codes[order].contiguous(), order.contiguous(), start.contiguous(), counts)
        return self._sorted

    def rows(self):
        self._materialise()
        return self._raw[0]

    # ---- search: nprobe nearest lists (grafp_ivfpq_probe_f32), then ONE launch that scans their codes with the running
    # top-k fused in (grafp_ivfpq_search_f32): no (query x probed codes) scratch, no host round trip ------------------------
    def search(self, q, k):
        as_numpy = isinstance(q, np.ndarray)
        qt = torch.as_tensor(np.ascontiguousarray(q) if as_numpy else q).to(self.device, torch.float32).reshape(-1, self.d)
        qt = qt.contiguous()
        nq = qt.shape[0]
        k = int(k)
        if k < 1:
            raise ValueError("IVFPQIndex.search: k >= 1")
        D = torch.full((nq, k), float("inf"), device=self.device)
        I = torch.full((nq, k), -1, dtype=torch.int64, device=self.device)
        if self.ntotal and nq:
            codes, ids, start, counts = self._materialise()
            nprobe = max(1, min(int(self.nprobe), self.nlist))
            probe = torch.empty((nq, nprobe), dtype=torch.int32, device=self.device)
            stream = _stream()
            check(lib.grafp_ivfpq_probe_f32(_ptr(qt), nq, self.d, _ptr(self.centroids), self.nlist, nprobe, _ptr(probe),
                                            stream), "ivfpq_probe")
            if k <= 32:                                   # GRAFP_SEARCH_MAX_K: the running top-k lives in registers
                check(lib.grafp_ivfpq_search_f32(_ptr(qt), nq, self.d, _ptr(self.centroids), self.nlist,
                                                 _ptr(self.codebooks), self.M, _ptr(codes), _ptr(start), _ptr(ids),
                                                 _ptr(probe), nprobe, k, _ptr(D), _ptr(I), stream), "ivfpq_search")
            else:
                self._search_dense(qt, k, probe, nprobe, codes, ids, start, counts, D, I, stream)
        return (D.cpu().numpy(), I.cpu().numpy()) if as_numpy else (D, I)

    def _search_dense(self, qt, k, probe, nprobe, codes, ids, start, counts, D, I, stream, rows_per_pass=1024):
        """k beyond the fused kernel's register list (the reference's k_probe is a free argument, eval.py:177): every
        estimate of the probed lists through grafp_ivfpq_scan_f32 -- the same table and sub-space order as the fused
        search, so the same floats -- then the k best by (distance, id) with two stable sorts.  Queries in passes that
        bound the (rows x probed codes) scratch."""
        nq = qt.shape[0]
        for lo in range(0, nq, rows_per_pass):
            qb, pb = qt[lo:lo + rows_per_pass], probe[lo:lo + rows_per_pass].contiguous()
            lens = counts[pb.long()]
            ostart = (torch.cumsum(lens, 1) - lens).contiguous()
            stride = max(1, int(lens.sum(1).max().item()))
            dist = torch.full((qb.shape[0], stride), float("inf"), device=self.device)
            pos = torch.full((qb.shape[0], stride), -1, dtype=torch.int32, device=self.device)
            check(lib.grafp_ivfpq_scan_f32(_ptr(qb), qb.shape[0], self.d, _ptr(self.centroids), self.nlist,
                                           _ptr(self.codebooks), self.M, _ptr(codes), _ptr(start), _ptr(pb), nprobe,
                                           _ptr(ostart), stride, _ptr(dist), _ptr(pos), stream), "ivfpq_scan")
            have = pos >= 0
            cid = torch.where(have, ids[pos.clamp_min(0).long()], torch.full_like(pos, -1, dtype=torch.int64))
            key = torch.where(have, cid, torch.full_like(cid, torch.iinfo(torch.int64).max))
            o1 = torch.argsort(key, dim=1, stable=True)                                # id ascending ...
            d1, i1 = torch.gather(dist, 1, o1), torch.gather(cid, 1, o1)
            o2 = torch.argsort(d1, dim=1, stable=True)[:, :k]                          # ... then distance, stable
            kk = o2.shape[1]
            D[lo:lo + qb.shape[0], :kk] = torch.gather(d1, 1, o2)
            I[lo:lo + qb.shape[0], :kk] = torch.gather(i1, 1, o2)
