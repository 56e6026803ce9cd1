"""`import faiss` for the reference's scripts (eval.py:3, test_fp.py:7) without faiss.

The subset of the faiss API that the reference touches (eval.py:42-43, 54-122; test_fp.py only imports it), served by this
package's resident-in-HBM indexes: `IndexFlatL2` IS ops.FlatL2Index (exact search, bit-equal to the CPU oracle),
`IndexIVFPQ(quantizer, d, nlist, M, nbits)` is ivfpq.IVFPQIndex (the published protocol's approximate index), the GPU
plumbing (`StandardGpuResources`, `GpuClonerOptions`, `index_cpu_to_gpu*`) is inert -- an index of this package already
lives on the device.  The other index types of eval.py:58-100 ('ivf', 'ivfpq-rr', 'lsh', 'hnsw': approximations of the
same search) come back as the exact index, a superset in accuracy, with a one-line notice -- what grafp_amd.eval.get_index
does for them too.  `grafp_amd.dropin.install()` registers this module as `faiss` unless a real faiss is importable.
"""
from .ops import FlatL2Index as IndexFlatL2  # noqa: F401  (same constructor: IndexFlatL2(d))

METRIC_L2 = 1
METRIC_INNER_PRODUCT = 0
INDICES_CPU = 0


class StandardGpuResources:
    """faiss.StandardGpuResources(): scratch-memory management of faiss-gpu; nothing to manage here."""

    def setTempMemory(self, nbytes):
        pass

    def noTempMemory(self):
        pass


class GpuClonerOptions:
    """faiss.GpuClonerOptions(): attributes the reference sets (eval.py:44-46) are accepted and ignored."""

    def __init__(self):
        self.useFloat16 = False
        self.usePrecomputed = False
        self.indicesOptions = INDICES_CPU
        self.reserveVecs = 0
        self.storeTransposed = False
        self.verbose = False


GpuMultipleClonerOptions = GpuClonerOptions


def index_cpu_to_gpu(resources, device, index, options=None):
    return index


def index_cpu_to_all_gpus(index, co=None, ngpu=-1):
    return index


def index_cpu_to_gpu_multiple_py(resources, index, co=None, gpus=None):
    return index


def index_gpu_to_cpu(index):
    return index


def get_num_gpus():
    import torch
    return torch.cuda.device_count()


def IndexIVFPQ(quantizer, d, nlist, M, nbits, metric=METRIC_L2):
    """faiss.IndexIVFPQ(quantizer, d, nlist, M, nbits) (eval.py:69): the quantizer argument is the flat index faiss trains
    its coarse centroids in; IVFPQIndex runs its own coarse k-means."""
    from .ivfpq import IVFPQIndex
    return IVFPQIndex(int(d), nlist=int(nlist), M=int(M), nbits=int(nbits))


def _exact(name, d):
    print(f"faiss.{name}: served by exact brute-force L2 search on the GPU (grafp_amd.ops.FlatL2Index)")
    return IndexFlatL2(int(d))


def IndexIVFFlat(quantizer, d, nlist, metric=METRIC_L2):
    return _exact("IndexIVFFlat", d)


def IndexIVFPQR(quantizer, d, nlist, M, nbits, M_refine, nbits_refine):
    return _exact("IndexIVFPQR", d)


def IndexLSH(d, nbits):
    return _exact("IndexLSH", d)


def IndexHNSWFlat(d, M, metric=METRIC_L2):
    return _exact("IndexHNSWFlat", d)
