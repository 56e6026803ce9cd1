"""tools/search_abl.py: what the bf16 scan loop of knn_search.hip costs without each of its parts.  Runs the pre-pass
kernel (stream_tiles_bf16 + 16 fma / 8 max3 per block) over ALL rows of a 1M x 128 database for 4096 queries with parts of the
loop compiled out (measurement build only: GRAFP_HIP_LIB=.../libgrafp_hip_measure.so).
bit 0 no MFMA, bit 1 no per-block callback, bit 2 no DMA after the prologue, bit 3 no fragment reads."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafp_amd import ops  # noqa: E402
from grafp_amd._lib import lib  # noqa: E402

NAMES = {0: "full loop", 1: "no MFMA", 2: "no block callback", 3: "no MFMA, no callback", 4: "no DMA", 8: "no fragment reads",
         9: "no fragments, no MFMA", 10: "no fragments, no callback", 11: "DMA + barriers only", 12: "no DMA, no fragments",
         14: "MFMA only", 15: "barriers only", 6: "no DMA, no callback", 7: "fragment reads only"}


def main():
    n, nq = 1_000_000, 4096
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(2)
    db = torch.nn.functional.normalize(torch.randn(n, 128, generator=gen, device=dev), dim=1)
    q = torch.nn.functional.normalize(torch.randn(nq, 128, generator=gen, device=dev), dim=1)
    sq = ops.row_sqnorm(db)
    qq = ops.row_sqnorm(q)
    dbh = ops.rows_to_bf16(db)
    gmin = torch.zeros((nq, 256), dtype=torch.int32, device=dev)      # raw group maxima (floats) land here
    fn = lib.grafp_measure_search_loop
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                   ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    nqs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    print(f"# {nqs} query set(s) per wave")
    for abl in (0, 1, 2, 3, 4, 6, 8, 9, 10, 11, 12, 14, 7, 15):
        def run():
            rc = fn(dbh.data_ptr(), sq.data_ptr(), n, q.data_ptr(), qq.data_ptr(), nq, abl, nqs, gmin.data_ptr(), st)
            assert rc == 0, rc
        run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            run()
        e.record()
        torch.cuda.synchronize()
        print(f"abl={abl:2d}  {s.elapsed_time(e) / 5 * 1e3:8.1f} us   {NAMES[abl]}", flush=True)


if __name__ == "__main__":
    main()
