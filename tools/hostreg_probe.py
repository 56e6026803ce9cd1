import numpy as np, torch, tempfile, os, ctypes
mm = np.memmap(os.path.join(tempfile.mkdtemp(), "x.mm"), dtype="float32", mode="w+", shape=(1<<20, 128))
rt = torch.cuda.cudart()
t = torch.from_numpy(mm)
try:
    rc = rt.cudaHostRegister(t.data_ptr(), t.numel()*4, 0)
    print("hipHostRegister on file-backed memmap pages ->", rc)
    if int(rc) == 0:
        x = torch.randn(1<<20, 128, device="cuda")
        import time; torch.cuda.synchronize(); t0=time.time(); t.copy_(x, non_blocking=True); torch.cuda.synchronize(); print("direct D2H GB/s", t.numel()*4/(time.time()-t0)/1e9, bool((t==x.cpu()).all()))
        rt.cudaHostUnregister(t.data_ptr())
except Exception as e:
    print("exception", type(e).__name__, e)
