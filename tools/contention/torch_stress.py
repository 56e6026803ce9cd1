"""N processes share cuda:0; each repeats a fixed chain of plain torch kernels (no grafp code) enqueued without host syncs and
checks that the result is the same bits every time."""
import os, sys, subprocess, hashlib

def worker(rank, iters, extra="none"):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    g = torch.Generator().manual_seed(100 + rank)
    a = torch.randn(2048, 2048, generator=g).to(dev)
    ws = [(torch.randn(2048, 2048, generator=g) / 45).to(dev) for _ in range(6)]
    img = torch.randn(64, 32, 64, 64, generator=g).to(dev)
    cw = (torch.randn(32, 32, 3, 3, generator=g) / 17).to(dev)
    bn = torch.nn.BatchNorm2d(32).to(dev)
    if extra != "none":
        from grafp_amd import ops
        wav = (torch.rand(64, 16000, generator=g) * 0.2 - 0.1).to(dev)
        spec = torch.randn(64, 64, 32, generator=g).to(dev)
        pw = (torch.randn(8, 3, 7, 7, generator=g) * 0.1).to(dev).requires_grad_()
        pb = torch.zeros(8, device=dev).requires_grad_()
        go = torch.randn(64, 8, 32 * 32, generator=g).to(dev)
    def side():
        if extra == "logmel1024":
            return ops.logmel(wav, 16000, 1024, 1024, 512, 64)
        if extra == "logmel2048":
            return ops.logmel(wav, 16000, 2048, 2048, 512, 64)
        if extra == "peak":
            with torch.enable_grad():
                pw.grad = pb.grad = None
                ops.peak_extract(spec, pw, pb, 2).backward(go)
            return pw.grad
        return None
    def chain():
        x = a
        y = img
        sides = []
        for rep in range(12):
            for w in ws:
                x = torch.nn.functional.gelu(x @ w)
                x = torch.nn.functional.layer_norm(x, (2048,))
            y = torch.relu(bn(torch.nn.functional.conv2d(y, cw, padding=1)))
            sides.append(side())
            x = x + y.mean() * 0.01
            x = torch.softmax(x, dim=1) * 2048 + x.cumsum(1) * 1e-3
        return x, y, sides
    hashes, shashes = [], []
    for it in range(iters):
        with torch.no_grad():
            x, y, sides = chain()
        torch.cuda.synchronize()
        hashes.append(hashlib.md5(x.cpu().numpy().tobytes() + y.cpu().numpy().tobytes()).hexdigest())
        if sides[0] is not None:
            shashes.append(hashlib.md5(b"".join(t.detach().cpu().numpy().tobytes() for t in sides)).hexdigest())
    print(f"[torch+{extra}] rank {rank}: {iters} iterations, {len(set(hashes))} distinct torch results, {len(set(shashes))} distinct side results", flush=True)

if __name__ == "__main__":
    if sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    else:
        n, iters = int(sys.argv[1]), int(sys.argv[2])
        extra = sys.argv[3] if len(sys.argv) > 3 else "none"
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(iters), extra]) for r in range(n)]
        for p in ps: p.wait()
