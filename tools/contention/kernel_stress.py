"""Several processes share cuda:0; each repeats ONE kernel on fixed inputs and reports where an output differs from its own
first output.

    [PRELUDE=bf16|f32[,fwdonly][,free][,depthN]] python tools/contention/kernel_stress.py NPROC KIND[,KIND..] SECONDS [--disjoint-cus]

KIND: logmel (the register-FFT kernel), logmel512 / logmel2048 (the generic kernel), peak (peak extractor forward +
backward), bn / bn2 (single-pass / two-pass BatchNorm forward + backward), mm (torch.mm, no code of this repository);
rank r runs KIND[r % len].  PRELUDE runs two training steps of the model in the process first (fwdonly: forward only;
depthN: peak extractor, weight preparation, stem and the first N backbone modules only; free: the model is deleted and the
allocator emptied before the loop).
Findings on MI355X / ROCm 7.2: without a prelude every kind is bit-stable over millions of
launches.  With a bf16 prelude logmel, peak and bn show wrong 64-byte pieces in their FIRST ~1 000 launches -- while the
other processes are still running their prelude, i.e. next to those processes' bf16 GEMMs (the deeper the prelude runs,
the longer that window) -- logmel512 / logmel2048 / bn2 / mm never; an f32 prelude: never; disjoint CU sets: never.
two_stream.py shows the same in ONE process (the GEMM on a second stream) and isolates the cause: a packed-f32
operand form (low lane from the high register of a pair) that reads wrong on lanes 48-63 next to bf16 MFMA waves."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def report(tag, rank, it, name, out, ref):
    import torch
    bad = (out != ref) & ~(torch.isnan(out) & torch.isnan(ref))
    idx = bad.nonzero()
    vals = [(float(out[tuple(i)]), float(ref[tuple(i)])) for i in idx[:4]]
    print(f"[{tag}] rank {rank} launch {it} {name}{tuple(out.shape)}: {int(bad.sum())} differ, index range "
          f"{idx.min(0).values.tolist()}..{idx.max(0).values.tolist()}, first {idx[:4].tolist()}, (got, want) {vals}", flush=True)


def prelude(pre, dev):
    import torch
    from torch import nn
    from grafp_amd import ops
    from grafp_amd.encoder._dense import conv_bn_act, deferred_counters, to_cbn
    from grafp_amd.encoder.graph_encoder import Downsample
    from grafp_amd.simclr.ntxent import ntxent_loss
    from grafp_amd.train import Trainer, build_model, synthetic_batch
    from grafp_amd.util import load_config
    cfg = load_config()
    cfg["bsz_train"] = 128
    torch.manual_seed(1234)
    model = build_model(cfg, device=dev)
    tr = Trainer(cfg, model, dev, amp_dtype=torch.bfloat16 if "bf16" in pre else None)
    xi, xj = synthetic_batch(128, 7, dev)
    model.train()
    depth = next((int(o[5:]) for o in pre.split(",") if o.startswith("depth")), None)

    def partial(X):
        enc = model.encoder
        x = to_cbn(model.peak_extractor(X))
        if enc._lowp is None:
            enc._lowp = ops.lowp_weights([m for m in enc.modules() if isinstance(m, nn.Conv2d) and m.kernel_size == (1, 1)
                                          and m is not enc.proj])
        enc._lowp.refresh(torch.get_autocast_dtype("cuda"))
        with deferred_counters():
            x = conv_bn_act(enc.stem[0], enc.stem[1], x, act=ops.ACT_LEAKY, slope=enc.stem[2].negative_slope, groups=2)
            for i, mod in enumerate(enc.backbone):
                if i >= depth:
                    break
                x = mod.forward_cbn(x, 2) if isinstance(mod, Downsample) else mod[1].forward_cbn(mod[0].forward_cbn(x, 2), 2)

    for _ in range(2):
        with torch.no_grad():
            Xi, Xj = tr.augment(xi, xj)
        if depth is not None:
            with tr._autocast(), torch.no_grad():
                partial(torch.cat((Xi, Xj), 0))
            continue
        with tr._autocast():
            _, _, zi, zj = model(Xi, Xj)
        if "fwdonly" not in pre:
            ntxent_loss(zi, zj, cfg).backward()
    torch.cuda.synchronize()
    if "free" in pre:
        del model, tr
        torch.cuda.empty_cache()
        return None
    return model, tr


def worker(rank, what, secs):
    import torch
    from grafp_amd import ops
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    g = torch.Generator(device="cpu")
    g.manual_seed(5 + rank)
    keep = prelude(os.environ["PRELUDE"], dev) if os.environ.get("PRELUDE") else None
    if what.startswith("logmel"):
        nf = int(what[6:] or 1024)
        wav = (torch.rand(256, 16000, generator=g) * 2 - 1).to(dev)
        f = lambda: {"X": ops.logmel(wav, 16000, nf, nf, 512, 64)}
    elif what == "peak":
        spec = torch.randn(256, 64, 32, generator=g).to(dev)
        w = (torch.randn(8, 3, 7, 7, generator=g) * 0.1).to(dev).requires_grad_()
        b = torch.zeros(8, device=dev).requires_grad_()
        go = torch.randn(256, 8, 32 * 32, generator=g).to(dev)

        def f():
            w.grad = b.grad = None
            out = ops.peak_extract(spec, w, b, 2)
            out.backward(go)
            return {"out": out.detach(), "dw": w.grad, "db": b.grad}
    elif what in ("bn", "bn2"):
        ops.switches.bn_two_pass = what == "bn2"
        C, M = 160, 256 * 512
        x = torch.randn(C, M, generator=g).to(dev).requires_grad_()
        ga, be = torch.ones(C, device=dev).requires_grad_(), torch.zeros(C, device=dev).requires_grad_()
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        go = torch.randn(C, M, generator=g).to(dev)

        def f():
            x.grad = ga.grad = be.grad = None
            out = ops.bn_act(x, ga, be, rm.clone(), rv.clone(), True, act=1, groups=2)
            out.backward(go)
            return {"out": out.detach(), "dx": x.grad, "dgamma": ga.grad}
    elif what == "mm":
        a, b2 = torch.randn(2048, 2048, generator=g).to(dev), torch.randn(2048, 2048, generator=g).to(dev)
        f = lambda: {"c": a @ b2}
    else:
        raise SystemExit(what)
    ref = {k: v.clone() for k, v in f().items()}
    torch.cuda.synchronize()
    t0, it, nbad = time.time(), 0, 0
    while time.time() - t0 < secs:
        it += 1
        out = f()
        if bool(torch.stack([(out[k] != ref[k]).any() for k in ref]).any()):
            nbad += 1
            if nbad <= 3:
                for k in ref:
                    if not torch.equal(out[k], ref[k]):
                        report(what, rank, it, k, out[k], ref[k])
    print(f"[{what}] rank {rank}: {it} launches, {nbad} bad", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), sys.argv[3], float(sys.argv[4]))
    else:
        args = [a for a in sys.argv[1:] if not a.startswith("--")]
        n, kinds, secs = int(args[0]), args[1].split(","), float(args[2])
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from step_stress import cu_mask
        procs = []
        for r in range(n):
            env = dict(os.environ)
            if "--disjoint-cus" in sys.argv:
                env["ROC_GLOBAL_CU_MASK"] = cu_mask(r, n)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), kinds[r % len(kinds)],
                                           str(secs)], env=env))
        sys.exit(max(p.wait() for p in procs))
