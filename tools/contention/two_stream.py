"""ONE process, two streams: stream A repeats a kernel on fixed inputs and checks its output bits, stream B keeps the device
busy with matrix products.  What it showed: next to a skinny bf16 GEMM -- torch's hipBLASLt
kernel or this repository's -- a packed-f32 instruction whose LOW lane reads the HIGH register of a pair (victims pk_add_hi,
pk_add_swap; hipcc emits the form in logmel / bn / peak) is wrong on lanes 48-63 in a third of the launches; every other
victim below, and every victim without the neighbour, is bit-stable.  OFFENDER_PER_ROUND (default 6) = products per 24
victim launches.

    python tools/contention/two_stream.py VICTIM OFFENDER SECONDS
VICTIM: bn | bn2 | bn_spin0 | logmel | logmel512 | logmel2048 | logmel_dbg (logmel with a per-stage trace) | peak | mm | t_layernorm | t_softmax | t_batchnorm | t_cumsum | t_gelu | t_conv (torch only) | copy4 | copy16 | copy32 | alu_pk | alu_scalar | lds8 | lds64 | conf_2way | conf_1bank | shfl | barrier | sgpr_chain | vgpr_chain | pk_plain | pk_sel | pk_mul_lo | pk_add_swap | pk_add_hi | pk_fma_lo | pk_fma_hi0 | pk_fma_hi2 | pk_fma_hi1 | pk_mul_hi1 | wide64 | wide128 | wide224 (inflight.hip); OFFENDER: mmbf16 | mmf32 | mmbf16small | mmf32small | mmbf16mid | ewadd (torch) | gemm | knn | mr | wgrad (kernels of this repository) | none"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from grafp_amd import ops  # noqa: E402

victim, offender, secs = sys.argv[1], sys.argv[2], float(sys.argv[3])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
g = torch.Generator(device="cpu").manual_seed(5)
if victim in ("bn", "bn2", "bn_spin0"):
    ops.switches.bn_two_pass = victim == "bn2"
    if victim == "bn_spin0":
        ops.switches.bn_spin_limit = 0          # never wait for row-mates: every workgroup recomputes what is missing
    C, M = 160, 256 * 512
    x = torch.randn(C, M, generator=g).to(dev)
    ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    f = lambda: ops.bn_act(x, ga, be, rm.clone(), rv.clone(), True, act=1, groups=2)
elif victim in ("logmel", "logmel2048", "logmel512"):
    nf = int(victim[6:] or 1024)                 # 1024: the register-FFT kernel; 512 / 2048: the generic radix-2 kernel
    wav = (torch.rand(256, 16000, generator=g) * 2 - 1).to(dev)
    f = lambda: ops.logmel(wav, 16000, nf, nf, 512, 64)
elif victim in ("logmel_dbg", "logmel_dbg1", "logmel_dbg2", "logmel_dbg3", "logmel_dbg4"):                   # logmel_dbg.hip: the register-FFT kernel with a per-stage trace per frame pair
    import ctypes
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblogmel_dbg.so")
    if not os.path.exists(so):
        raise SystemExit("build it first: (cd tools/contention && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off "
                         "-fhip-fp32-correctly-rounded-divide-sqrt -I../../include -Wno-inline-asm -fPIC -shared logmel_dbg.hip "
                         "-o liblogmel_dbg.so)")
    cl = ctypes.CDLL(so)
    vp = ctypes.c_void_p
    cl.logmel1024_dbg_launch.argtypes = [vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp,
                                         vp, vp, vp, ctypes.c_int, vp]
    variant = int(victim[10:] or 0)               # 1: stop after the first FFT; 2: the same with no LDS and no barrier at all
    wav = (torch.rand(256, 16000, generator=g) * 2 - 1).to(dev)
    plan = ops._mel_plan(dev, 16000, 1024, 1024, 64)
    STAGES = ("windowed samples", "first FFT x twiddle", "transposed read", "second FFT", "power spectrum", "mel sums")

    def f():
        out = torch.zeros((256, 64, 32), dtype=torch.float32, device=dev)
        dbg = torch.zeros((256, 16, 6), dtype=torch.int32, device=dev)
        rc = cl.logmel1024_dbg_launch(wav.data_ptr(), wav.stride(0), 256, 16000, 512, 64, plan.window.data_ptr(),
                                      plan.twiddle.data_ptr(), plan.fb.data_ptr(), plan.band_lo.data_ptr(),
                                      plan.band_hi.data_ptr(), out.data_ptr(), dbg.data_ptr(), variant,
                                      torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
        assert rc == 0, rc
        return torch.cat((dbg.reshape(256, -1), out.reshape(256, -1).view(torch.int32)), dim=1)      # (256, 96 + 2048)
elif victim == "peak":
    spec = torch.randn(256, 64, 32, generator=g).to(dev)
    w = (torch.randn(8, 3, 7, 7, generator=g) * 0.1).to(dev)
    b = torch.zeros(8, device=dev)
    f = lambda: ops.peak_extract(spec, w, b, 2)
elif victim.startswith("copy"):                # tools/contention/inflight.hip: N 16-byte loads per thread in flight, then N stores
    import ctypes
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libinflight.so")
    if not os.path.exists(so):
        raise SystemExit("build it first: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared tools/contention/inflight.hip "
                         "-o tools/contention/libinflight.so")
    cl = ctypes.CDLL(so)
    cl.inflight_copy_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    per = int(victim[4:])
    n_vec = 256 * per * 2048                                    # 2048 workgroups
    src = torch.randn(n_vec * 4, generator=g).to(dev)

    def f():
        out = torch.empty_like(src)
        rc = cl.inflight_copy_launch(src.data_ptr(), out.data_ptr(), n_vec, per,
                                     torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
        assert rc == 0, rc
        return out
elif victim in ("alu_pk", "alu_scalar"):       # inflight.hip: a chain of packed (v_pk_fma_f32) / scalar (v_fma_f32) multiply-adds
    import ctypes
    cl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libinflight.so"))
    cl.alu_chain_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    src = torch.rand(65536, generator=g).to(dev)

    def f():
        out = torch.empty(2048 * 256, device=dev)
        rc = cl.alu_chain_launch(src.data_ptr(), out.data_ptr(), 2048, 400, int(victim == "alu_pk"),
                                 torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
        assert rc == 0, rc
        return out
elif victim.startswith("lds"):                 # inflight.hip: a pattern held in LDS (ldsNN = NN KB per workgroup) and re-read for ~100 us
    import ctypes
    cl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libinflight.so"))
    cl.lds_hold_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    kb = int(victim[3:])

    def f():
        out = torch.zeros(1024, dtype=torch.int32, device=dev)
        rc = cl.lds_hold_launch(out.data_ptr(), 1024, 40, kb * 1024, torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
        assert rc == 0, rc
        return out
elif victim in ("sgpr_chain", "vgpr_chain"):    # inflight.hip: v_fma_f32 chains whose multiplier is an SGPR / a VGPR
    import ctypes
    cl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libinflight.so"))
    cl.sgpr_chain_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    src = torch.rand(65536, generator=g).to(dev)

    def f():
        out = torch.empty(2048 * 256, device=dev)
        rc = cl.sgpr_chain_launch(src.data_ptr(), out.data_ptr(), 2048, 300, int(victim == "sgpr_chain"),
                                  torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
        assert rc == 0, rc
        return out.reshape(2048 * 4, 64)                      # one row per wave: the column is the lane
elif victim.startswith("wide"):                # inflight.hip: wideNN = NN accumulators per thread, all live, plain arithmetic
    import ctypes
    cl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libinflight.so"))
    cl.wide_chain_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    src = torch.rand(65536, generator=g).to(dev)
    nreg = int(victim[4:])

    def f():
        out = torch.empty(2048 * 256, device=dev)
        rc = cl.wide_chain_launch(src.data_ptr(), out.data_ptr(), 2048, 2400 // nreg, nreg,
                                  torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
        assert rc == 0, rc
        return out.reshape(2048 * 4, 64)                      # one row per wave: the column is the lane
elif victim in ("conf_2way", "conf_1bank", "shfl"):  # inflight.hip: LDS round trips with bank conflicts / shuffle butterflies
    import ctypes
    cl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libinflight.so"))
    cl.lds_conflict_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    mode = ("conf_2way", "conf_1bank", "shfl").index(victim)

    def f():
        out = torch.zeros(2048, dtype=torch.int32, device=dev)
        rc = cl.lds_conflict_launch(out.data_ptr(), 2048, 60, mode, torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
        assert rc == 0, rc
        return out
elif victim in ("pk_plain", "pk_sel", "pk_mul_lo", "pk_add_swap", "pk_add_hi", "pk_fma_lo", "pk_fma_hi0", "pk_fma_hi2", "pk_fma_hi1", "pk_mul_hi1"):         # inflight.hip: packed-f32 chains with / without lane selects on the second source
    import ctypes
    cl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libinflight.so"))
    cl.pk_sel_chain_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    src = torch.rand(65536, generator=g).to(dev)

    def f():
        out = torch.empty(2048 * 256, device=dev)
        rc = cl.pk_sel_chain_launch(src.data_ptr(), out.data_ptr(), 2048, 300,
                                    ("pk_plain", "pk_sel", "pk_mul_lo", "pk_add_swap", "pk_add_hi", "pk_fma_lo", "pk_fma_hi0", "pk_fma_hi2", "pk_fma_hi1", "pk_mul_hi1").index(victim),
                                    torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
        assert rc == 0, rc
        return out.reshape(2048 * 4, 64)                      # one row per wave: the column is the lane
elif victim == "barrier":                      # inflight.hip: 200 rounds of publish / barrier / read a word of another wave
    import ctypes
    cl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libinflight.so"))
    cl.barrier_ring_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

    def f():
        out = torch.zeros(2048, dtype=torch.int32, device=dev)
        rc = cl.barrier_ring_launch(out.data_ptr(), 2048, 200, 8, torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))
        assert rc == 0, rc
        return out
elif victim == "t_layernorm":
    a1 = torch.randn(8192, 2048, generator=g).to(dev)
    f = lambda: torch.nn.functional.layer_norm(a1, (2048,))
elif victim == "t_softmax":
    a1 = torch.randn(8192, 2048, generator=g).to(dev)
    f = lambda: torch.softmax(a1, dim=1)
elif victim == "t_batchnorm":
    a1 = torch.randn(256, 160, 512, generator=g).to(dev)
    f = lambda: torch.nn.functional.batch_norm(a1, None, None, training=True)
elif victim == "t_cumsum":
    a1 = torch.randn(8192, 2048, generator=g).to(dev)
    f = lambda: torch.cumsum(a1, dim=1)
elif victim == "t_gelu":
    a1 = torch.randn(8192, 2048, generator=g).to(dev)
    f = lambda: torch.nn.functional.gelu(a1 * 1.5 + a1.sin())
elif victim == "t_conv":
    a1 = torch.randn(64, 32, 64, 64, generator=g).to(dev); cw = (torch.randn(32, 32, 3, 3, generator=g) / 17).to(dev)
    f = lambda: torch.nn.functional.conv2d(a1, cw, padding=1)
else:
    a1, a2 = torch.randn(2048, 2048, generator=g).to(dev), torch.randn(2048, 2048, generator=g).to(dev)
    f = lambda: a1 @ a2
if offender == "mmbf16":
    o1 = torch.randn(4096, 4096, device=dev).bfloat16()
    off = lambda: o1 @ o1
elif offender == "mmf32":
    o1 = torch.randn(4096, 4096, device=dev)
    off = lambda: o1 @ o1
elif offender == "mmbf16small":
    o1, o2 = torch.randn(128, 128, device=dev).bfloat16(), torch.randn(128, 262144, device=dev).bfloat16()
    off = lambda: o1 @ o2
elif offender == "mmf32small":
    o1, o2 = torch.randn(128, 128, device=dev), torch.randn(128, 262144, device=dev)
    off = lambda: o1 @ o2
elif offender == "mmbf16mid":
    o1, o2 = torch.randn(128, 128, device=dev).bfloat16(), torch.randn(128, 32768, device=dev).bfloat16()
    off = lambda: o1 @ o2
elif offender == "ewadd":                     # a streaming elementwise kernel over the same 64 MB, no matrix cores
    o1, o2 = torch.randn(128, 262144, device=dev).bfloat16(), torch.randn(128, 262144, device=dev).bfloat16()
    off = lambda: o1 + o2
elif offender == "gemm":                      # this repository's streaming GEMM on the same product as mmbf16small
    o1, o2 = torch.randn(128, 128, device=dev).bfloat16(), torch.randn(128, 262144, device=dev).bfloat16()
    off = lambda: ops.conv1x1_gemm(o1, o2)
elif offender == "knn":
    o1 = torch.randn(64, 96, 1024, device=dev).bfloat16()
    off = lambda: ops.knn_graph(o1, 3)
elif offender == "mr":
    o1 = torch.randn(64, 96, 1024, device=dev).bfloat16()
    o2 = torch.randint(0, 1024, (64, 1024, 3), device=dev)
    off = lambda: ops.max_relative(o1, o2)
elif offender == "wgrad":
    o1, o2 = torch.randn(128, 262144, device=dev).bfloat16(), torch.randn(128, 262144, device=dev).bfloat16()
    off = lambda: ops._wgrad_bf16(o1, o2, 128, 128, 1, 262144, may_defer=False)
else:
    off = None
with torch.no_grad():
    ref = f().clone()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    t0, n, nbad = time.time(), 0, 0
    while time.time() - t0 < secs:
        if off is not None:
            with torch.cuda.stream(sb):
                for _ in range(int(os.environ.get("OFFENDER_PER_ROUND", "6"))):    # per 24 victim launches
                    keep = off()
        with torch.cuda.stream(sa):
            flags = []
            outs = []
            for _ in range(24):
                outs.append(f())
                flags.append((outs[-1] != ref).any())
                n += 1
            bad = int(torch.stack(flags).sum())
            if bad and victim.startswith("logmel_dbg") and nbad < 6:
                for o, fl in zip(outs, flags):
                    if not bool(fl):
                        continue
                    dd = (o[:, :96] != ref[:, :96]).reshape(256, 16, 6)
                    pairs = dd.any(2).nonzero().tolist()
                    firsts = [STAGES[int(dd[b_, p_].nonzero()[0])] for b_, p_ in pairs[:6]]
                    outd = (o[:, 96:] != ref[:, 96:]).reshape(256, 64, 32)
                    if not pairs:
                        continue
                    print(f"   wrong launch: {len(pairs)} frame pairs with a differing trace; (clip, pair) {pairs[:6]}; first differing stage "
                          f"{firsts}; all differing stages of the first: {[STAGES[i] for i in dd[pairs[0][0], pairs[0][1]].nonzero().flatten().tolist()] if pairs else None}; "
                          f"outputs differing {int(outd.sum())} in clips {sorted(set(outd.nonzero()[:, 0].tolist()))[:6]}", flush=True)
            elif bad and (victim in ("sgpr_chain", "vgpr_chain") or victim.startswith("wide") or victim.startswith("pk_")) and nbad < 6:
                o = next(t for t, fl in zip(outs, flags) if bool(fl))
                d = o != ref
                lanes = d.any(0).nonzero().flatten().tolist()
                print(f"   wrong launch: {int(d.sum())} values in {int(d.any(1).sum())} waves; lanes {lanes}", flush=True)
            elif bad and nbad < 3 and ref.dim() >= 2:
                o = next(t for t, fl in zip(outs, flags) if bool(fl))
                d = (o != ref).reshape(ref.shape[0], -1)
                rows = d.any(1).nonzero().flatten().tolist()
                desc = [(r, int(d[r].sum()), int(d[r].nonzero().min()), int(d[r].nonzero().max())) for r in rows[:5]]
                print(f"   wrong launch: {int(d.sum())} elements in {len(rows)} of {ref.shape[0]} rows; (row, count, first, last) {desc}", flush=True)
                if victim.startswith("bn"):
                    # f32 rows: element m of a row sits in lane ((m / 4) % 256) % 64 of its workgroup (bn_fwd1: 4 floats per
                    # 16-byte vector, 256 threads); rows whose statistics moved differ on every lane, a workgroup whose OWN
                    # arithmetic was hit differs on the lanes that were hit
                    cols = d.nonzero()[:, 1]
                    lanes = ((cols // 4) % 256) % 64
                    hist = torch.bincount(lanes, minlength=64)
                    print(f"      wrong elements by lane quarter [0-15, 16-31, 32-47, 48-63]: {[int(hist[q * 16:(q + 1) * 16].sum()) for q in range(4)]}", flush=True)
        nbad += bad
    torch.cuda.synchronize()
print(f"[one process, two streams] victim {victim}, offender {offender}: {n} launches, {nbad} bad", flush=True)
