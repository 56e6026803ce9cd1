#!/usr/bin/env python3
"""Build-time checks on the generated code: (1) the register contract of wgrad_gr_kernel (grafp_amd/csrc/wgrad.hip), (2) no
high-register operand-selection splat in a packed-f32 subtraction of conv1x1_gemm_kernel (see check_gemm_splat).

(1):

That kernel keeps in-flight G-operand loads in the PHYSICAL registers v224-v255, named inside inline asm, and relies on
`amdgpu_num_vgpr(224)` keeping the compiler out of them.  A compiler or flag change that breaks any of the following
would give silently wrong weight gradients, so `make check` (and __graft_entry__.build(), and the CPU test-suite)
compile wgrad.hip to assembly and verify, for every instantiation:
  * the kernel descriptor allocates 256 VGPRs (so v224-v255 exist), no AGPRs, no scratch, no spills;
  * outside the `;;#ASMSTART ... ;;#ASMEND` blocks no instruction mentions v224 ... v255 (alone or inside a range).
Usage: check_kernel_regs.py [--hipcc PATH] [--asm-wgrad FILE --asm-gemm FILE] [--measure]   (exit status 0 = contract holds).
`make -C grafp_amd/csrc` runs it on the assembly of the very flags the objects were built with (the library is not linked
when it fails); without --asm-* it compiles the two sources itself.  Toolchain the contract was last verified on: the
hipcc of ROCm 7.2.0 (`hipcc --version`: AMD clang 20) -- any other compiler is checked the same way, not trusted."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "grafp_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-I" + os.path.join(ROOT, "include"), "-Wno-inline-asm",
         "--cuda-device-only", "-S"]


def high_regs(line):
    """True if an instruction line touches a VGPR >= 224 (v230, or a range like v[220:227])."""
    for m in re.finditer(r"\bv(\d+)\b", line):
        if int(m.group(1)) >= 224:
            return True
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
        if int(m.group(2)) >= 224:
            return True
    return False


def check(asm):
    errors, kernels = [], 0
    for m in re.finditer(r"^(_ZN5grafp15wgrad_gr_kernel\w+):\s*;[^\n]*\n(.*?)^\s*s_endpgm", asm, flags=re.S | re.M):
        name, body = m.group(1), m.group(2)
        kernels += 1
        in_asm = False
        for ln in body.split("\n"):
            s = ln.strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
            elif s.startswith(";;#ASMEND"):
                in_asm = False
            elif not in_asm and s and not s.startswith((";", ".")) and high_regs(s.split(";")[0]):
                errors.append(f"{name}: compiler-generated instruction touches v224+: {s}")
        meta = re.search(r"\.name:\s+" + re.escape(name) + r"\n(.*?)\.wavefront_size", asm, flags=re.S)
        before = asm[:meta.start()] if meta else ""
        blk = (before[before.rfind("- .agpr_count"):] if meta else "") + (meta.group(0) if meta else "")
        def field(key):
            f = re.search(r"\." + key + r":\s+(\d+)", blk)
            return int(f.group(1)) if f else None
        want = {"vgpr_count": 256, "agpr_count": 0, "vgpr_spill_count": 0, "private_segment_fixed_size": 0}
        for key, val in want.items():
            got = field(key)
            if got != val:
                errors.append(f"{name}: .{key} = {got}, expected {val}")
    if kernels == 0:
        errors.append("no wgrad_gr_kernel instantiation found in the assembly")
    return kernels, errors


def check_gemm_splat(asm):
    """conv1x1_gemm_kernel (gemm.hip): no packed-f32 subtraction may take its subtrahend by the HIGH-register splat of a
    register pair (`v_pk_add_f32 d, a, v[n:n+1] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]`).  That is the form hipcc chose for
    the statistics shift of the odd row tile when two shifts shared a pair; on gfx950 one accumulate of lanes 48-63 then
    took 0 instead of the shift in a few workgroups per launch (round 3: run-to-run differences of the partial
    statistics).  The kernel keeps the shift in a pair of its own; this check notices if a compiler change undoes that."""
    errors, kernels = [], 0
    for m in re.finditer(r"^(_ZN5grafp19conv1x1_gemm_kernel\w+):\s*;[^\n]*\n(.*?)^\s*s_endpgm", asm, flags=re.S | re.M):
        kernels += 1
        for ln in m.group(2).split("\n"):
            if "v_pk_add_f32" in ln and "op_sel:[0,1]" in ln and "neg_lo:[0,1]" in ln:
                errors.append(f"{m.group(1)[:60]}...: {ln.strip()}")
    if kernels == 0:
        errors.append("no conv1x1_gemm_kernel instantiation found in the assembly")
    return kernels, errors


def check_mfma_packed_select(asm):
    """EVERY kernel that issues matrix instructions: no packed-f32 instruction may take the LOW lane of a source from the
    HIGH register of its pair (`v_pk_{add,mul,fma}_f32 ... op_sel:[..1..]`).  Measured on MI355X (DESIGN.md section 12.7b,
    tools/contention/two_stream.py pk_add_hi / pk_add_swap): while bf16 MFMA waves are active on the SIMD -- the kernel's
    own or, under two streams / processes, another kernel's -- that form returns a wrong value on lanes 48-63 in a third of
    the launches; the forms that only broadcast the LOW register (`op_sel_hi:[1,0]`) and the unselected ones do not, nor does the same
    select on the first or third source (pk_fma_hi0 / pk_fma_hi2) -- the rule here is the conservative one, any source.
    check_gemm_splat is the instance found in round 3; this is the rule for the whole library."""
    errors, kernels = [], 0
    for m in re.finditer(r"^(\w+):\s*; @\w+\s*$", asm, flags=re.M):
        name = m.group(1)
        end = asm.find(".amdhsa_kernel " + name, m.start())
        if end < 0:
            continue
        body = asm[m.start():end]
        if not re.search(r"^\s*v_mfma", body, flags=re.M):
            continue
        kernels += 1
        for ln in body.split("\n"):
            if re.match(r"\s*v_pk_\w+_f32", ln) and re.search(r"op_sel:\[[01,]*1[01,]*\]", ln):
                errors.append(f"{name[:60]}...: {ln.strip()}")
    return kernels, errors[:20]


def compile_all(hipcc, flags, jobs=8):
    """Device assembly of every csrc/*.hip (the product flags), in parallel: name -> text."""
    import concurrent.futures
    import glob
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        def one(src):
            dst = os.path.join(tmp, os.path.basename(src) + ".s")
            res = subprocess.run([hipcc] + flags + [src, "-o", dst], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if res.returncode != 0:
                raise RuntimeError(res.stdout[-2000:])
            return os.path.basename(src), open(dst).read()
        with concurrent.futures.ThreadPoolExecutor(jobs) as ex:
            for name, text in ex.map(one, sorted(glob.glob(os.path.join(CSRC, "*.hip")))):
                out[name] = text
    return out


def check_gemm_xl(asm):
    """conv1x1_gemm_xl_kernel (gemm_xl.h): its 256 accumulators are AGPRs a0-a255 NAMED in asm text; the compiler only
    knows sixteen placeholder values that occupy the file from the first instruction to the last.  Every build must show:
    no v_accvgpr_* instruction outside the ;;#ASMSTART / ;;#ASMEND blocks (the compiler moving a placeholder or parking a
    value of its own in an accumulator is silent corruption), no scratch access (spilled VGPRs), .agpr_count 256."""
    errors, kernels = [], 0
    for m in re.finditer(r"^(_ZN5grafp22conv1x1_gemm_xl_kernel\w+):\s*;[^\n]*\n(.*?)^\s*s_endpgm", asm, flags=re.S | re.M):
        kernels += 1
        name, in_asm = m.group(1), False
        for ln in m.group(2).split("\n"):
            t = ln.strip()
            if ";;#ASMSTART" in t:
                in_asm = True
            elif ";;#ASMEND" in t:
                in_asm = False
            elif not in_asm and t.startswith("v_accvgpr"):
                errors.append(f"{name[:60]}...: compiler-generated {t}")
            elif t.startswith("scratch_"):
                errors.append(f"{name[:60]}...: scratch access {t}")
        meta = re.search(r"\.name:\s+" + re.escape(name) + r"\n(.*?)\.wavefront_size", asm, flags=re.S)
        before = asm[:meta.start()] if meta else ""
        agpr = re.findall(r"\.agpr_count:\s+(\d+)", before)
        if not agpr or agpr[-1] != "256":
            errors.append(f"{name[:60]}...: .agpr_count = {agpr[-1] if agpr else None}, expected 256")
    if kernels == 0:
        errors.append("no conv1x1_gemm_xl_kernel instantiation found in the assembly")
    return kernels, errors[:20]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hipcc", default=os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))
    ap.add_argument("--asm-wgrad", default=None, help="assembly of wgrad.hip (device only) instead of compiling it here")
    ap.add_argument("--asm-gemm", default=None, help="assembly of gemm.hip")
    ap.add_argument("--measure", action="store_true", help="compile with -DGRAFP_MEASURE (the measurement library)")
    ap.add_argument("--all", action="store_true", help="only the packed-select rule, over EVERY csrc/*.hip (compiles them all)")
    args = ap.parse_args()
    flags = FLAGS + (["-DGRAFP_MEASURE"] if args.measure else [])
    if args.all:
        total, bad = 0, []
        for name, text in compile_all(args.hipcc, flags).items():
            k, errs = check_mfma_packed_select(text)
            total += k
            bad += [f"{name}: {e}" for e in errs]
        for e in bad:
            print("PACKED-F32 LOW LANE FROM A HIGH REGISTER IN A MATRIX KERNEL:", e)
        if not bad:
            print(f"no packed-f32 high-register select in any of the {total} kernels that issue matrix instructions")
        return 1 if bad else 0
    with tempfile.TemporaryDirectory() as tmp:
        out = args.asm_wgrad or os.path.join(tmp, "wgrad.s")
        if not args.asm_wgrad:
            res = subprocess.run([args.hipcc] + flags + [os.path.join(CSRC, "wgrad.hip"), "-o", out],
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if res.returncode != 0:
                print(res.stdout)
                return 2
        asm = open(out).read()
    kernels, errors = check(asm)
    for e in errors:
        print("REGISTER CONTRACT VIOLATED:", e)
    if not errors:
        print(f"wgrad_gr_kernel register contract holds for {kernels} instantiations")
    with tempfile.TemporaryDirectory() as tmp:
        out = args.asm_gemm or os.path.join(tmp, "gemm.s")
        if not args.asm_gemm:
            res = subprocess.run([args.hipcc] + flags + [os.path.join(CSRC, "gemm.hip"), "-o", out],
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if res.returncode != 0:
                print(res.stdout)
                return 2
        gemm_asm = open(out).read()
        gk, gerrors = check_gemm_splat(gemm_asm)
        xk, xerrors = check_gemm_xl(gemm_asm)
    for e in xerrors:
        print("ACCUMULATOR FILE TOUCHED BY THE COMPILER:", e)
    if not xerrors:
        print(f"conv1x1_gemm_xl_kernel: accumulator registers untouched by the compiler ({xk} instantiations)")
    pk, perrors = check_mfma_packed_select(gemm_asm)
    pk2, perrors2 = check_mfma_packed_select(asm)
    for e in perrors + perrors2:
        print("PACKED-F32 LOW LANE FROM A HIGH REGISTER IN A MATRIX KERNEL:", e)
    if not (perrors or perrors2):
        print(f"gemm.hip + wgrad.hip: no packed-f32 high-register select in {pk + pk2} kernels that issue matrix instructions")
    gerrors = gerrors + perrors + perrors2
    for e in gerrors[:len(gerrors) - len(perrors) - len(perrors2)]:
        print("HIGH-REGISTER SPLAT IN A PACKED SUBTRACTION:", e)
    if not gerrors:
        print(f"conv1x1_gemm_kernel: no high-register splat in a packed subtraction ({gk} instantiations)")
    return 1 if (errors or gerrors or xerrors) else 0


if __name__ == "__main__":
    sys.exit(main())
