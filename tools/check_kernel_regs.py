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
Usage: check_kernel_regs.py [--hipcc PATH]      (exit status 0 = contract holds)."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hipcc", default=os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "wgrad.s")
        res = subprocess.run([args.hipcc] + FLAGS + [os.path.join(CSRC, "wgrad.hip"), "-o", out],
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
        out = os.path.join(tmp, "gemm.s")
        res = subprocess.run([args.hipcc] + FLAGS + [os.path.join(CSRC, "gemm.hip"), "-o", out],
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            print(res.stdout)
            return 2
        gk, gerrors = check_gemm_splat(open(out).read())
    for e in gerrors:
        print("HIGH-REGISTER SPLAT IN A PACKED SUBTRACTION:", e)
    if not gerrors:
        print(f"conv1x1_gemm_kernel: no high-register splat in a packed subtraction ({gk} instantiations)")
    return 1 if (errors or gerrors) else 0


if __name__ == "__main__":
    sys.exit(main())
