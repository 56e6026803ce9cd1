"""CPU test of the build-time register contract of wgrad_gr_kernel (tools/check_kernel_regs.py): the shipped source
passes, and the checker does flag a compiler-generated use of the staging registers or a wrong allocation."""
import importlib.util
import os

from _common import ROOT


def _tool():
    spec = importlib.util.spec_from_file_location("check_kernel_regs", os.path.join(ROOT, "tools", "check_kernel_regs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_wgrad_staging_registers_are_left_alone_by_the_compiler():
    assert _tool().main.__call__ is not None
    import subprocess
    import sys
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_kernel_regs.py")], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout
    assert "contract holds for 2 instantiations" in res.stdout
    assert "no high-register splat in a packed subtraction" in res.stdout
    assert "accumulator registers untouched by the compiler (3 instantiations)" in res.stdout


_FAKE = """
_ZN5grafp15wgrad_gr_kernelINS_5WgCfgILi2EEEEEvPKt: ; @fake
\tv_mov_b32_e32 v3, v1
\t;;#ASMSTART
\tglobal_load_dwordx4 v[224:227], v[4:5], off
\t;;#ASMEND
{extra}
\ts_endpgm
  - .agpr_count:     {agpr}
    .name:           _ZN5grafp15wgrad_gr_kernelINS_5WgCfgILi2EEEEEvPKt
    .private_segment_fixed_size: 0
    .vgpr_count:     {vgpr}
    .vgpr_spill_count: 0
    .wavefront_size: 64
"""


def test_checker_flags_violations():
    tool = _tool()
    assert tool.check(_FAKE.format(extra="", agpr=0, vgpr=256)) == (1, [])
    _, errs = tool.check(_FAKE.format(extra="\tv_mov_b32_e32 v230, v1", agpr=0, vgpr=256))
    assert len(errs) == 1 and "v224+" in errs[0]
    _, errs = tool.check(_FAKE.format(extra="\tv_pk_add_f32 v[222:225], v[0:3], v[4:7]", agpr=0, vgpr=256))
    assert len(errs) == 1
    _, errs = tool.check(_FAKE.format(extra="", agpr=16, vgpr=224))
    assert len(errs) == 2
    assert tool.check("nothing here")[1]


def test_checker_flags_the_high_register_splat():
    tool = _tool()
    good = "_ZN5grafp19conv1x1_gemm_kernelIfoo: ; @x\n\tv_pk_add_f32 v[0:1], v[0:1], v[2:3] neg_lo:[0,1] neg_hi:[0,1]\n\ts_endpgm\n"
    bad = good.replace("v[2:3] neg_lo", "v[2:3] op_sel:[0,1] neg_lo")
    assert tool.check_gemm_splat(good) == (1, [])
    assert len(tool.check_gemm_splat(bad)[1]) == 1


_FAKE_XL = """
_ZN5grafp22conv1x1_gemm_xl_kernelILb1ELb0ELb0ELi0EEEvPKt: ; @fake
\t;;#ASMSTART
\tv_mfma_f32_32x32x16_bf16 a[0:15], v[2:5], v[6:9], a[0:15]
\t;;#ASMEND
\t;;#ASMSTART
\tv_accvgpr_read_b32 v3, a[4]
\t;;#ASMEND
{extra}
\ts_endpgm
  - .agpr_count:     {agpr}
    .name:           _ZN5grafp22conv1x1_gemm_xl_kernelILb1ELb0ELb0ELi0EEEvPKt
    .private_segment_fixed_size: 0
    .vgpr_count:     484
    .wavefront_size: 64
"""


def test_checker_guards_the_accumulator_file_of_the_four_wave_gemm():
    """gemm_xl.h names its accumulators a0-a255 in asm text: a compiler-generated v_accvgpr_* (a parked value, a moved
    placeholder), a scratch access or a descriptor without all 256 AGPRs must fail the build check; the shipped source
    passes (asserted through the tool's own run above: its output names the kernel)."""
    tool = _tool()
    assert tool.check_gemm_xl(_FAKE_XL.format(extra="", agpr=256)) == (1, [])
    assert len(tool.check_gemm_xl(_FAKE_XL.format(extra="\tv_accvgpr_write_b32 a17, v3", agpr=256))[1]) == 1
    assert len(tool.check_gemm_xl(_FAKE_XL.format(extra="\tscratch_store_dword off, v0, off", agpr=256))[1]) == 1
    assert len(tool.check_gemm_xl(_FAKE_XL.format(extra="", agpr=128))[1]) == 1
    assert tool.check_gemm_xl("nothing here")[1]


def test_no_matrix_kernel_takes_a_packed_low_lane_from_a_high_register():
    """The rule of tools/check_kernel_regs.py:check_mfma_packed_select over EVERY kernel source (the hazard measured in
    DESIGN.md section 12.7b): it holds for the shipped sources, and the checker does flag the form."""
    import subprocess
    import sys
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_kernel_regs.py"), "--all"], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout[-3000:]
    assert "no packed-f32 high-register select in any of the" in res.stdout
    tool = _tool()
    k = "foo_kernel: ; @foo_kernel\n\tv_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n{}\n\ts_endpgm\n\t.amdhsa_kernel foo_kernel\n"
    assert tool.check_mfma_packed_select(k.format("\tv_pk_mul_f32 v[0:1], v[0:1], v[2:3] op_sel_hi:[1,0]")) == (1, [])
    n, errs = tool.check_mfma_packed_select(k.format("\tv_pk_add_f32 v[0:1], v[0:1], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]"))
    assert n == 1 and len(errs) == 1
    nomfma = k.replace("\tv_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n", "")
    assert tool.check_mfma_packed_select(nomfma.format("\tv_pk_add_f32 v[0:1], v[0:1], v[2:3] op_sel:[0,1]")) == (0, [])
