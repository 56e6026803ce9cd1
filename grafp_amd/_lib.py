"""ctypes binding of libgrafp_hip.so (include/grafp_hip.h).

There is NO fallback: if the library is missing the import fails, and every op in grafp_amd.ops refuses
tensors that are not on a HIP device.  `python -c "import __graft_entry__ as g; g.build()"` (or
`make -C grafp_amd/csrc`) builds it for gfx950.
"""
import ctypes
import os
import subprocess

import torch  # noqa: F401  -- FIRST: the library must bind to the HIP runtime torch ships (same soname as /opt/rocm's);
#                               loaded the other way round the process holds two runtimes and launches find no device

_HERE = os.path.dirname(os.path.abspath(__file__))
# GRAFP_HIP_LIB: another build of the same ABI (the measurement build: make -C grafp_amd/csrc measure)
LIB_PATH = os.environ.get("GRAFP_HIP_LIB") or os.path.join(_HERE, "libgrafp_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "grafp_hip.h")

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_L = _c.c_int64
_Z = _c.c_size_t
_F = _c.c_float

# name -> (restype, argtypes); mirrors include/grafp_hip.h one to one (tests/test_abi.py cross-checks)
SIGNATURES = {
    "grafp_abi_version": (_I, []),
    "grafp_last_error": (_c.c_char_p, []),
    "grafp_logmel_f32": (_I, [_P, _L, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "grafp_unfold_segments_f32": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "grafp_peak_extract_fwd_f32": (_I, [_P, _I, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "grafp_peak_extract_bwd_f32": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "grafp_peak_extract_bwd_workspace": (_Z, [_I, _I, _I, _I]),
    "grafp_knn_graph_workspace": (_Z, [_I, _I, _I]),
    "grafp_knn_normalize_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "grafp_knn_topk_f32": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "grafp_knn_topk_i32": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "grafp_knn_normalize_strided": (_I, [_P, _I, _L, _L, _I, _I, _I, _I, _P, _P, _P]),
    "grafp_knn_graph_f32": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _Z, _P]),
    "grafp_knn_pre_supported": (_I, [_I, _I, _I]),
    "grafp_knn_pre_workspace": (_Z, [_I, _I, _I]),
    "grafp_knn_graph_pre": (_I, [_P, _I, _L, _L, _I, _I, _I, _I, _I, _P, _I, _P, _Z, _P]),
    "grafp_knn_split_supported": (_I, [_I, _I, _I]),
    "grafp_knn_split_preferred": (_I, [_I, _I, _I]),
    "grafp_knn_split_workspace": (_Z, [_I, _I, _I]),
    "grafp_knn_split_preferred_for": (_I, [_I, _I, _I, _I]),
    "grafp_knn_split_workspace_for": (_Z, [_I, _I, _I, _I]),
    "grafp_knn_graph_split": (_I, [_P, _I, _L, _L, _I, _I, _I, _I, _P, _I, _P, _Z, _P, _P]),
    "grafp_mrconv_fwd_f32": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "grafp_mrconv_bwd_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "grafp_mrconv_fwd_strided": (_I, [_P, _I, _L, _L, _P, _I, _I, _I, _I, _P, _L, _L, _P]),
    "grafp_mrconv_bwd_strided": (_I, [_P, _I, _L, _L, _P, _P, _L, _L, _I, _I, _I, _I, _P, _P]),
    "grafp_mrconv_fwd_strided_i32": (_I, [_P, _I, _L, _L, _P, _I, _I, _I, _I, _P, _L, _L, _P]),
    "grafp_mrconv_bwd_strided_i32": (_I, [_P, _I, _L, _L, _P, _P, _L, _L, _I, _I, _I, _I, _P, _P]),
    "grafp_mrconv_arg_supported": (_I, [_I, _L, _L, _L, _L, _I, _I]),
    "grafp_mrconv_fwd_arg": (_I, [_P, _I, _L, _L, _P, _I, _I, _I, _I, _I, _P, _L, _L, _P, _P]),
    "grafp_mrconv_bwd_arg": (_I, [_P, _I, _P, _I, _P, _L, _L, _I, _I, _I, _I, _P, _L, _L, _P]),
    "grafp_stride2_taps_fwd": (_I, [_P, _I, _L, _I, _P, _P]),
    "grafp_stride2_taps_bwd": (_I, [_P, _I, _L, _I, _P, _P]),
    "grafp_bn_workspace": (_Z, [_I, _L]),
    "grafp_bn_fwd": (_I, [_P, _I, _I, _L, _I, _P, _P, _P, _P, _I, _F, _F, _F, _I, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "grafp_bn_bwd": (_I, [_P, _P, _I, _I, _L, _I, _P, _P, _P, _P, _P, _I, _F, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "grafp_bn_sync_bytes": (_Z, [_I, _L]),
    "grafp_bn_fwd_1pass": (_I, [_P, _I, _I, _L, _I, _P, _P, _P, _P, _I, _F, _F, _F, _I, _P, _P, _P, _P, _P, _P, _Z, _P, _I, _P]),
    "grafp_bn_bwd_1pass": (_I, [_P, _P, _I, _I, _L, _I, _P, _P, _P, _P, _P, _I, _F, _I, _P, _P, _P, _P, _P, _Z, _P, _I, _P]),
    "grafp_ivfpq_scan_f32": (_I, [_P, _I, _I, _P, _I, _P, _I, _P, _P, _P, _I, _P, _L, _P, _P, _P]),
    "grafp_pq_assign_f32": (_I, [_P, _L, _I, _I, _P, _P, _P, _I, _P, _P, _P]),
    "grafp_kmeans_workspace": (_Z, [_L, _I, _I, _I]),
    "grafp_kmeans_f32": (_I, [_P, _L, _I, _I, _P, _P, _P, _I, _I, _P, _P, _Z, _P]),
    "grafp_ivfpq_probe_f32": (_I, [_P, _I, _I, _P, _I, _I, _P, _P]),
    "grafp_ivfpq_search_f32": (_I, [_P, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "grafp_debug_occupy": (_I, [_I, _I, _L, _P]),
    "grafp_conv1x1_gemm_supported": (_I, [_I, _I, _I, _L, _I]),
    "grafp_conv1x1_gemm_partials": (_I, [_I, _I, _I, _L, _I]),
    "grafp_conv1x1_gemm_plan": (_I, [_I, _I, _I, _L, _I, _P]),
    "grafp_conv1x1_gemm_bf16": (_I, [_P, _P, _I, _I, _I, _L, _I, _P, _I, _F, _P, _P, _P]),
    "grafp_conv1x1_gemm_affine_bf16": (_I, [_P, _P, _I, _I, _I, _L, _I, _P, _I, _F, _P, _P]),
    "grafp_conv1x1_gemm_cat_bf16": (_I, [_P, _P, _I, _P, _I, _I, _L, _P, _P]),
    "grafp_split_bf16_planes": (_I, [_P, _L, _P, _P, _P]),
    "grafp_conv1x1_gemm_split_f32": (_I, [_P, _P, _I, _I, _L, _P, _P]),
    "grafp_weights_prepare": (_I, [_P, _P, _I, _P]),
    "grafp_bn_finalize": (_I, [_P, _I, _I, _I, _L, _I, _P, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _P]),
    "grafp_bn_affine_bf16": (_I, [_P, _I, _L, _I, _P, _P, _I, _F, _P, _P]),
    "grafp_bn_finalize_affine_bf16": (_I, [_P, _P, _I, _I, _I, _L, _I, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _I, _F, _P, _P]),
    "grafp_conv1x1_wgrad_workspace": (_Z, [_I, _I, _I, _L]),
    "grafp_conv1x1_wgrad_bf16": (_I, [_P, _P, _I, _I, _I, _L, _P, _P, _Z, _P]),
    "grafp_conv1x1_wgrad_pro_workspace": (_Z, [_I, _I, _I, _L, _I]),
    "grafp_conv1x1_wgrad_plan": (_I, [_I, _I, _I, _L, _I, _P]),
    "grafp_conv1x1_wgrad_pro_bf16": (_I, [_P, _P, _I, _I, _I, _L, _I, _P, _I, _F, _P, _P, _Z, _P]),
    "grafp_conv1x1_wgrad_tile_workspace": (_Z, [_I, _I, _I, _L, _I, _I]),
    "grafp_conv1x1_wgrad_tile_bf16": (_I, [_P, _P, _I, _I, _I, _L, _I, _P, _I, _F, _I, _P, _P, _Z, _P]),
    "grafp_conv1x1_wgrad_partials_bf16": (_I, [_P, _P, _I, _I, _I, _L, _I, _P, _I, _F, _I, _P, _Z, _P, _P]),
    "grafp_wgrad_reduce_multi": (_I, [_P, _P, _P, _P, _I, _P]),
    "grafp_adam_multi_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _P]),
    "grafp_conv1x1_wgrad_f32_workspace": (_Z, [_I, _I, _I, _L]),
    "grafp_conv1x1_wgrad_f32": (_I, [_P, _P, _I, _I, _I, _L, _P, _P, _Z, _P]),
    "grafp_ntxent_workspace": (_Z, [_I]),
    "grafp_ntxent_num_partials": (_I, [_I]),
    "grafp_ntxent_fwd_bwd_f32": (_I, [_P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P, _Z, _P]),
    "grafp_row_sqnorm_f32": (_I, [_P, _L, _I, _P, _P]),
    "grafp_knn_search_workspace": (_Z, [_L, _I, _I, _I]),
    "grafp_knn_search_l2_f32": (_I, [_P, _P, _L, _P, _I, _I, _I, _L, _P, _P, _P, _Z, _P]),
    "grafp_f32_to_bf16": (_I, [_P, _L, _P, _P]),
    "grafp_knn_search_pre_workspace": (_Z, [_L, _I, _I, _I]),
    "grafp_knn_search_l2_pre": (_I, [_P, _P, _P, _L, _P, _I, _I, _I, _L, _P, _P, _P, _Z, _P]),
    "grafp_merge_topk": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "grafp_seq_rerank_f32": (_I, [_P, _L, _P, _L, _P, _I, _P, _P, _I, _I, _I, _P, _P, _P]),
    "grafp_seq_rerank_shard_f32": (_I, [_P, _L, _L, _L, _L, _L, _P, _L, _P, _I, _P, _P, _I, _I, _I, _P, _P, _P]),
    "grafp_ir_convolve_f32": (_I, [_P, _L, _I, _I, _P, _P, _I, _P, _P, _P, _L, _P]),
    "grafp_mix_snr_workspace": (_Z, [_I, _I]),
    "grafp_mix_snr_f32": (_I, [_P, _L, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _L, _P, _Z, _P]),
}


def build(verbose=False):
    """Compile libgrafp_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j", str(min(8, os.cpu_count() or 1))]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libgrafp_hip.so failed (see output above)")
    return LIB_PATH


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the GraFPrint hot path is hand-written HIP and has no CPU/eager "
            "fallback.  Build it with `make -C grafp_amd/csrc` (hipcc, --offload-arch=gfx950).")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export the symbol
        fn.restype = res
        fn.argtypes = args
    got = lib.grafp_abi_version()
    if got != 1:
        raise ImportError(f"libgrafp_hip.so ABI version {got} != 1: rebuild with `make -C grafp_amd/csrc`")
    return lib


lib = _load()


def check(rc, what=""):
    if rc != 0:
        msg = lib.grafp_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libgrafp_hip {what} failed (code {rc}): {msg}")
