"""CPU tests of the drop-in boundary: include/grafp_hip.h <-> libgrafp_hip.so <-> ctypes table.
No compute call is made (there is no GPU here)."""
import ctypes
import os
import re

import pytest

from _common import ROOT

HEADER = os.path.join(ROOT, "include", "grafp_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int|size_t|const char \*)\s*(grafp_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(3).replace("\n", " ").split(",")]
        decls[m.group(2)] = (m.group(1).strip(), [] if args == ["void"] else args)
    return decls


def test_header_declares_the_whole_path():
    d = _declared()
    for name in ("grafp_logmel_f32", "grafp_unfold_segments_f32", "grafp_peak_extract_fwd_f32",
                 "grafp_peak_extract_bwd_f32", "grafp_knn_graph_f32", "grafp_knn_normalize_f32", "grafp_knn_topk_f32",
                 "grafp_mrconv_fwd_f32", "grafp_mrconv_bwd_f32", "grafp_ntxent_fwd_bwd_f32", "grafp_row_sqnorm_f32",
                 "grafp_knn_search_l2_f32", "grafp_merge_topk", "grafp_last_error", "grafp_abi_version"):
        assert name in d, name
    # every compute entry takes a stream last and returns an int status
    for name, (ret, args) in d.items():
        if name.endswith("_f32") or name == "grafp_merge_topk":
            assert ret == "int" and args[-1].startswith("grafp_stream_t"), name
    # the boundary carries no torch / C++ types
    code = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    assert "torch" not in code.lower() and "std::" not in code and "hip/" not in code
    assert 'extern "C"' in code


def test_library_exports_every_declared_symbol():
    from grafp_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(raw, name), f"{name} declared in grafp_hip.h but not exported"
    assert raw.grafp_abi_version() == 1


def _ctype_of(arg):
    arg = arg.strip()
    if "*" in arg or arg.startswith("grafp_stream_t"):
        return ctypes.c_void_p
    base = arg.rsplit(" ", 1)[0].replace("const ", "").strip()
    return {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t, "float": ctypes.c_float,
            "int32_t": ctypes.c_int32, "double": ctypes.c_double}[base]


def test_ctypes_table_matches_header():
    from grafp_amd import _lib
    decl = _declared()
    assert set(decl) == set(_lib.SIGNATURES), set(decl) ^ set(_lib.SIGNATURES)
    for name, (ret, args) in decl.items():
        res, argtypes = _lib.SIGNATURES[name]
        assert [_ctype_of(a) for a in args] == list(argtypes), name
        want = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "const char *": ctypes.c_char_p}[ret]
        assert res is want, name


def test_argument_errors_are_reported_without_a_gpu():
    from grafp_amd._lib import check, lib
    rc = lib.grafp_knn_graph_f32(None, 1, 1, 1, 1, 1, None, None, 0, None)
    assert rc == -1 and b"null pointer" in lib.grafp_last_error()
    with pytest.raises(RuntimeError, match="null pointer"):
        check(rc, "knn_graph")
    assert lib.grafp_knn_graph_workspace(256, 64, 1024) >= 256 * 64 * 1024 * 4 + 256 * 1024 * 4
    assert lib.grafp_ntxent_num_partials(256) == 16 and lib.grafp_ntxent_workspace(256) == (2 * 512 + 2 * 512 * 3 + 2 * 512 * 128) * 4
    assert lib.grafp_knn_search_workspace(1_000_000, 41, 128, 20) > 0
    assert lib.grafp_knn_search_workspace(1_000_000, 41, 64, 20) == 0          # only 128-d fingerprints


def test_ops_refuse_cpu_tensors():
    import torch
    from grafp_amd import ops
    x = torch.zeros(1, 8, 16)
    idx = torch.zeros(1, 16, 3, dtype=torch.int64)
    for call in (lambda: ops.knn_graph(x, 3), lambda: ops.max_relative(x, idx), lambda: ops.logmel(torch.zeros(1, 16000)),
                 lambda: ops.ntxent(torch.zeros(4, 128), torch.zeros(4, 128), 0.05),
                 lambda: ops.peak_extract(torch.zeros(1, 64, 32), torch.zeros(8, 3, 7, 7), torch.zeros(8), 2),
                 lambda: ops.search_l2(torch.zeros(8, 128), torch.zeros(8), torch.zeros(1, 128), 1)):
        with pytest.raises(RuntimeError, match="no CPU"):
            call()
