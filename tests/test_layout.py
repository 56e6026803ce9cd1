"""Repository-level rules: the product never touches oracle/ or the reference; nothing is copied."""
import os
import re

from _common import ROOT


def _py_files(sub):
    for d, _, fs in os.walk(os.path.join(ROOT, sub)):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                yield os.path.join(d, f)


def test_product_never_imports_the_oracle_or_reads_the_reference():
    for path in _py_files("grafp_amd"):
        src = open(path).read()
        code = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith(("#", "//", "*", "/*")))
        assert not re.search(r"^\s*(from|import)\s+oracle\b", code, flags=re.M), path
        assert "liboracle" not in code, path
        assert not re.search(r"open\([^)]*/root/reference", code), path
        assert "sys.path" not in code or "/root/reference" not in code, path


def test_only_allowed_callers_import_the_oracle():
    allowed = {"bench.py", "__graft_entry__.py"}
    for f in os.listdir(ROOT):
        if f.endswith(".py") and f not in allowed:
            assert not re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(ROOT, f)).read(), flags=re.M), f


def test_oracle_header_says_test_infrastructure():
    assert "TEST INFRASTRUCTURE" in open(os.path.join(ROOT, "oracle", "__init__.py")).read()
    for f in ("model.py", "native.py", "retrieval.py", "csrc/knn_graph.c", "csrc/flat_search.c"):
        assert "TEST INFRASTRUCTURE" in open(os.path.join(ROOT, "oracle", f)).read(), f


def test_no_compat_layers():
    for path in _py_files("grafp_amd/csrc"):
        src = open(path).read()
        assert "__HIP_PLATFORM_AMD__" not in src and "hipify" not in src.lower() and "cuda_runtime" not in src, path


def test_required_layout():
    for p in ("bench.py", "__graft_entry__.py", "DESIGN.md", "INTEGRATION.md", "include/grafp_hip.h", "oracle/Makefile",
              "tests/golden/make_golden.py", "profiles"):
        assert os.path.exists(os.path.join(ROOT, p)), p
