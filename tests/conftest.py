import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "statistical: scores briefly trained models (hit rates, free-running drift); "
                                       "collected LAST so that `-x` never hides a parity test behind it")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# Order of a GPU run: kernel parity -> matrix products -> model / training step -> headline sizes -> data parallel ->
# bf16 bars -> the statistical tests.  Under `-x` a failure then hides only what depends on what failed (one
# hit-rate assertion, collected first by file name, hid 324 parity tests).
_FILE_ORDER = ("test_gpu_kernels.py", "test_gpu_gemm.py", "test_gpu_model.py", "test_gpu_repro.py",
               "test_gpu_headline.py", "test_gpu_dist.py", "test_gpu_bf16_backward.py", "test_gpu_bf16.py")


def _rank(item):
    name = os.path.basename(str(item.fspath))
    file_rank = _FILE_ORDER.index(name) if name in _FILE_ORDER else -1        # CPU files keep their place in front
    return (1 if item.get_closest_marker("statistical") is not None else 0, file_rank)


def pytest_collection_modifyitems(config, items):
    config._grafp_gpu_selected = any(it.get_closest_marker("gpu") is not None for it in items)
    items.sort(key=_rank)              # stable: the order inside a file is the order of definition


@pytest.fixture(scope="session", autouse=True)
def _poisoned_allocator(request):
    """GPU runs: fill part of the caching allocator's blocks with NaNs once, so that the tensors the tests (and the ops'
    workspaces) allocate afterwards do not start out as zeros -- a kernel that reads what nobody wrote shows up instead of
    passing by luck on a fresh box.  (Round 3: the normalise-on-load race was noticed this way.)
    Only when GPU tests were selected (a CPU-only session on a GPU box never initialises the device), sized from the free
    memory (at most 8 GiB, at most a quarter of what is free), and an out-of-memory here is not an error of the tests."""
    if not getattr(request.config, "_grafp_gpu_selected", False):
        yield
        return
    try:
        import torch
        if torch.cuda.is_available():
            free, _total = torch.cuda.mem_get_info(0)
            n = min(1 << 31, int(free // 4) // 4)
            if n > 0:
                junk = torch.full((n,), float("nan"), device="cuda:0")
                del junk
    except Exception as exc:       # noqa: BLE001 -- a smaller or shared GPU: run the tests without the poison
        print(f"[conftest] allocator poisoning skipped: {type(exc).__name__}: {exc}")
    yield
