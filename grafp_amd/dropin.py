"""Make the reference's scripts import this package's modules unchanged.

    import grafp_amd.dropin; grafp_amd.dropin.install()
    from encoder.graph_encoder import GraphEncoder      # -> grafp_amd.encoder.graph_encoder
    from simclr.simclr import SimCLR                     # -> grafp_amd.simclr.simclr
    from simclr.ntxent import ntxent_loss
    from modules.transformations import GPUTransformNeuralfp
    from eval import eval_faiss, get_index, load_memmap_data
    from test_fp import create_fp_db, create_dummy_db    # -> grafp_amd.fpdb
    from generate import create_db
    import faiss                                         # -> grafp_amd.faiss_standin (when no real faiss is installed)

These are the names train.py:16-23, generate.py:10-17 and test_fp.py:18-26 import.  With the reference's directory on
sys.path, `python train.py` / `test_fp.py` / `generate.py` then run as they are (tests/test_host_cpu.py imports them that
way); `util` and `modules.data` stay the reference's own host-side files there.
"""
import importlib
import importlib.util
import sys

_ALIASES = {
    "encoder": "grafp_amd.encoder",
    "encoder.graph_encoder": "grafp_amd.encoder.graph_encoder",
    "encoder.gcn_lib": "grafp_amd.encoder.gcn_lib",
    "encoder.gcn_lib.torch_nn": "grafp_amd.encoder.gcn_lib.torch_nn",
    "encoder.gcn_lib.torch_edge": "grafp_amd.encoder.gcn_lib.torch_edge",
    "encoder.gcn_lib.torch_vertex": "grafp_amd.encoder.gcn_lib.torch_vertex",
    "encoder.gcn_lib.pos_embed": "grafp_amd.encoder.gcn_lib.pos_embed",
    "simclr": "grafp_amd.simclr",
    "simclr.simclr": "grafp_amd.simclr.simclr",
    "simclr.ntxent": "grafp_amd.simclr.ntxent",
    "peak_extractor": "grafp_amd.peak_extractor",
    "modules": "grafp_amd.modules",
    "modules.transformations": "grafp_amd.modules.transformations",
    "eval": "grafp_amd.eval",
    "test_fp": "grafp_amd.fpdb",
    "generate": "grafp_amd.fpdb",
    "util": "grafp_amd.util",
}
# Host-side modules of the reference that this package only partly mirrors (dataset index / decoding helpers are out of
# scope): inside a checkout of the reference -- its directory on sys.path -- the scripts keep THEIR OWN file, elsewhere
# the alias applies.  `modules` is kept as the reference's package there too (modules/data.py is its dataset code), with
# only modules.transformations replaced.
_KEEP_OWN_IF_PRESENT = ("util", "modules")


def _real_module_on_path(name):
    """Is there a module `name` on sys.path that is not this package's alias?"""
    if name in sys.modules and not getattr(sys.modules[name], "__name__", "").startswith("grafp_amd"):
        return True
    try:
        spec = importlib.util.find_spec(name) if name not in sys.modules else None
    except (ImportError, ValueError):
        spec = None
    return spec is not None


def install(overwrite=False, faiss=True):
    """Register the aliases (and, unless a real faiss is importable, `faiss` = grafp_amd.faiss_standin: the reference's
    eval.py:3 and test_fp.py:7 import it at module level)."""
    done = []
    for alias, target in _ALIASES.items():
        if alias in sys.modules and not overwrite:
            continue
        if alias in _KEEP_OWN_IF_PRESENT:
            saved = sys.modules.pop(alias, None) if overwrite else None
            own = _real_module_on_path(alias)
            if saved is not None and not own:
                sys.modules[alias] = saved
            if own:
                continue
        sys.modules[alias] = importlib.import_module(target)
        parent, _, child = alias.rpartition(".")
        if parent and parent in sys.modules:          # `import modules.transformations` binds the attribute as well
            setattr(sys.modules[parent], child, sys.modules[alias])
        done.append(alias)
    if "modules.transformations" in sys.modules and "modules" not in sys.modules and _real_module_on_path("modules"):
        pkg = importlib.import_module("modules")        # the reference's own package, with our transformations inside
        setattr(pkg, "transformations", sys.modules["modules.transformations"])
    if faiss and "faiss" not in sys.modules:
        try:
            real = importlib.util.find_spec("faiss")
        except (ImportError, ValueError):
            real = None
        if real is None:
            sys.modules["faiss"] = importlib.import_module("grafp_amd.faiss_standin")
            done.append("faiss")
    return sorted(done)
