"""Make the reference's scripts import this package's modules unchanged.

    import grafp_amd.dropin; grafp_amd.dropin.install()
    from encoder.graph_encoder import GraphEncoder      # -> grafp_amd.encoder.graph_encoder
    from simclr.simclr import SimCLR                     # -> grafp_amd.simclr.simclr
    from simclr.ntxent import ntxent_loss
    from modules.transformations import GPUTransformNeuralfp
    from eval import eval_faiss, get_index, load_memmap_data
    from test_fp import create_fp_db, create_dummy_db    # -> grafp_amd.fpdb
    from generate import create_db

These are the names train.py:16-23, generate.py:10-17 and test_fp.py:18-26 import.
"""
import importlib
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


def install(overwrite=False):
    for alias, target in _ALIASES.items():
        if alias in sys.modules and not overwrite:
            continue
        sys.modules[alias] = importlib.import_module(target)
    return sorted(_ALIASES)
