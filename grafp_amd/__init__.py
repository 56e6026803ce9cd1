"""grafp_amd -- MI355X-native hot path of GraFPrint (contrastive step + fingerprint retrieval).

Module paths mirror the reference so its scripts can import this package's modules unchanged after
`grafp_amd.dropin.install()`:  encoder.graph_encoder, encoder.gcn_lib.*, simclr.simclr, simclr.ntxent,
peak_extractor, modules.transformations, eval, test_fp (create_*_db), generate (create_db), util.
"""
__version__ = "0.1.0"
