"""oracle/ -- CPU restatement of the GraFPrint hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import, call, link or
execute anything in this package; grafp_amd/ never does (tests/test_layout.py greps for it).

What is here and how each piece is pinned:

* model.py      functional torch-CPU restatement of peak extractor, k-NN graph, gather/MRConv,
                Grapher/FFN/Downsample, GraphEncoder, SimCLR, NT-Xent and one train step.
                PINNED: checked against tests/golden/*.npz, which tests/golden/make_golden.py
                produced by running the reference's own modules (imported from /root/reference).
* model.logmel  torchaudio==2.3.0 MelSpectrogram+AmplitudeToDB restated from its published
                definition (the dependency is not vendored in /root/reference and not installed).
                PARITY UNPINNED against torchaudio itself; cross-checked against an independent
                numpy rfft implementation (tests/test_oracle.py).
* csrc/*.c      plain-C restatements with a fully specified f32 arithmetic order (k-ordered fmaf
                chains, lowest-index tie-break) for the two index-valued results: k-NN edge indices
                and flat-L2 top-k ids.  The HIP kernels must match these BIT-EXACTLY.
                knn_graph.c PINNED to the reference's dense_knn_matrix on integer-valued
                (rounding-free, tie-free) goldens and, as neighbour SETS outside near-ties, on f32
                goldens.  flat_search.c restates faiss==1.7.2 IndexFlatL2 (not vendored, not
                installed): PARITY UNPINNED against faiss; checked against float64 exact search.
* retrieval.py  eval.py:170-332 (search + offset compensation + sequence rerank + hit rates)
                restated in numpy on top of flat_search.
"""
