"""CPU oracle for the Bathymetric-GNN hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and there only as the checker / the timed CPU baseline -- never as the
thing that is shipped.  The product path (``bathymetric_gnn_amd``) calls the
HIP library through the C ABI and raises if that library is missing.

Pinning status
--------------
* ``graph_cpu`` (grid -> graph, reference ``data/graph_construction.py``):
  PINNED.  Checked bit-for-bit (edge_index) / to float32 ulps (x, edge_attr)
  against golden vectors in ``tests/golden/`` that were produced by running the
  reference's own ``GraphBuilder`` in the build container
  (``tests/golden/make_golden.py``), plus the known answers in SURVEY.md
  Appendix A.
* tiling (reference ``data/tiling.py``) has no module here: the product's host mirror
  ``bathymetric_gnn_amd/data/tiling.py`` is itself checked bit-for-bit against
  ``tests/golden/tiling_reference.npz``, generated from the reference's ``TileManager`` /
  ``TileMerger`` (``tests/golden/make_golden_tiling.py``).
* ``gat_cpu`` (model forward, reference ``models/gnn.py``): **parity unpinned**.
  The arithmetic of GATConv / BatchNorm lives in ``torch_geometric``
  (un-vendored, version unpinned in the reference: ``environment.yml:50-52``,
  ``install.sh:58``), which is not installable here, and the reference's own
  tests assert no numerical value at this boundary
  (``scripts/test_pipeline.py:333-345`` prints shapes only).  ``gat_cpu``
  restates the published upstream semantics and is
  anchored on the reference's call sites (``models/gnn.py:125-132,176,181``)
  and on a hand-computed known-answer case in ``tests/test_oracle_gat.py``.  What the
  reference CAN pin of the model is pinned: ``LocalFeatureExtractor``, the three heads, the
  ``forward`` wiring / softmax / argmax and the ``predict`` flag logic are plain torch in
  ``models/gnn.py`` and were executed in the build container
  (``tests/golden/make_golden_model.py`` -> ``tests/golden/model_*.npz``); ``gat_cpu``
  reproduces those fixtures (``tests/test_oracle_model_golden.py``).  Only the GATConv
  arithmetic itself (a12) stays unpinned.
"""
