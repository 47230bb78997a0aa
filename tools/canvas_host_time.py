#!/usr/bin/env python3
"""Host time of bgnn_graph_build for ragged batches (the shelf-packing search over <= 9 canvas widths runs on the host): wall time
of the call itself (kernels are only enqueued) for 68-grid (50 000-node) and 4096-grid batches, canvas on / off."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bathymetric_gnn_amd import runtime as rt, synthetic          # noqa: E402
from bathymetric_gnn_amd.data import GraphBuilder                 # noqa: E402

dev = torch.device("cuda:0")
ctx = rt.get_context(dev)
gb = GraphBuilder(device=dev)
for n_grids in (68, 700, 4096):
    grids = synthetic.vr_grid_stream(n_grids, seed0=1000)
    masks = [(d != synthetic.NODATA) & np.isfinite(d) for d, _, _ in grids]
    hw, res, d_t, m_t, u_t = gb.upload_tiles([g[0] for g in grids], masks, [g[1] for g in grids], [g[2] for g in grids])
    for atlas in (1, 0):
        ctx.set_option("ragged_atlas", atlas)
        ts = []
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g = gb.build_from_device(hw, res, d_t, m_t, u_t)
            ts.append((time.perf_counter() - t0) * 1e6)
            torch.cuda.synchronize()
            del g
        print(f"{n_grids} grids ({int(sum(int(m.sum()) for m in masks))} nodes), ragged_atlas={atlas}: host time of build_from_device "
              f"min {min(ts[1:]):.0f} us, median {sorted(ts[1:])[len(ts[1:]) // 2]:.0f} us")
    ctx.set_option("ragged_atlas", 1)
