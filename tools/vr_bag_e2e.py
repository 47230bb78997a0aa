#!/usr/bin/env python3
"""BASELINE config 4 as a whole VR BAG: a synthetic varres_metadata / varres_refinements pair (refinement grids
3x3 .. 50x50) through NativeVRProcessor.process_refinements (records resident in HBM, one D2H of corrected records)
and, for comparison, through the reference-shaped grid-by-grid loop (run_refinements, 50 000-node batches).
Wall clock includes H2D of the records and D2H of the corrected records."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bathymetric_gnn_amd import synthetic
from bathymetric_gnn_amd.data import GraphBuilder, VRBagHandler
from bathymetric_gnn_amd.models import BathymetricGNN
from bathymetric_gnn_amd.scripts.inference_native import NativeVRProcessor, run_refinements

ap = argparse.ArgumentParser()
ap.add_argument("--base", type=int, default=70, help="base grid is base x base cells (70 -> ~4 100 refinement grids)")
ap.add_argument("--budget", type=int, default=8 << 20)
ap.add_argument("--loop", action="store_true", help="also time the grid-by-grid loop")
args = ap.parse_args()
md, ref = synthetic.synthetic_vr_bag(args.base, args.base, seed=1000)
h = VRBagHandler.from_arrays(md, ref)
sd = synthetic.synthetic_state_dict(in_channels=8, seed=1234)
m = BathymetricGNN(in_channels=8, edge_dim=3, dropout=0.0); m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
proc = NativeVRProcessor(m.to("cuda:0").eval(), GraphBuilder(), torch.device("cuda:0"))
out = {"grids": h.num_refinement_cells, "cells": h.total_refinement_nodes}
small = VRBagHandler.from_arrays(*synthetic.synthetic_vr_bag(6, 6, seed=1))
proc.process_refinements(small, small.copy_and_open_for_writing(), 0.01)          # warm-up
walls = []
for rep in range(6):                                  # the first repetition still grows the pinned staging buffers
    w = h.copy_and_open_for_writing()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = proc.process_refinements(h, w, 0.01, cell_budget=args.budget)
    torch.cuda.synchronize(); walls.append(time.perf_counter() - t0)
dt = float(np.median(walls[1:]))
out["device_path"] = {"wall_s": dt, "wall_s_all": walls, "nodes_per_s": st["cells_processed"] / dt, "stats": st}
if args.loop:
    w2 = h.copy_and_open_for_writing()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st2 = run_refinements(proc, h, w2, 0.01)
    torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
    out["grid_loop"] = {"wall_s": dt2, "nodes_per_s": st2["cells_processed"] / dt2}
    out["records_equal"] = bool(np.array_equal(w.refinements.view(np.uint32), w2.refinements.view(np.uint32)))
print(json.dumps(out))
