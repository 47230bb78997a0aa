#!/usr/bin/env python3
"""BASELINE config 5, scaled to one GPU and host RAM: a synthetic S x S survey @0.5 m through
BathymetricPipeline.process_grid (overlapping 512 x 512 tiles, overlap 128, Hann-ramp stitch, corrections).
Prints wall time per phase; the GPU share is what bench.py measures, the rest is host numpy (tile copies,
pageable H2D / D2H, TileMerger) -- the next thing to move to the device."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bathymetric_gnn_amd import synthetic
from bathymetric_gnn_amd.config import Config
from bathymetric_gnn_amd.data import BathymetricGrid
from bathymetric_gnn_amd.models import BathymetricGNN, BathymetricPipeline

ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=6000); ap.add_argument("--tile-batch", type=int, default=32)
args = ap.parse_args()
S = args.size
rng = np.random.default_rng(0)
r = np.arange(S, dtype=np.float32)[:, None]; c = np.arange(S, dtype=np.float32)[None, :]
depth = (-20 - 0.01 * c - 0.005 * r + 0.5 * np.sin(2 * np.pi * r / 37) * np.cos(2 * np.pi * c / 53)).astype(np.float32)
depth += 0.05 * rng.standard_normal((S, S), dtype=np.float32)
depth[: S // 10, : S // 8] = 1.0e6
grid = BathymetricGrid(depth=depth, nodata_value=1.0e6, resolution=(0.5, 0.5))
cfg = Config(); cfg.tile.tile_size, cfg.tile.overlap = 512, 128
pipe = BathymetricPipeline(cfg, tile_batch=args.tile_batch)
sd = synthetic.synthetic_state_dict(seed=1234)
m = BathymetricGNN(in_channels=7, edge_dim=3, dropout=0.0); m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
pipe.set_model(m)
_, _, specs = pipe.tile_manager.compute_tile_grid(grid.shape)
evals = sum((s.row_end - s.row_start) * (s.col_end - s.col_start) for s in specs)
out = {"survey": f"{S}x{S}", "tiles": len(specs), "node_evals": evals, "valid_cells": int(grid.valid_mask.sum())}
pipe.process_grid_device(BathymetricGrid(depth=depth[:1024, :1024].copy(), nodata_value=1.0e6, resolution=(0.5, 0.5)))  # warm-up
for name, host in (("device_stitch", False), ("host_stitch", True)):
    pipe.host_stitch = host
    torch.cuda.synchronize(); t0 = time.perf_counter(); res = pipe.process_grid(grid); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out[name] = {"wall_s": dt, "node_evals_per_s": evals / dt}
print(json.dumps(out))
