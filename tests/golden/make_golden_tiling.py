#!/usr/bin/env python3
"""Golden vectors for the tiling row (reference data/tiling.py), produced by the REFERENCE's own
TileManager / TileMerger in the build container:

    python tests/golden/make_golden_tiling.py

The reference's ``data`` package is not imported as a package (its __init__ pulls in GDAL / PyG
modules); ``data/loaders.py`` (dataclass only; GDAL optional) and ``data/tiling.py`` are loaded as
submodules of an empty stand-in package.  Fixtures hold inputs and expected outputs only.
"""
import importlib
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
pkg = types.ModuleType("refdata")
pkg.__path__ = ["/root/reference/data"]
sys.modules["refdata"] = pkg
tiling = importlib.import_module("refdata.tiling")
loaders = importlib.import_module("refdata.loaders")


def specs_array(specs):
    return np.array([[s.row_start, s.col_start, s.row_end, s.col_end, s.tile_row, s.tile_col] for s in specs], np.int64)


def main():
    out = {}
    cases = [((1300, 900), 512, 128), ((1300, 900), 256, 64), ((300, 300), 512, 128), ((512, 512), 512, 128),
             ((513, 1000), 256, 32), ((60000, 60000), 512, 128), ((100, 37), 64, 16), ((640, 640), 256, 0)]
    for i, (shape, ts, ov) in enumerate(cases):
        tm = tiling.TileManager(ts, ov)
        nr, nc, specs = tm.compute_tile_grid(shape)
        out[f"grid{i}_in"] = np.array([shape[0], shape[1], ts, ov], np.int64)
        out[f"grid{i}_n"] = np.array([nr, nc], np.int64)
        sa = specs_array(specs)
        out[f"grid{i}_specs"] = sa if len(sa) <= 64 else np.concatenate([sa[:32], sa[-32:]])
        out[f"grid{i}_specs_sum"] = sa.sum(0)
    for size, ov in [(16, 128), (512, 128), (256, 64), (7, 128), (100, 10), (3, 1)]:
        out[f"blend_{size}_{ov}"] = tiling.TileManager(max(size, 2 * ov + 1), ov)._create_1d_blend(size) \
            if False else tiling.TileManager(1024, ov)._create_1d_blend(size)
    # merged example: 150x130 grid, tile 64 / overlap 16, random per-tile results with NaN holes
    rng = np.random.default_rng(11)
    H, W = 150, 130
    tm = tiling.TileManager(64, 16, 0.1)
    _, _, specs = tm.compute_tile_grid((H, W))
    merger = tiling.TileMerger(tm)
    merger.initialize((H, W), ["cleaned_depth", "classification", "confidence", "correction"])
    tiles_in = {}
    for k, s in enumerate(specs):
        h, w = s.row_end - s.row_start, s.col_end - s.col_start
        data = {
            "cleaned_depth": (-20 + rng.standard_normal((h, w))).astype(np.float32),
            "classification": rng.integers(0, 3, (h, w)).astype(np.float32),
            "confidence": rng.random((h, w)).astype(np.float32),
            "correction": (0.1 * rng.standard_normal((h, w))).astype(np.float32),
        }
        if k % 3 == 0:
            data["cleaned_depth"][rng.random((h, w)) < 0.05] = np.nan
        if k == 4:
            continue      # a skipped tile
        for ch, v in data.items():
            tiles_in[f"merge_tile{k}_{ch}"] = v
        merger.add_tile(s, data)
    res = merger.finalize()
    out.update(tiles_in)
    out["merge_specs"] = specs_array(specs)
    for ch, v in res.items():
        out[f"merge_out_{ch}"] = v
    # BathymetricGrid.valid_mask (data/loaders.py:59)
    d = np.array([[1.0, np.nan, 1.0e6], [np.inf, -3.0, 0.0]], np.float32)
    g = loaders.BathymetricGrid(depth=d, uncertainty=None, nodata_value=1.0e6, transform=None, crs=None,
                                resolution=(1.0, 1.0), bounds=None, source_path=None)
    out["grid_valid_in"] = d
    out["grid_valid_mask"] = g.valid_mask
    np.savez_compressed(os.path.join(HERE, "tiling_reference.npz"), **out)
    print("wrote tiling_reference.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
