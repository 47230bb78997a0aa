"""Tiling row: our TileManager / TileMerger / BathymetricGrid against golden vectors
produced by the reference's own data/tiling.py + data/loaders.py (tests/golden/make_golden_tiling.py)."""
import os

import numpy as np
import pytest

from bathymetric_gnn_amd.data import BathymetricGrid, TileManager, TileMerger, TileSpec

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "tiling_reference.npz"))


def _specs(specs):
    return np.array([[s.row_start, s.col_start, s.row_end, s.col_end, s.tile_row, s.tile_col] for s in specs], np.int64)


@pytest.mark.parametrize("i", range(8))
def test_compute_tile_grid(i):
    h, w, ts, ov = (int(v) for v in Z[f"grid{i}_in"])
    nr, nc, specs = TileManager(ts, ov).compute_tile_grid((h, w))
    assert (nr, nc) == tuple(int(v) for v in Z[f"grid{i}_n"])
    sa = _specs(specs)
    assert np.array_equal(sa.sum(0), Z[f"grid{i}_specs_sum"])
    exp = Z[f"grid{i}_specs"]
    got = sa if len(sa) <= 64 else np.concatenate([sa[:32], sa[-32:]])
    assert np.array_equal(got, exp)


def test_survey_appendix_a4():
    _, _, specs = TileManager(512, 128).compute_tile_grid((1300, 900))
    cols = sorted({(s.col_start, s.col_end) for s in specs})
    assert cols == [(0, 512), (384, 896), (388, 900)]
    nr, nc, _ = TileManager(512, 128).compute_tile_grid((60000, 60000))
    assert (nr, nc) == (156, 156)
    b = TileManager(1024, 128)._create_1d_blend(16)
    np.testing.assert_allclose(b[:4], [0, .25, .75, 1], atol=1e-7)
    np.testing.assert_allclose(b[-4:], [1, .75, .25, 0], atol=1e-7)


@pytest.mark.parametrize("size,ov", [(16, 128), (512, 128), (256, 64), (7, 128), (100, 10), (3, 1)])
def test_blend_weights_bit_exact(size, ov):
    got = TileManager(1024, ov)._create_1d_blend(size)
    exp = Z[f"blend_{size}_{ov}"]
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), exp.view(np.uint32))


def test_merge_matches_reference():
    specs = [TileSpec(*[int(v) for v in row]) for row in Z["merge_specs"]]
    tm = TileManager(64, 16, 0.1)
    merger = TileMerger(tm)
    H = max(s.row_end for s in specs); W = max(s.col_end for s in specs)
    merger.initialize((H, W), ["cleaned_depth", "classification", "confidence", "correction"])
    for k, s in enumerate(specs):
        if f"merge_tile{k}_confidence" not in Z.files:
            continue
        merger.add_tile(s, {ch: Z[f"merge_tile{k}_{ch}"].copy() for ch in
                            ("cleaned_depth", "classification", "confidence", "correction")})
    res = merger.finalize()
    for ch, v in res.items():
        exp = Z[f"merge_out_{ch}"]
        assert np.array_equal(np.isnan(v), np.isnan(exp)), ch
        assert np.array_equal(np.nan_to_num(v).view(np.uint32), np.nan_to_num(exp).view(np.uint32)), ch
    with pytest.raises(ValueError):
        m2 = TileMerger(tm); m2.initialize((4, 4), ["a"]); m2.add_tile(specs[0], {"b": np.zeros((1, 1))})


def test_grid_valid_mask_and_iterate():
    g = BathymetricGrid(depth=Z["grid_valid_in"], nodata_value=1.0e6)
    assert np.array_equal(g.valid_mask, Z["grid_valid_mask"])
    rng = np.random.default_rng(0)
    d = rng.standard_normal((100, 90)).astype(np.float32)
    d[:40, :50] = 1.0e6
    grid = BathymetricGrid(depth=d, uncertainty=np.ones_like(d), resolution=(0.5, 0.5))
    tm = TileManager(32, 8, min_valid_ratio=0.5)
    tiles = list(tm.iterate_tiles(grid))
    _, _, specs = tm.compute_tile_grid(grid.shape)
    assert 0 < len(tiles) < len(specs)
    for t in tiles:
        assert t.valid_ratio >= 0.5 and t.data.shape == t.valid_mask.shape == t.uncertainty.shape
        assert np.array_equal(t.data, d[t.row_start:t.row_end, t.col_start:t.col_end])
    with pytest.raises(ValueError):
        TileManager(16, 16)
