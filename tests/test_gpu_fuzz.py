"""Seeded differential fuzz, GPU path vs the CPU oracle: many small random tiles with the shapes and defects the fixed
parametrisations do not enumerate -- degenerate extents (2 x N, N x 2), mask densities from empty to full, isolated
cells, anisotropic resolutions, NaN / inf / huge depths at valid cells, 4- and 8-connectivity, explicit self loops,
with and without the uncertainty channel.  Graph build: edge_index bit-exact, features within the bars of
test_gpu_graph.py.  Forward: logits within 1e-4 (north_star), ragged batches equal to the per-graph results bit for bit."""
import numpy as np
import pytest
import torch

from conftest import ulp_diff_f32
from oracle import gat_cpu, graph_cpu

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _random_tile(rng, allow_defects=True):
    h, w = int(rng.integers(2, 41)), int(rng.integers(2, 41))
    if rng.random() < 0.2:
        h = 2
    if rng.random() < 0.2:
        w = 2
    r = np.arange(h, dtype=np.float64)[:, None]; c = np.arange(w, dtype=np.float64)[None, :]
    depth = (-rng.uniform(5, 200) - rng.uniform(0, 0.05) * c - rng.uniform(0, 0.05) * r
             + rng.uniform(0, 2) * np.sin(r / rng.uniform(2, 9)) * np.cos(c / rng.uniform(2, 9))
             + rng.uniform(0, 0.3) * rng.standard_normal((h, w))).astype(np.float32)
    density = rng.choice([0.0, 0.03, 0.3, 0.6, 0.9, 1.0])
    mask = rng.random((h, w)) >= density
    if rng.random() < 0.1:
        mask[:] = False; mask[rng.integers(h), rng.integers(w)] = True            # one isolated cell
    depth = np.where(mask, depth, np.float32(1.0e6)).astype(np.float32)
    if allow_defects and rng.random() < 0.3 and mask.any():                        # defects AT valid cells
        rr, cc = np.nonzero(mask)
        for k in rng.integers(0, len(rr), size=min(3, len(rr))):
            depth[rr[k], cc[k]] = rng.choice([np.nan, np.inf, -np.inf, 3.0e38, -1.0e30, 0.0])
    unc = rng.uniform(0.01, 0.5, (h, w)).astype(np.float32) if rng.random() < 0.5 else None
    res = (float(rng.choice([0.25, 0.5, 1.0, 2.0, 3.7])), float(rng.choice([0.25, 0.5, 1.0, 2.0, 3.7])))
    return depth, mask, unc, res


def _check_graph(g, o, names):
    assert g.num_nodes == o.num_nodes
    if o.num_nodes == 0:
        return
    assert g.num_edges == o.num_edges
    assert np.array_equal(g.edge_index.cpu().numpy(), o.edge_index)
    x, ea = g.x.cpu().numpy(), g.edge_attr.cpu().numpy()
    assert x.shape == o.x.shape and ea.shape == o.edge_attr.shape
    u = ulp_diff_f32(x, o.x)
    for col in range(o.x.shape[1]):
        assert u[:, col].max(initial=0) <= (1 if names[col] == "local_std" else 0), (names[col], u[:, col].max())
    ue = ulp_diff_f32(ea, o.edge_attr)
    assert ue[:, 0].max(initial=0) == 0 and ue[:, 1].max(initial=0) == 0 and ue[:, 2].max(initial=0) <= 1
    assert np.array_equal(g.pos.cpu().numpy(), o.pos)
    assert np.array_equal(g.valid_rows.cpu().numpy(), o.valid_rows) and np.array_equal(g.valid_cols.cpu().numpy(), o.valid_cols)
    assert ulp_diff_f32(g.local_std.cpu().numpy(), o.local_std).max(initial=0) <= 1


@pytest.mark.parametrize("conn,loops", [("8-connected", False), ("4-connected", False), ("8-connected", True)])
def test_fuzz_graph_build(conn, loops, gpu_device):
    from bathymetric_gnn_amd.data import GraphBuilder
    rng = np.random.default_rng(20260 + len(conn) + int(loops))
    gb = GraphBuilder(connectivity=conn, include_self_loops=loops)
    n_nonempty = 0
    for case in range(60):
        d, m, u, res = _random_tile(rng)
        o = graph_cpu.build_graph(d, m, u, res, connectivity=conn, include_self_loops=loops)
        g = gb.build_graph(d, m, u, res)
        names = list(graph_cpu.DEFAULT_NODE_FEATURES) + (["uncertainty"] if u is not None else [])
        try:
            _check_graph(g, o, names)
        except AssertionError as e:
            raise AssertionError(f"case {case}: shape {d.shape}, valid {int(m.sum())}, res {res}, unc {u is not None}: {e}") from e
        n_nonempty += o.num_nodes > 0
    assert n_nonempty > 30


@pytest.mark.parametrize("conn", ["8-connected", "4-connected", "16-dilated"])
def test_fuzz_forward_and_ragged_batches(conn, gpu_device):
    """(Every stencil: the fused kernels rebuild the edge attributes from the compact storage -- unit and doubled (dilated) lengths at
    anisotropic, non-power-of-two resolutions, per-grid lengths on the ragged canvas.)"""
    from bathymetric_gnn_amd import synthetic
    from bathymetric_gnn_amd.data import GraphBuilder
    from bathymetric_gnn_amd.models import BathymetricGNN
    rng = np.random.default_rng(777)
    sd = synthetic.synthetic_state_dict(in_channels=7, seed=99)
    model = BathymetricGNN(in_channels=7, edge_dim=3, dropout=0.0)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    model.to(gpu_device).eval()
    gb = GraphBuilder(connectivity=conn)
    tiles = []
    while len(tiles) < 24:
        d, m, _, res = _random_tile(rng, allow_defects=False)      # finite inputs: the logits bar is absolute
        if m.sum() >= 2:
            tiles.append((d, m, res))
    singles = []
    for d, m, res in tiles:
        g = gb.build_graph(d, m, None, res)
        o = graph_cpu.build_graph(d, m, None, res, connectivity=conn)
        out = model.predict(g)
        ref = gat_cpu.predict(sd, o.x, o.edge_index, o.edge_attr)
        assert (out["class_logits"].cpu() - ref["class_logits"]).abs().max().item() < TOL, (d.shape, int(m.sum()))
        assert (out["confidence"].cpu() - ref["confidence"]).abs().max().item() < TOL
        singles.append(out["class_logits"].clone())
    batch = gb.build_graphs([t[0] for t in tiles], [t[1] for t in tiles], None, [t[2] for t in tiles])
    lg = model.predict(batch)["class_logits"]
    assert torch.equal(lg, torch.cat(singles))                     # block-diagonal batching changes nothing, bit for bit


def test_fuzz_survey_geometry_device_stitch_equals_host_merge(gpu_device):
    """Random survey extents / tile sizes / overlaps / min_valid_ratio, including surveys smaller than one tile, zero
    overlap, extents that trigger the shift-back rule of the last tile row / column, and large nodata regions: the
    device path (cut -> infer -> bgnn_stitch_tiles) equals the host TileMerger on the same per-tile grids bit for bit."""
    from bathymetric_gnn_amd import synthetic
    from bathymetric_gnn_amd.config import Config
    from bathymetric_gnn_amd.data import BathymetricGrid
    from bathymetric_gnn_amd.models import BathymetricGNN, BathymetricPipeline
    rng = np.random.default_rng(4242)
    sd = synthetic.synthetic_state_dict(in_channels=7, seed=1234)
    model = BathymetricGNN(in_channels=7, edge_dim=3, dropout=0.0)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    model.to(gpu_device).eval()
    cases = [(40, 300, 64, 16, 0.1), (64, 64, 64, 0, 0.0), (65, 129, 64, 32, 0.5), (33, 47, 64, 16, 0.0)]
    for _ in range(8):
        tile = int(rng.choice([32, 48, 64, 96]))
        cases.append((int(rng.integers(20, 260)), int(rng.integers(20, 260)), tile,
                      int(rng.choice([0, 8, tile // 4, tile // 2])), float(rng.choice([0.0, 0.1, 0.5, 0.9]))))
    for H, W, tile, overlap, ratio in cases:
        d, _, _ = synthetic.synthetic_tile(H, W, int(rng.integers(1 << 30)), "V0")
        holes = rng.random((H, W)) < rng.choice([0.0, 0.05, 0.4])
        d[holes] = 1.0e6
        if rng.random() < 0.5:                                   # a block of nodata: whole tiles skipped
            r0, c0 = int(rng.integers(0, H)), int(rng.integers(0, W))
            d[r0:r0 + H // 2, c0:c0 + W // 2] = 1.0e6
        cfg = Config()
        cfg.tile.tile_size, cfg.tile.overlap, cfg.tile.min_valid_ratio = tile, overlap, ratio
        grid = BathymetricGrid(depth=d, nodata_value=1.0e6, resolution=(0.5, 0.5))
        pipe = BathymetricPipeline(cfg, tile_batch=int(rng.integers(1, 9)))
        pipe.set_model(model)
        res = pipe.process_grid(grid)
        pipe.host_stitch = True
        res_host = pipe.process_grid(grid)
        for k in res_host:
            ctx = f"{k}: survey {H}x{W}, tile {tile}, overlap {overlap}, min_valid_ratio {ratio}"
            assert np.array_equal(np.isnan(res[k]), np.isnan(res_host[k])), ctx
            assert np.array_equal(np.nan_to_num(res[k]).view(np.uint32), np.nan_to_num(res_host[k]).view(np.uint32)), ctx
