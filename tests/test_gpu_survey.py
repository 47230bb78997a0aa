"""Survey resident in HBM (BASELINE config 5 at test size): device tile cutting / valid counts
against numpy slicing, and the size-independent crop property the full-size tool (tools/survey_c5.py) also checks."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cut_tiles_and_valid_counts(gpu_device):
    from bathymetric_gnn_amd import runtime as rt
    ctx = rt.get_context(gpu_device)
    rng = np.random.default_rng(1)
    H, W, th, tw = 301, 517, 64, 96
    depth = rng.normal(size=(H, W)).astype(np.float32); valid = rng.random((H, W)) < 0.6; unc = rng.random((H, W)).astype(np.float32)
    org = np.array([[0, 0], [H - th, W - tw], [100, 37], [237, 421], [1, 1]], np.int32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)
    d_t, v_t, u_t, o_t = t(depth), t(valid.view(np.uint8)), t(unc), t(org)
    n = len(org)
    od = torch.empty(n * th * tw, device=gpu_device); ou = torch.empty(n * th * tw, device=gpu_device)
    om = torch.empty(n * th * tw, dtype=torch.uint8, device=gpu_device); cnt = torch.empty(n, dtype=torch.int64, device=gpu_device)
    ctx.begin()
    rt.check(ctx.lib.bgnn_cut_tiles(ctx.handle, H, W, rt.ptr(d_t), rt.ptr(v_t), rt.ptr(u_t), n, rt.ptr(o_t), th, tw, rt.ptr(od), rt.ptr(om), rt.ptr(ou)))
    rt.check(ctx.lib.bgnn_tile_valid_counts(ctx.handle, H, W, rt.ptr(v_t), n, rt.ptr(o_t), th, tw, rt.ptr(cnt)))
    ctx.end()
    torch.cuda.synchronize()
    od, om, ou = (x.cpu().numpy().reshape(n, th, tw) for x in (od, om, ou))
    for k, (r, c) in enumerate(org):
        assert np.array_equal(od[k], depth[r:r + th, c:c + tw]) and np.array_equal(ou[k], unc[r:r + th, c:c + tw])
        assert np.array_equal(om[k].astype(bool), valid[r:r + th, c:c + tw])
        assert cnt[k].item() == valid[r:r + th, c:c + tw].sum()
    with pytest.raises(ValueError):
        rt.check(ctx.lib.bgnn_cut_tiles(ctx.handle, H, W, rt.ptr(d_t), rt.ptr(v_t), None, n, rt.ptr(o_t), th, tw, rt.ptr(d_t), rt.ptr(v_t), rt.ptr(u_t)))   # uncertainty out without in


def test_survey_crop_property(gpu_device):
    """A crop whose origin sits on the tile lattice re-creates the survey's tiles; its interior cells (those no tile
    outside the crop covers) must equal the survey's, bit for bit -- tile batching, result offsets, stitch lookup."""
    from bathymetric_gnn_amd import synthetic
    from bathymetric_gnn_amd.config import Config
    from bathymetric_gnn_amd.models import BathymetricGNN, BathymetricPipeline
    cfg = Config(); cfg.tile.tile_size, cfg.tile.overlap = 64, 16          # stride 48
    pipe = BathymetricPipeline(cfg, tile_batch=7)
    sd = synthetic.synthetic_state_dict(seed=1234)
    m = BathymetricGNN(in_channels=7, edge_dim=3, dropout=0.0); m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    pipe.set_model(m.to(gpu_device).eval())
    d, mk, _ = synthetic.synthetic_tile(500, 430, 21, "V1")
    d[:120, :90] = 1.0e6
    depth = torch.from_numpy(d).to(gpu_device); valid = (depth != 1.0e6) & torch.isfinite(depth)
    full = pipe.process_survey_device(depth, valid, None, (0.5, 0.5))
    n_proc, n_skip = pipe.last_tile_counts
    assert n_skip > 0 and n_proc > 50
    r0, c0, L = 48 * 4, 48 * 3, 160                                        # crop tiles at 0, 48, 96 (= 160 - 64)
    sub = pipe.process_survey_device(depth[r0:r0 + L, c0:c0 + L].contiguous(), valid[r0:r0 + L, c0:c0 + L].contiguous(), None, (0.5, 0.5))
    a = full[:, r0 + 16:r0 + L - 16, c0 + 16:c0 + L - 16].contiguous().view(torch.int32)
    b = sub[:, 16:L - 16, 16:L - 16].contiguous().view(torch.int32)
    assert torch.equal(a, b)
    # and the host-array wrapper returns the same grids
    from bathymetric_gnn_amd.data import BathymetricGrid
    res = pipe.process_grid_device(BathymetricGrid(depth=d, nodata_value=1.0e6, resolution=(0.5, 0.5)))
    f = full.cpu().numpy()
    for k, name in enumerate(("classification", "confidence", "correction", "cleaned_depth")):
        assert np.array_equal(np.nan_to_num(res[name]).view(np.uint32), np.nan_to_num(f[k]).view(np.uint32))


@pytest.mark.parametrize("in_channels,band,foreign", [(7, 1, False), (7, 2, False), (8, 3, False), (7, 2, True)])
def test_streamed_host_survey_equals_the_resident_one(in_channels, band, foreign, gpu_device):
    """process_grid_streamed (host survey in, host grids out: upload thread + band-wise stitch + download thread, a ring of tile-row
    result slots) against process_survey_device on the whole survey resident in HBM: the five result grids bit for bit -- for several
    band heights (so that bands, ring slots and the survey's ragged last tile row / column fall differently), with an uncertainty
    band, with skipped tiles (nodata corner), with NaN / inf depths, and for a foreign grid object whose valid mask is its own."""
    from bathymetric_gnn_amd import synthetic
    from bathymetric_gnn_amd.config import Config
    from bathymetric_gnn_amd.data import BathymetricGrid
    from bathymetric_gnn_amd.models import BathymetricGNN, BathymetricPipeline
    cfg = Config(); cfg.tile.tile_size, cfg.tile.overlap = 64, 16          # stride 48
    pipe = BathymetricPipeline(cfg, tile_batch=5)
    pipe.STREAM_UPLOAD_ROWS_BYTES = 37 * 4 * 611                           # upload chunks of 37 rows: tile rows become ready mid-way
    d, mk, u = synthetic.synthetic_tile(700, 611, 33, "V1", True)
    d[:150, :130] = 1.0e6; d[300, 200] = np.nan; d[301, 200] = np.inf
    unc = u if in_channels == 8 else None
    # heads calibrated on one tile-sized crop of the survey (classes mix, so the label arbitration and the corrections are exercised)
    from _calibration import calibrate_heads
    from oracle import graph_cpu
    cr = (slice(400, 464), slice(300, 364))
    og = graph_cpu.build_graph(d[cr], (d[cr] != 1.0e6) & np.isfinite(d[cr]), None if unc is None else unc[cr], (0.5, 0.5))
    sd = calibrate_heads(synthetic.synthetic_state_dict(in_channels=in_channels, seed=1234), og.x, og.edge_index, og.edge_attr)
    m = BathymetricGNN(in_channels=in_channels, edge_dim=3, dropout=0.0); m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    pipe.set_model(m.to(gpu_device).eval())
    grid = BathymetricGrid(depth=d, uncertainty=unc, nodata_value=1.0e6, resolution=(0.5, 0.5))
    valid_np = grid.valid_mask
    if foreign:                       # not a BathymetricGrid: its valid_mask is taken as given (here: a hole the depth does not show)
        valid_np = valid_np.copy(); valid_np[400:420, 100:140] = False

        class Foreign:
            depth, uncertainty, resolution, shape, valid_mask, nodata_value = d, unc, (0.5, 0.5), d.shape, valid_np, 1.0e6
        grid = Foreign()
    depth_t = torch.from_numpy(d).to(gpu_device); valid_t = torch.from_numpy(valid_np).to(gpu_device)
    unc_t = torch.from_numpy(unc).to(gpu_device) if unc is not None else None
    full = pipe.process_survey_device(depth_t, valid_t, unc_t, (0.5, 0.5)).cpu().numpy()
    counts = pipe.last_tile_counts
    assert counts[1] > 0
    res = pipe.process_grid_streamed(grid, band_tile_rows=band)
    assert pipe.last_tile_counts == counts
    for k, name in enumerate(("classification", "confidence", "correction", "cleaned_depth")):
        assert res[name].shape == d.shape and res[name].dtype == np.float32
        assert np.array_equal(res[name].view(np.uint32), full[k].view(np.uint32)), name
    assert np.array_equal(res["valid_mask"], valid_np.astype(np.float32))
    assert np.isnan(res["classification"]).any() and (res["classification"] == 2).any()      # (uncovered invalid cells stay NaN)
    # the default entry takes the streamed form from STREAM_MIN_CELLS on
    pipe.STREAM_MIN_CELLS = 1
    res2 = pipe.process_grid(grid) if not foreign else pipe.process_grid_device(grid)
    assert all(np.array_equal(res2[k].view(np.uint32), res[k].view(np.uint32)) for k in res)


@pytest.mark.parametrize("S", [60000, 20000])
def test_config5_full_size_survey_resident_in_hbm(gpu_device, S):
    """BASELINE config 5 at FULL size on one GPU (models/pipeline.py:170-190 is the loop it stands for): a 60000 x 60000
    survey @0.5 m resident in HBM (14.4 GB of depth, ~151 GB peak), 24 336 overlapping 512 x 512 tiles, classified and
    stitched on the device.  Size-independent property: a 1280 x 1280 crop on the tile lattice at the FAR corner (cell
    offsets > 2^31, per-tile result offsets > 2^32) re-processed alone reproduces the survey's interior bit for bit.
    The heads are calibrated (tests/_calibration.py) so that classes mix and _apply_corrections really fires.
    Two explicit sizes, so the pass line says what ran: [60000] is SKIPPED (never shrunk) when less than 200 GB of HBM is
    free; [20000] needs 30 GB."""
    import json, os, time
    from _calibration import calibrate_heads
    from oracle import graph_cpu
    from bathymetric_gnn_amd import synthetic
    from bathymetric_gnn_amd.config import Config
    from bathymetric_gnn_amd.models import BathymetricGNN, BathymetricPipeline
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    need = 200e9 if S == 60000 else 30e9
    if free < need:
        pytest.skip(f"{S}x{S} survey needs {need / 1e9:.0f} GB of free HBM, {free / 1e9:.0f} GB available")
    depth, valid = synthetic.synthetic_survey_device(S, gpu_device, seed=0)
    # the synthetic depth runs from -20 m to -20 - 0.015 S m across the survey, so the head outputs drift with position:
    # calibrate on a crop from the MIDDLE of the survey -- classes then change over across it
    d0 = depth[S // 2:S // 2 + 96, S // 2:S // 2 + 96].cpu().numpy()
    og = graph_cpu.build_graph(d0, np.ones_like(d0, bool), None, (0.5, 0.5))
    sd = calibrate_heads(synthetic.synthetic_state_dict(seed=1234), og.x, og.edge_index, og.edge_attr)
    m = BathymetricGNN(in_channels=7, edge_dim=3, dropout=0.0); m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    cfg = Config(); cfg.tile.tile_size, cfg.tile.overlap = 512, 128
    pipe = BathymetricPipeline(cfg, tile_batch=32)
    pipe.set_model(m.to(gpu_device).eval())
    _, _, specs = pipe.tile_manager.compute_tile_grid((S, S))
    pipe.process_survey_device(depth[:1024, :1024].contiguous(), valid[:1024, :1024].contiguous(), None, (0.5, 0.5))   # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = pipe.process_survey_device(depth, valid, None, (0.5, 0.5))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n_proc, n_skip = pipe.last_tile_counts
    assert n_proc + n_skip == len(specs) and n_skip > 0 and (S != 60000 or len(specs) == 24336)
    hist = [int(((o[0] == k) & valid).sum().item()) for k in range(3)]
    n_valid = int(valid.sum().item())
    corrected = int(((o[3] != depth) & valid).sum().item())
    nan = torch.isnan(o[0])
    assert not bool((nan & valid).any().item())                            # every valid cell has a class ...
    assert 0 < int(nan.sum().item()) <= S * S - n_valid                    # ... NaN only on nodata cells no processed tile covers
    del nan
    assert sum(hist) == n_valid and sum(h > 0.005 * n_valid for h in hist) >= 2, hist    # classes change over across the survey
    assert corrected > 0                                                    # _apply_corrections fired
    k = (S - 1280) // 384 - 1
    r0 = c0 = 384 * k
    sub = pipe.process_survey_device(depth[r0:r0 + 1280, c0:c0 + 1280].contiguous(), valid[r0:r0 + 1280, c0:c0 + 1280].contiguous(),
                                     None, (0.5, 0.5))
    a = o[:, r0 + 128:r0 + 1152, c0 + 128:c0 + 1152].contiguous().view(torch.int32)
    b = sub[:, 128:1152, 128:1152].contiguous().view(torch.int32)
    same = bool(torch.equal(a, b))
    row = {"survey": f"{S}x{S}", "tiles": len(specs), "tiles_processed": n_proc, "tiles_skipped": n_skip,
           "node_evals": n_proc * 512 * 512, "wall_s": dt, "node_evals_per_s": n_proc * 512 * 512 / dt, "valid_cells": n_valid,
           "class_histogram": hist, "cells_corrected": corrected, "hbm_peak_GB": torch.cuda.max_memory_allocated() / 1e9,
           "crop_origin": [r0, c0], "crop_bit_identical": same}
    print("config5", json.dumps(row))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir) and os.access(out_dir, os.W_OK):
        json.dump(row, open(os.path.join(out_dir, f"config5_survey_{S}.json"), "w"), indent=1)
    del o, sub, depth, valid
    torch.cuda.empty_cache()
    assert same
