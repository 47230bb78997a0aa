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
