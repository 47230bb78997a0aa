"""VR BAG array interface: handlers / writers of data/vr_bag.py on the two HDF5 arrays.

Parity status: the reference's iteration and write-back (data/vr_bag.py:243-298, 550-588) go through h5py, which is
absent here and on the GPU box, and the reference's tests hold no fixture for them -> "parity unpinned"; the
statements below restate the documented behaviour (iteration order, record slicing, filters, counters) and check
self-consistency."""
import numpy as np
import pytest

from bathymetric_gnn_amd import synthetic
from bathymetric_gnn_amd.data import vr_bag
from bathymetric_gnn_amd.data.vr_bag import (RefinementGrid, SRBagHandler, VRBagHandler, VRBagWriter,
                                             VARRES_METADATA_DTYPE, VARRES_REFINEMENT_DTYPE)


def _bag(seed=11, rows=5, cols=6):
    return synthetic.synthetic_vr_bag(rows, cols, seed=seed, lo=3, hi=20)


def test_refinement_grid_valid_mask():
    d = np.array([[1.0, 1.0e6, np.nan], [np.inf, -5.0, 0.0]], np.float32)
    g = RefinementGrid(0, 0, d, np.zeros_like(d), (1.0, 1.0), d.shape, (0.0, 0.0), 0)
    assert g.valid_mask.tolist() == [[True, False, False], [False, True, True]]
    assert g.num_valid == 3 and g.shape == (2, 3)


def test_iteration_order_and_slicing():
    md, ref = _bag()
    h = VRBagHandler.from_arrays(md, ref)
    grids = list(h.iterate_refinements())
    assert len(grids) == h.num_refinement_cells == int(np.sum(md["dimensions_x"] > 0))
    # row-major over the base grid, cells without refinement skipped
    assert [(g.base_row, g.base_col) for g in grids] == sorted((g.base_row, g.base_col) for g in grids)
    pos = 0
    for g in grids:
        m = md[g.base_row, g.base_col]
        assert g.dimensions == (int(m["dimensions_y"]), int(m["dimensions_x"])) == g.depth.shape
        assert g.start_index == int(m["index"]) == pos
        n = g.depth.size
        assert np.array_equal(g.depth.ravel(), ref["depth"][0, pos:pos + n])
        assert np.array_equal(g.uncertainty.ravel(), ref["depth_uncrt"][0, pos:pos + n])
        assert g.resolution == (float(m["resolution_x"]), float(m["resolution_y"]))
        g.depth[:] = 0            # copies: the handler's arrays are untouched
        pos += n
    assert pos == h.total_refinement_nodes == ref.shape[1]
    assert np.any(ref["depth"] != 0)
    info = h.get_refinement_info()
    assert info["num_refined_cells"] == len(grids) and info["total_refinement_nodes"] == pos
    assert h.finest_resolution == float(min(md["resolution_x"][md["dimensions_x"] > 0]))


def test_min_valid_ratio_filter():
    md, ref = synthetic.synthetic_vr_bag(8, 8, seed=5, lo=3, hi=20, empty_fraction=0.2, sparse_fraction=0.2)
    h = VRBagHandler.from_arrays(md, ref)
    all_g = list(h.iterate_refinements(0.0))
    kept = list(h.iterate_refinements(0.01))
    ratios = [g.num_valid / g.depth.size for g in all_g]
    assert len(kept) == sum(r >= 0.01 for r in ratios) < len(all_g)
    assert any(r == 0 for r in ratios)


def test_refinement_table_contiguity():
    md, ref = _bag()
    t = vr_bag.refinement_table(md)
    assert t["contiguous"] and int(t["cells"].sum()) == ref.shape[1]
    md2 = md.copy()
    r, c = np.argwhere(md["dimensions_x"] > 0)[1]
    md2[r, c]["index"] += 3
    assert not vr_bag.refinement_table(md2)["contiguous"]
    empty = np.zeros((2, 2), VARRES_METADATA_DTYPE)
    t0 = vr_bag.refinement_table(empty)
    assert len(t0["cells"]) == 0 and t0["contiguous"]


def test_writer_roundtrip_and_counters():
    md, ref = _bag(seed=3)
    h = VRBagHandler.from_arrays(md, ref)
    w = h.copy_and_open_for_writing()
    orig = ref.copy()
    grids = list(h.iterate_refinements())
    g = next(x for x in grids if x.num_valid > 4)
    d = g.depth.copy(); u = g.uncertainty.copy()
    ij = np.argwhere(g.valid_mask)[:3]
    for i, j in ij:
        d[i, j] -= 1.5; u[i, j] *= 1.2
    inv = np.argwhere(~g.valid_mask)
    if len(inv):
        d[tuple(inv[0])] = 7.0                      # a changed invalid cell is written but not counted
    w.update_refinement_batch(g, d, u)
    n = g.depth.size
    assert np.array_equal(w.refinements["depth"][0, g.start_index:g.start_index + n], d.ravel())
    assert np.array_equal(w.refinements["depth_uncrt"][0, g.start_index:g.start_index + n], u.ravel())
    assert w._corrections_applied == 3
    # everything outside the grid untouched; the handler's own array untouched
    keep = np.ones(ref.shape[1], bool); keep[g.start_index:g.start_index + n] = False
    assert np.array_equal(w.refinements[0, keep], orig[0, keep]) and np.array_equal(ref, orig)
    with pytest.raises(ValueError):
        w.update_refinement_batch(g, d[:-1], u[:-1])
    w2 = VRBagWriter.from_arrays(orig.copy())
    w2.update_refinement(g, d, u)
    assert np.array_equal(w2.refinements, w.refinements) and w2._uncertainty_updates == 3
    w3 = VRBagWriter.from_arrays(orig.copy())
    recs = np.stack([d.ravel(), u.ravel()], 1)
    w3.write_records(g.start_index, recs, corrections_applied=3)
    assert np.array_equal(w3.refinements, w.refinements)
    with w:
        pass


def test_sr_handler_single_grid():
    d = np.full((6, 7), -10.0, np.float32); d[0, :] = 1.0e6
    h = SRBagHandler.from_arrays(d, None, resolution=2.0)
    (g,) = list(h.iterate_refinements(0.5))
    assert g.shape == (6, 7) and g.resolution == (2.0, 2.0) and g.num_valid == 35
    assert list(h.iterate_refinements(0.9)) == []
    w = h.copy_and_open_for_writing()
    w.update_refinement_batch(g, g.depth + 1, g.uncertainty)
    assert np.array_equal(w.elevation, d + 1)


def test_file_backed_needs_h5py():
    if vr_bag.H5PY_AVAILABLE:
        pytest.skip("h5py present")
    for ctor in (VRBagHandler, VRBagWriter, SRBagHandler, vr_bag.detect_bag_type):
        with pytest.raises(ImportError):
            ctor("/nonexistent.bag")


def test_bad_arrays_rejected():
    md, ref = _bag()
    with pytest.raises(ValueError):
        VRBagHandler.from_arrays(np.zeros((2, 2), np.float32), ref)
    with pytest.raises(ValueError):
        VRBagHandler.from_arrays(md, np.zeros((1, 4), np.float32))
    assert VRBagHandler.from_arrays(md, ref[0]).varres_refinements.shape == ref.shape
    assert ref.dtype == VARRES_REFINEMENT_DTYPE
