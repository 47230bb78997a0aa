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


def _iterate_like_the_reference(h, min_valid_ratio=0.0):
    """The reference's loop, statement for statement (data/vr_bag.py:243-298): metadata cell by cell, one slice + reshape +
    copy per grid, the valid ratio from the grid's own mask."""
    ref, md = h.varres_refinements[0, :], h.varres_metadata
    for row in range(md.shape[0]):
        for col in range(md.shape[1]):
            meta = md[row, col]
            dx, dy = int(meta["dimensions_x"]), int(meta["dimensions_y"])
            if dx == 0 or dy == 0:
                continue
            s = int(meta["index"])
            sl = ref[s:s + dx * dy]
            g = RefinementGrid(row, col, sl["depth"].reshape(dy, dx).copy(), sl["depth_uncrt"].reshape(dy, dx).copy(),
                               (float(meta["resolution_x"]), float(meta["resolution_y"])), (dy, dx),
                               (float(meta["sw_corner_x"]), float(meta["sw_corner_y"])), s)
            if g.num_valid / g.depth.size >= min_valid_ratio:
                yield g


@pytest.mark.parametrize("ratio", [0.0, 0.3, 1.0])
def test_vectorised_iterator_yields_what_the_reference_loop_yields(ratio):
    md, ref = synthetic.synthetic_vr_bag(6, 7, seed=5, lo=3, hi=24, empty_fraction=0.15, sparse_fraction=0.2)
    ref = ref.copy()
    ref["depth"][0, 5] = np.nan; ref["depth"][0, 17] = np.inf           # non-finite depths are invalid cells too
    h = VRBagHandler.from_arrays(md, ref)
    a, b = list(_iterate_like_the_reference(h, ratio)), list(h.iterate_refinements(ratio))
    assert len(a) == len(b) and (ratio < 1.0 or len(b) < h.num_refinement_cells)
    for x, y in zip(a, b):
        assert (x.base_row, x.base_col, x.start_index, x.dimensions, x.resolution, x.sw_corner) == \
               (y.base_row, y.base_col, y.start_index, y.dimensions, y.resolution, y.sw_corner)
        assert np.array_equal(x.depth.view(np.uint32), y.depth.view(np.uint32)) and np.array_equal(x.uncertainty, y.uncertainty)
        assert y.depth.flags.c_contiguous and y.depth.dtype == np.float32
        assert x.num_valid == y.num_valid == int(np.sum(y.valid_mask))
    # the grids are views of private planes: writing into one does not touch the handler's records
    before = h.varres_refinements.copy()
    b[0].depth[:] = -1.0
    assert np.array_equal(h.varres_refinements.view(np.uint8), before.view(np.uint8))


def test_bulk_write_back_equals_the_per_grid_calls():
    md, ref = synthetic.synthetic_vr_bag(5, 5, seed=8, lo=3, hi=15, empty_fraction=0.1)
    h = VRBagHandler.from_arrays(md, ref)
    grids = [g for i, g in enumerate(h.iterate_refinements()) if i % 3 != 1]       # a subset with gaps, in order
    rng = np.random.default_rng(0)
    new_d = [np.where(rng.random(g.shape) < 0.3, g.depth - np.float32(0.5), g.depth) for g in grids]
    new_u = [g.uncertainty * np.float32(1.25) for g in grids]
    w1, w2 = h.copy_and_open_for_writing(), h.copy_and_open_for_writing()
    for g, d, u in zip(grids, new_d, new_u):
        w1.update_refinement_batch(g, d, u)
    changed = sum(int(np.sum((d != g.depth) & g.valid_mask)) for g, d in zip(grids, new_d))
    w2.update_refinements_bulk(grids, np.concatenate([d.ravel() for d in new_d]), np.concatenate([u.ravel() for u in new_u]), changed=changed)
    assert np.array_equal(w1.refinements.view(np.uint8), w2.refinements.view(np.uint8))
    assert w1._corrections_applied == w2._corrections_applied == changed > 0
    with pytest.raises(ValueError):
        w2.update_refinements_bulk(grids, np.zeros(3, np.float32), None)
