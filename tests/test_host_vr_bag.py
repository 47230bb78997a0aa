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


class _FakeProcessor:
    """The processor API run_refinements drives (add_to_batch / batch_ready / flush_batch / submit_batch / collect_batch_flat), with
    per-cell results that are a pure function of the cell's depth -- host logic of the loop without a GPU."""
    CLASS_NOISE = 2
    MAX_IN_FLIGHT = 2

    def __init__(self, budget):
        self.auto_correct_threshold = 0.5
        self.budget, self.fill, self.count, self.inflight = budget, [], 0, []

    @staticmethod
    def _results(d):
        valid = (d != np.float32(1.0e6)) & np.isfinite(d)
        frac = np.abs(np.where(valid, d, 0)).astype(np.float32) % np.float32(1.0)
        cls = np.where(valid, np.floor(frac * 3).astype(np.float32), 0).astype(np.float32)
        conf = np.where(valid, frac, 0).astype(np.float32)
        corr = np.where(valid, np.float32(0.25) + frac, 0).astype(np.float32)
        return cls, conf, corr

    def add_to_batch(self, depth, uncertainty, resolution, nodata=1.0e6, valid_count=None):
        nv = int(np.count_nonzero((depth != nodata) & np.isfinite(depth)))
        if nv == 0:
            z = np.zeros(depth.shape, np.float32)
            return (z, z.copy(), z.copy())
        self.fill.append(np.array(depth, np.float32)); self.count += nv
        return None

    batch_ready = property(lambda self: self.count >= self.budget)
    submit_ready = property(lambda self: self.count >= 2 * self.budget)
    batch_pending = property(lambda self: bool(self.fill))
    batches_in_flight = property(lambda self: len(self.inflight))

    def _take(self):
        grids, self.fill, self.count = self.fill, [], 0
        return grids

    def flush_batch(self):
        return [self._results(d) for d in self._take()]

    def submit_batch(self):
        assert len(self.inflight) < self.MAX_IN_FLIGHT
        self.inflight.append(self._take())

    def collect_batch_flat(self, copy=True):
        grids = self.inflight.pop(0)
        flat = np.stack([np.concatenate([r[k].ravel() for r in map(self._results, grids)]) for k in range(3)])
        return flat, [d.shape for d in grids]


def _permuted_index_bag(seed=21):
    """A BAG whose metadata index is NOT monotonic in iteration order: the grids' record ranges are dealt out in a shuffled order
    (still gap-free), so that a batch of consecutive grids spans one contiguous record range in another order."""
    md, ref = synthetic.synthetic_vr_bag(5, 6, seed=seed, lo=3, hi=14, empty_fraction=0.0, sparse_fraction=0.0)
    t = vr_bag.refinement_table(md)
    n_g = len(t["cells"])
    # (first and last grid stay where they are: a batch that holds the whole BAG then starts at the first record, ends at the last and
    #  covers exactly its cell count -- everything a check of the END POINTS can see -- with the grids in between in another order)
    order = np.concatenate([[0], 1 + np.random.default_rng(seed).permutation(n_g - 2), [n_g - 1]])
    md2, ref2, pos = md.copy(), ref.copy(), 0
    for k in order:
        n, i = int(t["cells"][k]), int(t["index"][k])
        ref2[0, pos:pos + n] = ref[0, i:i + n]
        md2[t["base_row"][k], t["base_col"][k]]["index"] = pos
        pos += n
    return md, ref, md2, ref2


@pytest.mark.parametrize("budget", [150, 600, 10 ** 9])
def test_pipelined_loop_on_a_bag_with_a_non_monotonic_index(budget):
    """run_refinements(pipelined=True) applies a collected batch in one pass over the grids' arrays 'back to back': that is a slice
    of the handler's plane only when every grid starts where the one before it ends.  On a BAG with a permuted index the slice
    shortcut would pair results with the wrong grids' depths; records, sink calls and statistics must equal the synchronous loop's
    -- and, grid for grid, those of the same BAG stored in iteration order."""
    from bathymetric_gnn_amd.scripts.inference_native import run_refinements
    md, ref, md2, ref2 = _permuted_index_bag()
    h, h2 = VRBagHandler.from_arrays(md, ref), VRBagHandler.from_arrays(md2, ref2)
    assert h.refinement_table()["contiguous"] and not h2.refinement_table()["contiguous"]
    runs = {}
    for name, handler, mode in (("plain", h, False), ("sync", h2, False), ("pipe", h2, True)):
        w, calls = handler.copy_and_open_for_writing(), []
        st = run_refinements(_FakeProcessor(budget), handler, w, 0.0, pipelined=mode, records_resident=False,
                             results_sink=lambda g, a, b, c: calls.append((g.base_row, g.base_col, a.copy(), b.copy(), c.copy())))
        runs[name] = (w, st, calls)
    (w_s, st_s, c_s), (w_p, st_p, c_p), (w_0, st_0, c_0) = runs["sync"], runs["pipe"], runs["plain"]
    assert np.array_equal(w_s.refinements.view(np.uint8), w_p.refinements.view(np.uint8))
    assert np.any(w_s.refinements["depth"] != ref2["depth"]) and w_s._corrections_applied == w_p._corrections_applied > 0
    for k in ("grids_processed", "cells_processed", "cells_classified_noise", "cells_corrected"):
        assert st_s[k] == st_p[k] == st_0[k], k
    assert abs(st_s["total_confidence"] - st_p["total_confidence"]) < 1e-4 * st_s["total_confidence"]
    assert len(c_s) == len(c_p) == len(c_0) == h.num_refinement_cells
    for x, y, z in zip(c_s, c_p, c_0):
        assert x[:2] == y[:2] == z[:2] and all(np.array_equal(a, b) and np.array_equal(a, c) for a, b, c in zip(x[2:], y[2:], z[2:]))
    # grid for grid the corrected records of the permuted BAG are those of the BAG in iteration order
    t, t2 = vr_bag.refinement_table(md), vr_bag.refinement_table(md2)
    for i, j, n in zip(t["index"], t2["index"], t["cells"]):
        assert np.array_equal(w_0.refinements[0, i:i + n], w_p.refinements[0, j:j + n])
