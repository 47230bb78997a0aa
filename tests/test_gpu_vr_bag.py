"""Whole-BAG device path through the C ABI: NativeVRProcessor.process_refinements (records resident
in HBM: bgnn_vr_unpack -> bgnn_infer_tiles -> bgnn_vr_apply) against the grid-by-grid loop of the reference's main
(run_refinements: iterate -> add_to_batch / flush_batch -> apply_results -> update_refinement_batch), and the two
record kernels against numpy statements of the same arithmetic.  Bit-exact: byte / float32 elementwise work."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _calibrated_state_dict(in_channels, seed, gain=100.0):
    """The seeded synthetic weights give almost node-independent outputs (logit spread ~0.01), i.e. one class and
    one confidence everywhere.  Re-centre and stretch the last layer of the classification and confidence heads
    around the oracle's mean outputs on one grid, so that classes mix and confidence straddles the threshold."""
    from bathymetric_gnn_amd import synthetic
    from oracle import gat_cpu, graph_cpu
    sd = dict(synthetic.synthetic_state_dict(in_channels=in_channels, seed=seed))
    d, u, r = synthetic.vr_grid_stream(1, seed0=4242)[0]
    og = graph_cpu.build_graph(d, (d != 1.0e6) & np.isfinite(d), u if in_channels == 8 else None, r)
    out = gat_cpu.forward(sd, og.x, og.edge_index, og.edge_attr)
    ml = out["class_logits"].numpy().astype(np.float64).mean(0)
    c = out["confidence"].numpy().astype(np.float64)
    mz = np.log(c / (1 - c)).mean()
    for head, mean in (("classification_head", ml), ("confidence_head", np.array([mz]))):
        w = np.asarray(sd[f"{head}.mlp.3.weight"], np.float64); b = np.asarray(sd[f"{head}.mlp.3.bias"], np.float64)
        sd[f"{head}.mlp.3.weight"] = (gain * w).astype(np.float32)
        sd[f"{head}.mlp.3.bias"] = (gain * (b - mean)).astype(np.float32)
    return sd


def _processor(in_channels=8, seed=1234, thr=0.5):
    from bathymetric_gnn_amd.data import GraphBuilder
    from bathymetric_gnn_amd.models import BathymetricGNN
    from bathymetric_gnn_amd.scripts.inference_native import NativeVRProcessor
    sd = _calibrated_state_dict(in_channels, seed)
    m = BathymetricGNN(in_channels=in_channels, hidden_channels=64, num_gnn_layers=4, heads=4, edge_dim=3, dropout=0.0)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    return NativeVRProcessor(m.to(torch.device("cuda:0")).eval(), GraphBuilder(), torch.device("cuda:0"), thr)


def test_unpack_and_apply_kernels(gpu_device):
    from bathymetric_gnn_amd import runtime as rt
    ctx = rt.get_context(gpu_device)
    rng = np.random.default_rng(0)
    sizes = rng.integers(1, 400, size=37); sizes[5] = 3000
    off = np.zeros(len(sizes) + 1, np.int64); np.cumsum(sizes, out=off[1:])
    n = int(off[-1])
    rec = np.empty((n, 2), np.float32)
    rec[:, 0] = rng.normal(-20, 3, n); rec[:, 1] = rng.uniform(0.05, 0.3, n)
    rec[rng.random(n) < 0.3, 0] = 1.0e6
    rec[rng.random(n) < 0.01, 0] = np.nan; rec[rng.random(n) < 0.01, 0] = np.inf; rec[rng.random(n) < 0.01, 0] = -np.inf
    lo, hi = off[7], off[8]; rec[lo:hi, 0] = 1.0e6                       # an empty grid
    lo, hi = off[5], off[6]; rec[lo:hi, 0] = 1.0e6; rec[lo + 2, 0] = -3.0  # 1 valid of 3000: below 0.01
    ratio = 0.01
    rec_t = torch.from_numpy(rec).to(gpu_device); off_t = torch.from_numpy(off).to(gpu_device)
    depth_t = torch.empty(n, device=gpu_device); unc_t = torch.empty(n, device=gpu_device)
    mask_t = torch.empty(n, dtype=torch.uint8, device=gpu_device)
    cnt_t = torch.empty(len(sizes), dtype=torch.int64, device=gpu_device)
    keep_t = torch.empty(len(sizes), dtype=torch.uint8, device=gpu_device)
    ctx.begin()
    rt.check(ctx.lib.bgnn_vr_unpack(ctx.handle, rt.ptr(rec_t), n, C.c_float(1.0e6), len(sizes), rt.ptr(off_t),
                                    C.c_double(ratio), rt.ptr(depth_t), rt.ptr(unc_t), rt.ptr(mask_t), rt.ptr(cnt_t),
                                    rt.ptr(keep_t)))
    ctx.end()
    torch.cuda.synchronize()
    valid = (rec[:, 0] != np.float32(1.0e6)) & np.isfinite(rec[:, 0])
    cnt = np.add.reduceat(valid.astype(np.int64), off[:-1])
    keep = (cnt / sizes) >= ratio
    assert not keep[5] and not keep[7] and keep.sum() > 30
    assert np.array_equal(depth_t.cpu().numpy().view(np.uint32), rec[:, 0].view(np.uint32))
    assert np.array_equal(unc_t.cpu().numpy(), rec[:, 1])
    assert np.array_equal(cnt_t.cpu().numpy(), cnt) and np.array_equal(keep_t.cpu().numpy().astype(bool), keep)
    exp_mask = valid & np.repeat(keep, sizes)
    assert np.array_equal(mask_t.cpu().numpy().astype(bool), exp_mask)

    # apply: numpy statement of scripts/inference_native.py:480-503
    cls = rng.integers(0, 3, n).astype(np.float32); conf = rng.random(n).astype(np.float32)
    conf[rng.random(n) < 0.05] = np.float32(0.85)                          # '>=' boundary
    corr = rng.normal(0, 1, n).astype(np.float32); corr[rng.random(n) < 0.1] = 0.0
    counts_t = torch.zeros(3, dtype=torch.int64, device=gpu_device); csum_t = torch.zeros(1, dtype=torch.float64, device=gpu_device)
    t = lambda a: torch.from_numpy(a).to(gpu_device)
    cls_t, conf_t, corr_t = t(cls), t(conf), t(corr)
    ctx.begin()
    rt.check(ctx.lib.bgnn_vr_apply(ctx.handle, rt.ptr(rec_t), n, rt.ptr(mask_t), rt.ptr(cls_t), rt.ptr(conf_t), rt.ptr(corr_t),
                                   C.c_float(0.85), rt.ptr(counts_t), rt.ptr(csum_t)))
    ctx.end()
    torch.cuda.synchronize()
    noise = (cls == 2) & exp_mask
    app = noise & (conf >= np.float32(0.85))
    d = rec[:, 0].copy(); u = rec[:, 1].copy()
    d[app] -= corr[app]; u[app] *= (2.0 - conf[app])
    got = rec_t.cpu().numpy()
    assert np.array_equal(got[:, 0].view(np.uint32), d.view(np.uint32))
    assert np.array_equal(got[:, 1].view(np.uint32), u.view(np.uint32))
    c = counts_t.cpu().numpy()
    assert c[0] == noise.sum() and c[1] == app.sum() and c[2] == ((d != rec[:, 0]) & exp_mask).sum()
    assert abs(csum_t.item() - conf[exp_mask].astype(np.float64).sum()) < 1e-6


@pytest.mark.parametrize("in_channels,ratio,budget", [(8, 0.01, 8 << 20), (7, 0.0, 3000), (8, 0.01, 1)])
def test_process_refinements_equals_grid_loop(in_channels, ratio, budget, gpu_device):
    from bathymetric_gnn_amd import synthetic
    from bathymetric_gnn_amd.data import VRBagHandler
    from bathymetric_gnn_amd.scripts.inference_native import run_refinements
    proc = _processor(in_channels)
    md, ref = synthetic.synthetic_vr_bag(6, 7, seed=77 + in_channels, lo=3, hi=30, empty_fraction=0.08, sparse_fraction=0.08)
    h = VRBagHandler.from_arrays(md, ref)
    # reference-shaped loop
    w_loop = h.copy_and_open_for_writing()
    sink = {}
    proc.BATCH_NODE_BUDGET = 4000
    st_loop = run_refinements(proc, h, w_loop, ratio, pipelined=False, results_sink=lambda g, a, b, c: sink.__setitem__(g.start_index, (a, b, c)))
    # device path
    w_dev = h.copy_and_open_for_writing()
    st_dev, res = proc.process_refinements(h, w_dev, ratio, cell_budget=budget, return_results=True)
    assert np.array_equal(w_dev.refinements.view(np.uint32), w_loop.refinements.view(np.uint32))
    assert 0 < st_dev["cells_corrected"] == st_loop["cells_corrected"] < st_dev["cells_classified_noise"] < st_dev["cells_processed"]
    for k in ("grids_processed", "cells_processed", "cells_classified_noise"):
        assert st_dev[k] == st_loop[k], k
    assert st_dev["grids_skipped"] == h.num_refinement_cells - st_loop["grids_processed"]
    assert (st_dev["grids_skipped"] > 0) == (ratio > 0)
    assert abs(st_dev["total_confidence"] - st_loop["total_confidence"]) < 1e-3 * max(1.0, st_loop["total_confidence"])
    assert w_dev._corrections_applied == w_loop._corrections_applied
    assert np.any(w_dev.refinements["depth"] != ref["depth"]) and np.array_equal(h.varres_refinements, ref)
    for start, (cls, conf, corr) in sink.items():
        n = cls.size
        assert np.array_equal(res[0, start:start + n], cls.ravel())
        assert np.array_equal(res[1, start:start + n].view(np.uint32), conf.ravel().view(np.uint32))
        assert np.array_equal(res[2, start:start + n].view(np.uint32), corr.ravel().view(np.uint32))


def test_process_refinements_non_contiguous_layout(gpu_device):
    """Records stored in reverse grid order with gaps: the path packs on the host and writes back per grid."""
    from bathymetric_gnn_amd import synthetic
    from bathymetric_gnn_amd.data import VRBagHandler
    from bathymetric_gnn_amd.data.vr_bag import refinement_table
    proc = _processor(8)
    md, ref = synthetic.synthetic_vr_bag(4, 5, seed=9, lo=3, hi=20)
    h = VRBagHandler.from_arrays(md, ref)
    w = h.copy_and_open_for_writing()
    proc.process_refinements(h, w, 0.01)
    t = refinement_table(md)
    md2 = md.copy(); gap = 5
    total = int(t["cells"].sum()) + gap * len(t["cells"])
    ref2 = np.zeros((1, total), ref.dtype); ref2["depth"] = 1.0e6
    pos = total
    for r, c, i, n in zip(t["base_row"], t["base_col"], t["index"], t["cells"]):
        pos -= n + gap
        ref2[0, pos:pos + n] = ref[0, i:i + n]
        md2[r, c]["index"] = pos
    h2 = VRBagHandler.from_arrays(md2, ref2)
    assert not h2.refinement_table()["contiguous"]
    w2 = h2.copy_and_open_for_writing()
    proc.process_refinements(h2, w2, 0.01)
    for r, c, i, n in zip(t["base_row"], t["base_col"], t["index"], t["cells"]):
        j = int(md2[r, c]["index"])
        assert np.array_equal(w2.refinements[0, j:j + n], w.refinements[0, i:i + n])
    assert w2._corrections_applied == w._corrections_applied


@pytest.mark.parametrize("base,min_grids", [(35, 900), (70, 4096)])
def test_config4_scale_stream_device_equals_loop(base, min_grids, gpu_device):
    """BASELINE config 4 at a quarter of its size (~1 000 refinement grids 3x3..50x50, ~0.7 M cells) and at FULL size
    (>= 4 096 grids, ~2.9 M cells): the whole-BAG device path and the reference-shaped 50 000-node loop
    (scripts/inference_native.py:520-538) write identical records and report identical counters."""
    from bathymetric_gnn_amd import synthetic
    from bathymetric_gnn_amd.data import VRBagHandler
    from bathymetric_gnn_amd.scripts.inference_native import run_refinements
    proc = _processor(8)
    md, ref = synthetic.synthetic_vr_bag(base, base, seed=4000)
    h = VRBagHandler.from_arrays(md, ref)
    assert h.num_refinement_cells >= min_grids
    w_loop, w_dev = h.copy_and_open_for_writing(), h.copy_and_open_for_writing()
    st_loop = run_refinements(proc, h, w_loop, 0.01, pipelined=False)
    st_dev = proc.process_refinements(h, w_dev, 0.01)
    assert np.array_equal(w_dev.refinements.view(np.uint32), w_loop.refinements.view(np.uint32))
    for k in ("grids_processed", "cells_processed", "cells_classified_noise", "cells_corrected"):
        assert st_dev[k] == st_loop[k], k
    assert st_dev["cells_corrected"] > 0


def test_pipelined_grid_loop_equals_the_synchronous_one(gpu_device):
    """run_refinements(pipelined=True) -- (i) the grid loop with two coalesced submissions in flight (submit_batch / collect_batch on two
    library contexts; ``records_resident=False``), (ii) what it does by default for a VRBagHandler + VRBagWriter: the records-resident
    whole-BAG path (process_refinements, several chunks in flight) -- writes the same records, feeds the sink the same per-grid
    arrays in the same order and reports the same statistics as the reference-shaped synchronous loop (one flush_batch per full
    batch); the submit / collect API itself keeps submission order and refuses a third batch in flight."""
    from bathymetric_gnn_amd import synthetic
    from bathymetric_gnn_amd.data import VRBagHandler
    from bathymetric_gnn_amd.scripts.inference_native import run_refinements
    proc = _processor(8)
    proc.BATCH_NODE_BUDGET = 1500
    proc.PIPELINE_COALESCE = 2
    proc.BAG_CHUNK_MIN = 2500                                             # (the whole-BAG path in ~4 chunks on this small BAG)
    md, ref = synthetic.synthetic_vr_bag(7, 8, seed=123, lo=3, hi=30, empty_fraction=0.1, sparse_fraction=0.05)
    h = VRBagHandler.from_arrays(md, ref)
    runs = {}
    for name, kw in (("sync", dict(pipelined=False)), ("loop", dict(pipelined=True, records_resident=False)), ("routed", dict(pipelined=True))):
        w = h.copy_and_open_for_writing()
        order, sink = [], {}
        chunks = []
        if name == "routed":
            launched = proc._bag_slot
            proc._bag_slot = lambda *a: (chunks.append(a[0]), launched(*a))[1]
        st = run_refinements(proc, h, w, 0.0, **kw,
                             results_sink=lambda g, a, b, c: (order.append(g.start_index), sink.__setitem__(g.start_index, (a.copy(), b.copy(), c.copy()))))
        runs[name] = (w.refinements.copy(), st, order, sink, w._corrections_applied)
        assert proc.batches_in_flight == 0
        if name == "loop":
            assert len(proc._engines) == 2                                # the second context was really used
        if name == "routed":
            del proc._bag_slot
            assert len(chunks) >= 3 and set(chunks) >= {0, 1, 2}         # several chunks, on several slots (two contexts)
    a = runs["sync"]
    for name in ("loop", "routed"):
        b = runs[name]
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), name
        assert a[2] == b[2] and len(a[2]) > 30 and a[4] == b[4] > 0, name
        assert set(a[1]) == set(b[1])
        for k in a[1]:
            if k in ("total_confidence", "mean_confidence"):     # per batch in float64 vs per grid in float32 pairwise order (logged mean only)
                assert abs(a[1][k] - b[1][k]) <= 1e-6 * max(1.0, abs(a[1][k])), (name, k)
            else:
                assert a[1][k] == b[1][k], (name, k)
        for s0 in a[3]:
            for x, y in zip(a[3][s0], b[3][s0]):
                assert x.shape == y.shape and np.array_equal(x.view(np.uint32), y.view(np.uint32)), name
    # the API: order of collection = order of submission; at most two in flight
    grids = synthetic.vr_grid_stream(9, seed0=900)
    single = [proc.process_grid(d, u, r) for d, u, r in grids]
    for lo in (0, 3, 6):
        for d, u, r in grids[lo:lo + 3]:
            assert proc.add_to_batch(d, u, r) is None
        if lo < 6:
            assert proc.submit_batch() == lo // 3 + 1
    with pytest.raises(RuntimeError):
        proc.submit_batch()                                              # a third batch in flight
    got = proc.collect_batch() + proc.collect_batch()
    got += proc.flush_batch()                                            # the queued third batch, synchronously
    assert proc.collect_batch() == [] and len(got) == 9
    for (c0, f0, r0), (c1, f1, r1) in zip(single, got):            # (same bar as the batched-vs-single comparison in test_gpu_forward)
        assert np.array_equal(c0, c1) and np.abs(f0 - f1).max() < 1e-6 and np.abs(r0 - r1).max() < 1e-6


def test_staged_batches_device_mask_fresh_results_and_mixed_nodata(gpu_device):
    """add_to_batch copies a grid into a pinned record slab and the valid mask is made on the device at flush time.  (i) The results
    equal the per-grid path (host mask) -- with and without the `valid_count` hint, NaN / inf / nodata cells included; (ii) a queued
    grid is COPIED at add time (changing the caller's array afterwards changes nothing); (iii) results are arrays of their own (a
    later batch does not overwrite them); (iv) a batch that mixes nodata values falls back to host-made masks and still agrees;
    (v) flush_batch() while two batches are in flight leaves those undisturbed."""
    from bathymetric_gnn_amd import synthetic
    proc = _processor(8)
    grids = synthetic.vr_grid_stream(12, seed0=4100)
    grids = [(d.copy(), u, r) for d, u, r in grids]
    grids[2][0][0, 0] = np.nan; grids[2][0][-1, -1] = np.inf; grids[5][0][1, :] = 1.0e6
    single = [proc.process_grid(d, u, r) for d, u, r in grids]

    def same(a, b):
        return all(np.array_equal(x[0], y[0]) and np.abs(x[1] - y[1]).max() < 1e-6 and np.abs(x[2] - y[2]).max() < 1e-6 for x, y in zip(a, b))

    for hint in (False, True):
        for d, u, r in grids:
            nv = int(np.count_nonzero((d != 1.0e6) & np.isfinite(d))) if hint else None
            assert proc.add_to_batch(d, u, r, valid_count=nv) is None
        got = proc.flush_batch()
        assert len(got) == 12 and same(single, got)
    # (ii) copied at add time
    d0 = grids[0][0].copy()
    proc.add_to_batch(d0, grids[0][1], grids[0][2])
    d0[:] = 123.0
    r0 = proc.flush_batch()
    assert same(single[:1], r0)
    # (iii) results own their memory
    keep = [tuple(x.copy() for x in t) for t in r0]
    for d, u, r in grids[6:]:
        proc.add_to_batch(d, u, r)
    proc.flush_batch()
    assert all(np.array_equal(a, b) for t, k in zip(r0, keep) for a, b in zip(t, k))
    # (iv) mixed nodata values in one batch: a grid whose nodata is -9999 and which holds 1e6 as a REAL depth
    dm = grids[3][0].copy(); dm[0, :] = -9999.0; dm[1, 1] = 1.0e6
    want = proc.process_grid(dm, grids[3][1], grids[3][2], nodata=-9999.0)
    other = proc.process_grid(dm, grids[3][1], grids[3][2])          # under the default nodata: row 0 valid, the 1e6 cell not
    assert not same([want], [other])
    proc.add_to_batch(grids[1][0], grids[1][1], grids[1][2])
    proc.add_to_batch(dm, grids[3][1], grids[3][2], nodata=-9999.0)
    proc.add_to_batch(grids[4][0], grids[4][1], grids[4][2])
    got = proc.flush_batch()
    assert same([single[1], want, single[4]], got)
    assert float(np.abs(got[1][1][0, :]).max()) == 0.0               # row 0 (this grid's nodata) is invalid: zeros
    # (v) flush while two batches are in flight
    for lo in (0, 4):
        for d, u, r in grids[lo:lo + 4]:
            proc.add_to_batch(d, u, r)
        proc.submit_batch()
    for d, u, r in grids[8:]:
        proc.add_to_batch(d, u, r)
    third = proc.flush_batch()
    first, second = proc.collect_batch(), proc.collect_batch()
    assert same(single[:4], first) and same(single[4:8], second) and same(single[8:], third)
    # a grid larger than the slab's initial capacity (131 072 cells) joins a batch that already holds grids: the slab grows, what was
    # staged before survives; a float64 / non-contiguous depth array is converted on the way in
    big_d, big_m, big_u = synthetic.synthetic_tile(400, 420, 9, "V1", True)
    big_d = np.where(big_m, big_d, np.float32(1.0e6)).astype(np.float32)
    want_big = proc.process_grid(big_d, big_u, (0.5, 0.5))
    proc.add_to_batch(grids[0][0], grids[0][1], grids[0][2])
    proc.add_to_batch(np.asfortranarray(big_d.astype(np.float64)), big_u, (0.5, 0.5))
    proc.add_to_batch(grids[2][0], grids[2][1], grids[2][2])
    got = proc.flush_batch()
    assert same([single[0], want_big, single[2]], got) and got[1][0].shape == (400, 420)
    # mixing grids with and without an uncertainty layer in one batch is refused when queued
    proc.add_to_batch(grids[0][0], grids[0][1], grids[0][2])
    with pytest.raises(ValueError, match="uncertainty"):
        proc.add_to_batch(grids[1][0], None, grids[1][2])
    proc.flush_batch()
    # an all-invalid grid never enters a batch
    z = proc.add_to_batch(np.full((4, 5), 1.0e6, np.float32), None, (1.0, 1.0))
    assert z is not None and all(float(np.abs(a).max()) == 0.0 and a.shape == (4, 5) for a in z) and not proc.batch_pending


def test_pipelined_loop_with_a_foreign_writer_takes_the_per_grid_path(gpu_device):
    """run_refinements(pipelined=True) applies a collected batch in one vectorised pass and writes back through the writer's bulk
    entry; a writer WITHOUT one (the reference's writer has only update_refinement_batch) gets the per-grid calls, same end state."""
    from bathymetric_gnn_amd import synthetic
    from bathymetric_gnn_amd.data import VRBagHandler
    from bathymetric_gnn_amd.scripts.inference_native import run_refinements
    proc = _processor(8)
    proc.BATCH_NODE_BUDGET = 2500
    md, ref = synthetic.synthetic_vr_bag(6, 6, seed=77, lo=3, hi=28, empty_fraction=0.1)
    h = VRBagHandler.from_arrays(md, ref)

    class PerGridOnly:
        def __init__(self, w):
            self.w, self.calls = w, 0

        def update_refinement_batch(self, grid, d, u):
            self.calls += 1
            self.w.update_refinement_batch(grid, d, u)

    w_bulk = h.copy_and_open_for_writing()
    st_bulk = run_refinements(proc, h, w_bulk, 0.0, pipelined=True, records_resident=False)     # the loop, bulk write-back
    w_plain = PerGridOnly(h.copy_and_open_for_writing())
    st_plain = run_refinements(proc, h, w_plain, 0.0, pipelined=True)        # no write_records: cannot be routed, per-grid calls
    assert w_plain.calls == st_plain["grids_processed"] == h.num_refinement_cells
    assert np.array_equal(w_bulk.refinements.view(np.uint32), w_plain.w.refinements.view(np.uint32))
    assert w_bulk._corrections_applied == w_plain.w._corrections_applied
    assert st_bulk == st_plain
    with pytest.raises(ValueError, match="records_resident"):
        run_refinements(proc, h, w_plain, 0.0, records_resident=True)
    w_dev = h.copy_and_open_for_writing()
    st_dev = run_refinements(proc, h, w_dev, 0.0)                            # default for this handler / writer: records resident
    assert np.array_equal(w_bulk.refinements.view(np.uint32), w_dev.refinements.view(np.uint32))
    assert w_bulk._corrections_applied == w_dev._corrections_applied
    assert all(st_dev[k] == st_bulk[k] for k in st_bulk if "confidence" not in k) and set(st_dev) == set(st_bulk)


def test_grid_count_cap_closes_a_batch_and_the_library_names_its_limit(gpu_device):
    """The per-grid kernels index grids by gridDim.y: the library refuses more than 60 000 grids in one batch with a message that
    says so (not a failed launch), and the processor closes a batch at MAX_GRIDS_PER_BATCH grids whatever the node budget."""
    from bathymetric_gnn_amd import runtime as rt, synthetic
    from bathymetric_gnn_amd.data import GraphBuilder
    from bathymetric_gnn_amd.models import BathymetricGNN
    from bathymetric_gnn_amd.scripts.inference_native import NativeVRProcessor
    sd = synthetic.synthetic_state_dict(in_channels=8, seed=5)
    model = BathymetricGNN(in_channels=8, edge_dim=3, dropout=0.0)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    model.to(gpu_device).eval()
    proc = NativeVRProcessor(model, GraphBuilder(device=gpu_device), gpu_device)
    proc.BATCH_NODE_BUDGET = 10 ** 9
    proc.MAX_GRIDS_PER_BATCH = 50
    rng = np.random.default_rng(0)
    for i in range(50):
        assert not proc.batch_ready
        d = rng.normal(-20.0, 1.0, (3, 3)).astype(np.float32)
        assert proc.add_to_batch(d, np.full((3, 3), 0.1, np.float32), (0.5, 0.5)) is None
    assert proc.batch_ready
    assert len(proc.flush_batch()) == 50
    n = 60001
    gb = GraphBuilder(device=gpu_device)
    d = np.zeros((n * 4,), np.float32)
    with pytest.raises((rt.BgnnError, ValueError), match="at most 60000"):
        gb.build_graphs([d[4 * i:4 * i + 4].reshape(2, 2) for i in range(n)], [np.ones((2, 2), bool)] * n, None, [(0.5, 0.5)] * n)
