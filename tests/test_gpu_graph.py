"""GPU parity, graph construction (K1/K2): the HIP path through the C ABI vs the golden vectors
produced by the reference's own GraphBuilder, and vs the CPU oracle on seeded inputs.

Bars: edge_index bit-exact (integer work).  x / edge_attr / local_std: the kernels restate the
numpy/scipy float sequences operation by operation and are expected bit-exact too; the asserted
tolerance is <= 1 float32 ulp for the two quantities that go through a float64 libm call on the
device (slope: atan; local_std: sqrt of a cancelling difference), 0 ulp for everything else."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden, ulp_diff_f32
from oracle import graph_cpu

pytestmark = pytest.mark.gpu


def _builder(g):
    from bathymetric_gnn_amd.data import GraphBuilder
    return GraphBuilder(connectivity=g["conn"], include_self_loops=g["loops"])


def _check_x(x, ref, names=graph_cpu.DEFAULT_NODE_FEATURES):
    assert x.shape == ref.shape
    u = ulp_diff_f32(x, ref)
    for c in range(ref.shape[1]):
        tol = 1 if (c < len(names) and names[c] == "local_std") else 0
        assert u[:, c].max(initial=0) <= tol, f"column {c}: {u[:, c].max()} ulp (max abs {np.abs(x[:, c] - ref[:, c]).max()})"


def _check_ea(ea, ref):
    assert ea.shape == ref.shape
    u = ulp_diff_f32(ea, ref)
    assert u[:, 0].max(initial=0) == 0 and u[:, 1].max(initial=0) == 0
    assert u[:, 2].max(initial=0) <= 1


@pytest.mark.parametrize("name", golden_names())
def test_graph_matches_reference_golden(name, gpu_device):
    g = load_golden(name)
    gb = _builder(g)
    graph = gb.build_graph(g["depth"], g["mask_arg"], g["unc_arg"], g["res"])
    N, E = int(g["num_nodes"]), int(g["num_edges"])
    assert graph.num_nodes == N
    if N == 0:
        assert graph.x.shape == (0, 7) and graph.edge_index.shape == (2, 0) and not hasattr(graph, "grid_shape")
        return
    assert graph.num_edges == E
    ei = graph.edge_index.cpu().numpy()
    assert ei.dtype == np.int64 and ei.shape == (2, E)
    assert hashlib.sha256(np.ascontiguousarray(ei).tobytes()).hexdigest() == str(g["edge_index_sha256"])
    x = graph.x.cpu().numpy(); ea = graph.edge_attr.cpu().numpy(); ls = graph.local_std.cpu().numpy()
    if "x" in g:
        assert np.array_equal(ei, g["edge_index"].astype(np.int64))
        _check_x(x, g["x"]); _check_ea(ea, g["edge_attr"])
        assert ulp_diff_f32(ls, g["local_std"]).max(initial=0) <= 1
        assert np.array_equal(graph.pos.cpu().numpy(), g["pos"])
        assert np.array_equal(graph.valid_rows.cpu().numpy(), g["valid_rows"])
        assert np.array_equal(graph.valid_cols.cpu().numpy(), g["valid_cols"])
        assert graph.grid_shape == g["depth"].shape and graph.num_valid_cells == N
    else:
        _check_x(x[:1024], g["x_head"]); _check_x(x[-1024:], g["x_tail"])
        _check_ea(ea[:1024], g["ea_head"]); _check_ea(ea[-1024:], g["ea_tail"])
        o = graph_cpu.build_graph(g["depth"], g["mask_arg"], g["unc_arg"], g["res"])
        _check_x(x, o.x); _check_ea(ea, o.edge_attr)


@pytest.mark.parametrize("shape,variant,unc,seed", [((33, 70), "V1", False, 5), ((128, 96), "V1", True, 6),
                                                     ((50, 50), "V0", True, 7), ((2, 2), "V0", False, 8),
                                                     ((300, 17), "V1", False, 9)])
def test_graph_matches_oracle_seeded(shape, variant, unc, seed, gpu_device):
    from bathymetric_gnn_amd import synthetic
    from bathymetric_gnn_amd.data import GraphBuilder
    d, m, u = synthetic.synthetic_tile(shape[0], shape[1], seed, variant if min(shape) >= 16 else "V0", unc)
    res = (0.5, 2.0)
    o = graph_cpu.build_graph(d, m, u, res)
    g = GraphBuilder().build_graph(d, m, u, res)
    assert g.num_nodes == o.num_nodes and g.num_edges == o.num_edges
    assert np.array_equal(g.edge_index.cpu().numpy(), o.edge_index)
    names = graph_cpu.DEFAULT_NODE_FEATURES + (["uncertainty"] if unc else [])
    _check_x(g.x.cpu().numpy(), o.x, names)
    _check_ea(g.edge_attr.cpu().numpy(), o.edge_attr)


def test_batched_graph_equals_concatenation(gpu_device):
    """Batch.from_data_list semantics: node / edge tensors concatenated, edge_index offset by the
    cumulative node count, batch[N] = graph id."""
    from bathymetric_gnn_amd import synthetic
    from bathymetric_gnn_amd.data import GraphBuilder
    grids = synthetic.vr_grid_stream(40, seed0=1000)
    gb = GraphBuilder()
    depths = [g[0] for g in grids]; uncs = [g[1] for g in grids]; ress = [g[2] for g in grids]
    masks = [(d != 1.0e6) & np.isfinite(d) for d in depths]
    batch = gb.build_graphs(depths, masks, uncs, ress)
    os_ = [graph_cpu.build_graph(d, m, u, r) for d, m, u, r in zip(depths, masks, uncs, ress)]
    bx, bei, bea, bls, bb = graph_cpu.batch_graphs(os_)
    assert batch.num_nodes == bx.shape[0] and batch.num_edges == bei.shape[1]
    assert np.array_equal(batch.edge_index.cpu().numpy(), bei)
    assert np.array_equal(batch.batch.cpu().numpy(), bb)
    _check_x(batch.x.cpu().numpy(), bx, graph_cpu.DEFAULT_NODE_FEATURES + ["uncertainty"])
    _check_ea(batch.edge_attr.cpu().numpy(), bea)
    assert np.array_equal(batch.ptr.numpy(), np.cumsum([0] + [o.num_nodes for o in os_]))


def test_feature_selection_and_edge_feature_order(gpu_device):
    from bathymetric_gnn_amd import synthetic
    from bathymetric_gnn_amd.data import GraphBuilder
    d, m, u = synthetic.synthetic_tile(20, 24, 3, "V1", True)
    nf = ["curvature", "depth", "uncertainty", "bogus", "gradient_x"]
    ef = ["slope", "distance", "other"]
    o = graph_cpu.build_graph(d, m, u, (1.0, 0.5), node_feature_names=nf, edge_feature_names=ef)
    g = GraphBuilder(node_features=nf, edge_features=ef).build_graph(d, m, u, (1.0, 0.5))
    x = g.x.cpu().numpy(); ea = g.edge_attr.cpu().numpy()
    assert x.shape == o.x.shape == (o.num_nodes, 4)
    assert ulp_diff_f32(x, o.x).max() == 0
    assert ulp_diff_f32(ea[:, 1:], o.edge_attr[:, 1:]).max() == 0 and ulp_diff_f32(ea[:, 0], o.edge_attr[:, 0]).max() <= 1


def test_graph_to_grid_and_errors(gpu_device):
    from bathymetric_gnn_amd import synthetic
    from bathymetric_gnn_amd.data import GraphBuilder
    d, m, _ = synthetic.synthetic_tile(19, 23, 2, "V1")
    gb = GraphBuilder()
    g = gb.build_graph(d, m, None, (1.0, 1.0))
    vals = torch.arange(g.num_nodes, dtype=torch.float32)
    grid = gb.graph_to_grid(g, vals)                       # default fill NaN
    o = graph_cpu.build_graph(d, m, None, (1.0, 1.0))
    exp = graph_cpu.graph_to_grid(o, vals.numpy())
    assert grid.dtype == np.float32 and np.array_equal(np.isnan(grid), np.isnan(exp))
    assert np.array_equal(np.nan_to_num(grid), np.nan_to_num(exp))
    grid0 = gb.graph_to_grid(g.cpu(), vals, fill_value=0.0)   # CPU copy path, as models/pipeline.py:278-303
    assert np.array_equal(grid0, graph_cpu.graph_to_grid(o, vals.numpy(), 0.0))
    with pytest.raises(ValueError):
        gb.graph_to_grid(g, torch.zeros(g.num_nodes, 2))
    empty = gb.build_graph(np.full((4, 4), np.nan, np.float32))
    with pytest.raises(ValueError):
        gb.graph_to_grid(empty, torch.zeros(0))
    with pytest.raises(ValueError):
        GraphBuilder(connectivity="6-connected")
    with pytest.raises(ValueError):                        # np.gradient needs >= 2 samples (behaviour.json)
        gb.build_graph(np.zeros((1, 5), np.float32))


def test_full_tile_properties_256(gpu_device):
    """BASELINE size: size-independent properties -- E = 4(n-1)(2n-1), symmetric edge set, in-degree
    histogram of a full 8-connected tile, repeat-run determinism."""
    from bathymetric_gnn_amd import synthetic
    from bathymetric_gnn_amd.data import GraphBuilder
    n = 256
    d, m, _ = synthetic.synthetic_tile(n, n, 1, "V0")
    gb = GraphBuilder()
    g = gb.build_graph(d, m, None, (0.5, 0.5))
    assert g.num_nodes == n * n and g.num_edges == 4 * (n - 1) * (2 * n - 1)
    ei = g.edge_index
    key = ei[0] * (n * n) + ei[1]; rkey = ei[1] * (n * n) + ei[0]
    assert torch.equal(torch.sort(key).values, torch.sort(rkey).values)
    deg = torch.bincount(ei[1], minlength=n * n)
    assert int((deg == 8).sum()) == (n - 2) ** 2 and int((deg == 5).sum()) == 4 * (n - 2) and int((deg == 3).sum()) == 4
    g2 = gb.build_graph(d, m, None, (0.5, 0.5))
    assert torch.equal(g.x, g2.x) and torch.equal(g.edge_attr, g2.edge_attr) and torch.equal(ei, g2.edge_index)


def test_many_live_graphs_of_distinct_shapes_keep_their_tables(gpu_device):
    """the per-context cache of tile / work-item tables holds 8 shapes.  Graphs that are still alive pin
    their entry: building a 9th, 10th, ... shape must neither free tables a live graph points into nor hand it another
    shape's tables.  Ten graphs of distinct shapes are kept alive; the FIRST is then exported and run and must equal a
    fresh build of the same tile bit for bit."""
    from bathymetric_gnn_amd import synthetic
    from bathymetric_gnn_amd.data import GraphBuilder
    from bathymetric_gnn_amd.models import BathymetricGNN
    gb = GraphBuilder(device=gpu_device)
    shapes = [(40 + 3 * i, 64 - 2 * i) for i in range(12)]
    tiles = [synthetic.synthetic_tile(h, w, 300 + i, "V1") for i, (h, w) in enumerate(shapes)]
    live = [gb.build_graph(d, m, None, (0.5, 0.5)) for d, m, _ in tiles]            # 12 distinct shapes, all alive
    sd = synthetic.synthetic_state_dict(seed=1234)
    model = BathymetricGNN(in_channels=7, edge_dim=3, dropout=0.0)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    model = model.to(gpu_device).eval()
    for i in (0, 1, 11):
        d, m, _ = tiles[i]
        og = graph_cpu.build_graph(d, m, None, (0.5, 0.5))
        g = live[i]
        assert torch.equal(g.edge_index.cpu(), torch.from_numpy(og.edge_index))
        assert np.array_equal(g.x.cpu().numpy()[:, [0, 3, 4, 5, 6]], og.x[:, [0, 3, 4, 5, 6]])
        out_live = model.predict(g)["class_logits"].clone()
        fresh = gb.build_graph(d, m, None, (0.5, 0.5))
        assert torch.equal(out_live, model.predict(fresh)["class_logits"])
        del fresh
    # dropping the graphs unpins the entries: the same shapes can be built (and evicted) again
    del live, g
    again = [gb.build_graph(d, m, None, (0.5, 0.5)) for d, m, _ in tiles]
    og = graph_cpu.build_graph(tiles[5][0], tiles[5][1], None, (0.5, 0.5))
    assert torch.equal(again[5].edge_index.cpu(), torch.from_numpy(og.edge_index))


def test_graphs_of_a_closed_context_are_inert(gpu_device):
    """Context.close() destroys the library context while GraphData objects built on it are alive.  The
    library releases those graphs with the context; the Python objects refuse to run afterwards and their finaliser
    does not touch the dead handle."""
    import gc
    from bathymetric_gnn_amd import runtime as rt, synthetic
    from bathymetric_gnn_amd.data import GraphBuilder
    gb = GraphBuilder(device=gpu_device)
    ctx = rt.new_context(gpu_device)
    d, m, _ = synthetic.synthetic_tile(48, 40, 3, "V1")
    hw, res, dt, mt, ut = gb.upload_tiles([d], [m], None, [(0.5, 0.5)])
    graphs = [gb.build_from_device(hw, res, dt, mt, ut, ctx=ctx) for _ in range(3)]
    n = graphs[0].num_nodes
    assert n == int(m.sum()) and graphs[0].x.shape == (n, 7)
    free0 = torch.cuda.mem_get_info(gpu_device)[0]
    ctx.close()
    assert ctx.handle is None
    assert torch.cuda.mem_get_info(gpu_device)[0] >= free0          # the graphs' arenas went back with the context
    assert graphs[0].x.shape == (n, 7)                              # tensors already exported stay valid (torch owns them)
    with pytest.raises(rt.BgnnError):
        graphs[1].edge_index                                        # a new export would need the dead handle
    del graphs
    gc.collect()                                                    # finalisers run: must not call into the freed context
    g2 = gb.build_graph(d, m, None, (0.5, 0.5))                     # the default context is unaffected
    assert g2.num_nodes == n


@pytest.mark.parametrize("conn", ["8-connected", "16-dilated"])
def test_tiled_feature_kernel_is_bit_identical_to_the_cell_kernel(conn, gpu_device):
    """The LDS-tiled feature kernel computes the float64 slope of an edge once for both directions (atan is odd, the depth
    difference antisymmetric) and reads stencil operands from a staged tile; the thread-per-cell kernel (option
    features_tiled = 0) is the statement it must reproduce BIT FOR BIT: random masks, NaN / inf / nodata depths inside the
    mask, equal depths (zero slope: the sign must not flip), huge and tiny resolutions, ragged shapes that leave partial
    8 x 64 chunks, widths below and above one chunk."""
    from bathymetric_gnn_amd import runtime as rt, synthetic
    from bathymetric_gnn_amd.data import GraphBuilder
    ctx = rt.get_context(gpu_device)
    rng = np.random.default_rng(5)
    cases = []
    for (h, w), res in [((37, 150), (0.5, 0.5)), ((64, 64), (1.0, 2.0)), ((9, 300), (1000.0, 1000.0)), ((130, 70), (1e-3, 0.25)), ((5, 3), (0.5, 0.5)),
                         ((256, 256), (0.5, 0.5))]:
        d, m, _ = synthetic.synthetic_tile(h, w, int(rng.integers(1 << 30)), "V1" if min(h, w) >= 16 else "V0")
        d = d.copy(); m = m.copy()
        k = max(1, h * w // 50)
        ii = rng.integers(0, h * w, size=(6, k))
        d.flat[ii[0]] = np.nan; d.flat[ii[1]] = np.inf; d.flat[ii[2]] = -np.inf; d.flat[ii[3]] = 1.0e6
        d.flat[ii[4]] = -20.0                                   # equal depths: zero depth difference, slope +0 both ways
        d.flat[ii[5]] = d.flat[np.minimum(ii[5] + 1, h * w - 1)]
        m.flat[ii[0][: k // 2]] = True; m.flat[ii[1][: k // 2]] = True; m.flat[ii[2][: k // 2]] = True     # non-finite depths INSIDE the mask
        cases.append((d.astype(np.float32), m, res))
    gb = GraphBuilder(connectivity=conn, device=gpu_device)
    outs = {}
    for tiled in (0, 2):                     # 2: the tiled form for every shape (1, the default, keeps it to wide uniform batches)
        with ctx.options(features_tiled=tiled):
            per = []
            for d, m, res in cases:
                g = gb.build_graph(d, m, None, res)
                per.append((g.x.clone(), g.edge_attr.clone(), g.edge_index.clone(), g.local_std.clone()))
            # and one ragged batch (several grids in one build)
            g = gb.build_graphs([c[0] for c in cases[:5]], [c[1] for c in cases[:5]], None, [c[2] for c in cases[:5]])
            per.append((g.x.clone(), g.edge_attr.clone(), g.edge_index.clone(), g.local_std.clone()))
            outs[tiled] = per
    n_edges = 0
    for a, b in zip(outs[0], outs[2]):
        for ta, tb in zip(a, b):
            assert ta.shape == tb.shape
            va = ta.view(torch.int32) if ta.dtype == torch.float32 else ta
            vb = tb.view(torch.int32) if tb.dtype == torch.float32 else tb
            assert torch.equal(va, vb)
        n_edges += a[1].shape[0]
    assert n_edges > 500000
    # the zero-slope sign case really occurred: some edge has depth difference exactly 0 and slope +0 (not -0)
    ea = outs[2][0][1]
    zero = ea[:, 1] == 0
    assert int(zero.sum()) > 0 and bool((ea[zero, 2].view(torch.int32) == 0).all())


def test_box_statistics_forms_agree_bit_for_bit_and_with_the_oracle(gpu_device):
    """The 5 x 5 box statistics divide by 5.0 with a three-operation exact quotient and share one refined reciprocal between the
    two quotients by the count; the 16-wide instances (option stats_narrow = 1, picked for small launches) regroup the same
    running sums.  All of that must leave local_mean / local_std bit-identical: narrow == wide on every input (non-finite depths
    inside the mask included -- they take the general division), and both == the float64 oracle (scipy's uniform_filter order)
    on finite inputs spanning twelve decades of depth."""
    from bathymetric_gnn_amd import runtime as rt, synthetic
    from bathymetric_gnn_amd.data import GraphBuilder
    ctx = rt.get_context(gpu_device)
    rng = np.random.default_rng(11)
    feats = ["depth", "local_mean", "local_std"]
    gb = GraphBuilder(node_features=feats, device=gpu_device)
    cases = []
    for (h, w), scale in [((256, 256), 1.0), ((50, 47), 1e-4), ((19, 130), 3.0e4), ((3, 3), 1.0), ((70, 5), 977.0), ((131, 66), 1e-7)]:
        d, m, _ = synthetic.synthetic_tile(h, w, int(rng.integers(1 << 30)), "V1" if min(h, w) >= 16 else "V0")
        d = (d.astype(np.float64) * scale).astype(np.float32)
        d.flat[rng.integers(0, h * w, size=max(1, h * w // 40))] *= np.float32(1.0e3)     # outliers: cancelling variances
        cases.append((d, m.copy(), True))
    d, m, _ = synthetic.synthetic_tile(90, 200, 3, "V1")
    d = d.copy(); m = m.copy()
    ii = rng.integers(0, d.size, size=(3, 60))
    d.flat[ii[0]] = np.inf; d.flat[ii[1]] = -np.inf; d.flat[ii[2]] = np.nan
    m.flat[ii.ravel()] = True
    cases.append((d, m, False))                                                           # identity of the two forms only
    outs = {}
    for narrow in (0, 1):
        with ctx.options(stats_narrow=narrow):
            per = [gb.build_graph(d, m, None, (0.5, 0.5)).x.clone() for d, m, _ in cases]
            per.append(gb.build_graphs([c[0] for c in cases], [c[1] for c in cases], None, [(0.5, 0.5)] * len(cases)).x.clone())
            outs[narrow] = per
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    for (d, m, finite), x in zip(cases, outs[1]):
        if not finite:
            continue
        o = graph_cpu.build_graph(d, m, None, (0.5, 0.5), node_feature_names=feats)
        _check_x(x.cpu().numpy(), o.x, feats)
        assert ulp_diff_f32(x.cpu().numpy()[:, 1], o.x[:, 1]).max(initial=0) == 0       # local_mean: 0 ulp


def test_box_statistics_shape_sweep(gpu_device):
    """Chunk boundaries of the running-sum kernels: heights / widths of 2..5 cells (windows that never fill; a side of 1 is refused
    like np.gradient refuses it in the reference), 16 k - 1 .. 16 k + 3
    (outputs 1 + 16 i + j: the last chunk holds 15, 16, 1, 2 outputs), 64 and 65 (a second 64-wide workgroup with one row).  Every
    shape alone and all of them as one ragged batch, both workgroup widths, local_mean 0 ulp / local_std <= 1 ulp against the oracle."""
    from bathymetric_gnn_amd import runtime as rt
    from bathymetric_gnn_amd.data import GraphBuilder
    ctx = rt.get_context(gpu_device)
    rng = np.random.default_rng(23)
    feats = ["depth", "local_mean", "local_std"]
    gb = GraphBuilder(node_features=feats, device=gpu_device)
    sizes = [2, 3, 4, 5, 15, 16, 17, 18, 19, 31, 32, 33, 34, 35, 47, 48, 49, 50, 51, 64, 65]
    shapes = [(int(rng.choice(sizes)), int(rng.choice(sizes))) for _ in range(40)] + [(2, 2), (2, 65), (65, 2), (2, 33), (33, 2), (17, 17), (65, 65)]
    cases = []
    for h, w in shapes:
        d = (rng.normal(-30.0, 8.0, size=(h, w)) * rng.choice([1.0, 1e-3, 250.0])).astype(np.float32)
        m = rng.random((h, w)) < rng.choice([1.0, 0.9, 0.5])
        if not m.any():
            m[0, 0] = True
        cases.append((d, m))
    oracle = [graph_cpu.build_graph(d, m, None, (0.5, 0.5), node_feature_names=feats).x for d, m in cases]
    for narrow in (0, 1):
        with ctx.options(stats_narrow=narrow):
            for (d, m), o in zip(cases, oracle):
                _check_x(gb.build_graph(d, m, None, (0.5, 0.5)).x.cpu().numpy(), o, feats)
            xb = gb.build_graphs([c[0] for c in cases], [c[1] for c in cases], None, [(0.5, 0.5)] * len(cases)).x.cpu().numpy()
            _check_x(xb, np.concatenate(oracle, axis=0), feats)
            assert ulp_diff_f32(xb[:, 1], np.concatenate(oracle, axis=0)[:, 1]).max(initial=0) == 0
