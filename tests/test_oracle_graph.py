"""Pins oracle/graph_cpu.py to the reference: golden vectors produced by the reference's
own GraphBuilder (tests/golden/make_golden.py) and the known answers of SURVEY Appendix A."""
import hashlib

import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle import graph_cpu


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_golden(name):
    g = load_golden(name)
    o = graph_cpu.build_graph(g["depth"], g["mask_arg"], g["unc_arg"], g["res"],
                              connectivity=g["conn"], include_self_loops=g["loops"])
    assert o.num_nodes == int(g["num_nodes"])
    assert o.num_edges == int(g["num_edges"])
    # integer work: bit-exact
    assert hashlib.sha256(np.ascontiguousarray(o.edge_index).tobytes()).hexdigest() == str(g["edge_index_sha256"])
    if "x" in g:
        assert np.array_equal(o.edge_index, g["edge_index"].astype(np.int64))
        # the oracle issues the same numpy/scipy calls as the reference: bit-exact floats too
        assert np.array_equal(o.x.view(np.uint32), g["x"].view(np.uint32))
        assert np.array_equal(o.edge_attr.view(np.uint32), g["edge_attr"].view(np.uint32))
        assert np.array_equal(o.local_std.view(np.uint32), g["local_std"].view(np.uint32))
        assert np.array_equal(o.pos, g["pos"])
        if o.num_nodes:
            assert np.array_equal(o.valid_rows, g["valid_rows"])
            assert np.array_equal(o.valid_cols, g["valid_cols"])
    else:
        assert hashlib.sha256(o.x.tobytes()).hexdigest() == str(g["x_sha256"])
        assert hashlib.sha256(o.edge_attr.tobytes()).hexdigest() == str(g["ea_sha256"])
        assert hashlib.sha256(o.local_std.tobytes()).hexdigest() == str(g["local_std_sha256"])


def test_appendix_a1_known_answer():
    # 3x3 all-valid, depth=(3r+c)^1.5, resolution (0.5, 1.0)
    d = (np.arange(9, dtype=np.float32).reshape(3, 3)) ** 1.5
    o = graph_cpu.build_graph(d.astype(np.float32), resolution=(0.5, 1.0))
    src = "4 5 7 8 3 4 5 6 7 8 3 4 6 7 1 2 4 5 7 8 0 1 3 4 6 7 1 2 4 5 0 1 2 3 4 5 0 1 3 4"
    tgt = "0 1 3 4 0 1 2 3 4 5 1 2 4 5 0 1 3 4 6 7 1 2 4 5 7 8 3 4 6 7 3 4 5 6 7 8 4 5 7 8"
    assert o.edge_index[0].tolist() == [int(v) for v in src.split()]
    assert o.edge_index[1].tolist() == [int(v) for v in tgt.split()]
    np.testing.assert_allclose(o.edge_attr[0], [1.1180340, -8.0, -82.044197], rtol=1e-6)
    assert o.edge_attr[4, 0] == 1.0


def test_full_tile_edge_count_and_symmetry():
    n = 20
    d = np.random.default_rng(0).standard_normal((n, n)).astype(np.float32)
    o = graph_cpu.build_graph(d)
    assert o.num_edges == 4 * (n - 1) * (2 * n - 1)
    fwd = set(map(tuple, o.edge_index.T.tolist()))
    assert all((t, s) in fwd for s, t in fwd)


def test_empty_graph_has_no_grid_shape():
    o = graph_cpu.build_graph(np.full((4, 4), np.nan, np.float32))
    assert o.num_nodes == 0 and o.x.shape == (0, 7) and o.edge_index.shape == (2, 0)
    assert o.grid_shape is None
    with pytest.raises(ValueError):
        graph_cpu.graph_to_grid(o, np.zeros(0))


def test_unknown_connectivity():
    with pytest.raises(ValueError):
        graph_cpu.build_graph(np.zeros((3, 3), np.float32), connectivity="6-connected")
