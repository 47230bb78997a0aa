#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

What runs: the reference's own ``data/graph_construction.py`` (all arithmetic is the
reference's numpy/scipy code) and ``data/tiling.py``, loaded by path.  The only
stand-in is an in-memory attribute bag for ``torch_geometric.data.Data`` (a
container, no arithmetic) because torch_geometric is not installed here
(SURVEY.md section 8(c)).  Nothing of the reference's source is copied: the
fixtures hold inputs and expected outputs only.
"""
import hashlib
import importlib.util
import json
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class Data:  # container stand-in only
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def num_nodes(self):
        return self.x.shape[0]

    @property
    def num_edges(self):
        return self.edge_index.shape[1]


tg = types.ModuleType("torch_geometric")
tgd = types.ModuleType("torch_geometric.data")
tgd.Data = Data
tg.data = tgd
sys.modules["torch_geometric"] = tg
sys.modules["torch_geometric.data"] = tgd

ref_gc = _load("ref_graph_construction", os.path.join(REF, "data/graph_construction.py"))
synth = _load("bgnn_synthetic", os.path.join(ROOT, "bathymetric-gnn_amd/synthetic.py"))


def run_case(name, depth, mask, unc, resolution, connectivity="8-connected",
             self_loops=False, store_full=True):
    gb = ref_gc.GraphBuilder(connectivity=connectivity, include_self_loops=self_loops)
    g = gb.build_graph(depth, mask, unc, resolution)
    out = dict(
        depth=depth, mask=(np.isfinite(depth) if mask is None else mask).astype(np.uint8),
        mask_given=np.array(mask is not None), resolution=np.array(resolution, dtype=np.float64),
        connectivity=np.array(connectivity), self_loops=np.array(self_loops),
    )
    if unc is not None:
        out["unc"] = unc
    x = g.x.numpy(); ei = g.edge_index.numpy(); ea = g.edge_attr.numpy()
    out["num_nodes"] = np.array(g.num_nodes); out["num_edges"] = np.array(g.num_edges)
    out["edge_index_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(ei).tobytes()).hexdigest())
    if store_full:
        out.update(x=x, edge_index=ei.astype(np.int32), edge_attr=ea, pos=g.pos.numpy(),
                   local_std=g.local_std.numpy())
        if g.num_nodes:
            out.update(valid_rows=g.valid_rows.numpy().astype(np.int32),
                       valid_cols=g.valid_cols.numpy().astype(np.int32))
    else:  # big case: hashes + head/tail rows
        out.update(x_head=x[:1024], x_tail=x[-1024:], ea_head=ea[:1024], ea_tail=ea[-1024:],
                   x_sha256=np.array(hashlib.sha256(x.tobytes()).hexdigest()),
                   ea_sha256=np.array(hashlib.sha256(ea.tobytes()).hexdigest()),
                   local_std_sha256=np.array(hashlib.sha256(g.local_std.numpy().tobytes()).hexdigest()))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: N={g.num_nodes} E={g.num_edges}")


def main():
    rng = np.random.default_rng(7)
    # G1: 8x8 with holes, a 1e6 nodata cell and a NaN (mask derived by the caller like
    #     NativeVRProcessor does: (depth != nodata) & isfinite)
    d = (-15 + rng.standard_normal((8, 8))).astype(np.float32)
    d[2, 3] = 1.0e6; d[5, 5] = np.nan; d[0, 7] = 1.0e6
    m = (d != 1.0e6) & np.isfinite(d)
    run_case("G1_8x8_holes", d, m, None, (1.0, 1.0))
    # G1b: same but mask=None (reference default isfinite -> the 1e6 cells are VALID nodes)
    run_case("G1b_8x8_maskNone", d, None, None, (1.0, 1.0))
    # G2: 16x16 random 20 % invalid incl. border
    d, m, _ = synth.synthetic_tile(16, 16, 2, "V0")
    m = rng.random((16, 16)) >= 0.2
    d = np.where(m, d, np.float32(1.0e6)).astype(np.float32)
    run_case("G2_16x16_rand20", d, m, None, (0.5, 0.5))
    # G3: 64x64 all-valid (BASELINE config 1)
    d, m, _ = synth.synthetic_tile(64, 64, 0, "V0")
    run_case("G3_64x64_full", d, m, None, (0.5, 0.5))
    # G4: 50x37 non-square, uncertainty, anisotropic resolution, V1 mask
    d, m, u = synth.synthetic_tile(50, 37, 4, "V1", True)
    run_case("G4_50x37_unc_aniso", d, m, u, (0.5, 1.0))
    # G5: tiny grids
    d = (np.arange(9, dtype=np.float32).reshape(3, 3)) ** 1.5
    run_case("G5_3x3", d.astype(np.float32), np.ones((3, 3), bool), None, (0.5, 1.0))
    d = np.array([[-3.0, -4.0], [-5.0, 1.0e6]], dtype=np.float32)
    run_case("G5_2x2_one_invalid", d, d != 1.0e6, None, (1.0, 1.0))
    # G6: 4-connected, and self loops
    d, m, _ = synth.synthetic_tile(24, 40, 6, "V1")
    run_case("G6_24x40_4conn", d, m, None, (1.0, 1.0), connectivity="4-connected")
    d = (np.arange(9, dtype=np.float32).reshape(3, 3)) ** 1.5
    run_case("G6_3x3_4conn_selfloops", d.astype(np.float32), np.ones((3, 3), bool), None,
             (1.0, 1.0), connectivity="4-connected", self_loops=True)
    # G7: cancellation cases for local_std
    run_case("G7_flat", np.full((32, 32), -20.0, np.float32), np.ones((32, 32), bool), None, (1.0, 1.0))
    r = np.arange(40, dtype=np.float64)[:, None]; c = np.arange(48, dtype=np.float64)[None, :]
    run_case("G7_deep_ramp", (-4000.0 - 0.01 * c - 0.02 * r).astype(np.float32),
             np.ones((40, 48), bool), None, (2.0, 2.0))
    # A2: 4x4 with row 1 / col 1 invalid + uncertainty (isolated node 0)
    d = (-10 - np.arange(16, dtype=np.float32).reshape(4, 4) * 0.25).astype(np.float32)
    m = np.ones((4, 4), bool); m[1, :] = False; m[:, 1] = False
    d = np.where(m, d, np.float32(1.0e6)).astype(np.float32)
    u = rng.uniform(0.05, 0.3, (4, 4)).astype(np.float32)
    run_case("A2_4x4_isolated", d, m, u, (1.0, 1.0))
    # all-invalid tile
    run_case("A2_all_nan", np.full((5, 5), np.nan, np.float32), None, None, (1.0, 1.0))
    # C2: 256x256 V1 and V0 -- hashes + head/tail only
    d, m, _ = synth.synthetic_tile(256, 256, 1, "V0")
    run_case("C2_256_V0", d, m, None, (0.5, 0.5), store_full=False)
    d, m, _ = synth.synthetic_tile(256, 256, 1, "V1")
    run_case("C2_256_V1", d, m, None, (0.5, 0.5), store_full=False)

    # 1xN raises in the reference (np.gradient needs >= 2 samples): record the behaviour
    try:
        ref_gc.GraphBuilder().build_graph(np.zeros((1, 5), np.float32))
        raised = ""
    except Exception as e:  # noqa
        raised = type(e).__name__
    json.dump({"build_graph_1xN_raises": raised}, open(os.path.join(HERE, "behaviour.json"), "w"))
    print("1xN ->", raised)


if __name__ == "__main__":
    main()
