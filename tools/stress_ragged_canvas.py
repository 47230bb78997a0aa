#!/usr/bin/env python3
"""Race hunt for the ragged-batch paths: many random refinement-grid batches, each inferred with the canvas walk and with per-grid
blocks (exact path: bit-identical required), on two library contexts in flight at once, plus the bf16 path against the exact one."""
import argparse, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bathymetric_gnn_amd import runtime as rt, synthetic
from bathymetric_gnn_amd.data import GraphBuilder
from bathymetric_gnn_amd.models import BathymetricGNN
from bathymetric_gnn_amd.models.pipeline import TileBatchEngine

ap = argparse.ArgumentParser(); ap.add_argument("--rounds", type=int, default=120); ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
dev = torch.device("cuda:0")
sd = synthetic.synthetic_state_dict(in_channels=8, seed=1234)
model = BathymetricGNN(in_channels=8, edge_dim=3, dropout=0.0); model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}); model.to(dev).eval()
rng = np.random.default_rng(args.seed)
bad = 0; t0 = time.time(); nodes = 0
for rnd in range(args.rounds):
    conn = ["8-connected", "4-connected", "16-dilated"][rnd % 3]
    gb = GraphBuilder(device=dev, connectivity=conn)
    engs = [TileBatchEngine(model, gb, dev), TileBatchEngine(model, gb, dev, ctx=rt.new_context(dev))]
    n = int(rng.integers(2, 160))
    shapes = [(int(rng.integers(2, 51)), int(rng.integers(2, 51))) for _ in range(n)]
    grids = [synthetic.synthetic_tile(h, w, int(rng.integers(1 << 30)), "V1" if min(h, w) >= 20 and rng.random() < 0.5 else "V0") for h, w in shapes]
    depth = [g[0] for g in grids]; mask = [g[1] & (rng.random(g[1].shape) > rng.choice([0.0, 0.02, 0.3])) for g in grids]
    unc = [np.abs(g[0]) * 0.01 for g in grids]
    hw, res, d, m, u = gb.upload_tiles(depth, mask, unc, [(0.7, 0.9)] * n)
    outs = {}
    for atlas in (1, 0):
        for e in engs:
            e.ctx.set_option("ragged_atlas", atlas)
        o = [e.infer_device(hw, res, d, m, u, defer_end=True) for e in engs]      # both contexts in flight together
        for e in engs:
            e.ctx.end()
        torch.cuda.synchronize()
        outs[atlas] = o
    ok = torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][0]) and torch.equal(outs[0][1], outs[0][0])
    engs[0].ctx.set_option("matrix_path", "bf16"); engs[0].ctx.set_option("ragged_atlas", 1)
    b = engs[0].infer_device(hw, res, d, m, u)
    engs[0].ctx.set_option("matrix_path", "exact_f32")
    err = float((b[1] - outs[0][0][1]).abs().max())
    ok = ok and err < 5e-2 and bool(torch.isfinite(b).all())
    nodes += int(sum(int(x.sum()) for x in mask))
    if not ok:
        bad += 1
        print("MISMATCH round", rnd, conn, n, "bf16 conf err", err)
print(f"{args.rounds} rounds, {nodes} nodes, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
