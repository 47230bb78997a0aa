#!/usr/bin/env python3
"""Stress HostTilePipeline against the blocking path (race hunt)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bathymetric_gnn_amd import synthetic
from bathymetric_gnn_amd.data import GraphBuilder
from bathymetric_gnn_amd.models import BathymetricGNN
from bathymetric_gnn_amd.models.pipeline import HostTilePipeline, TileBatchEngine
dev = torch.device("cuda:0")
sd = synthetic.synthetic_state_dict(in_channels=8, seed=1234)
m = BathymetricGNN(in_channels=8, edge_dim=3, dropout=0.0); m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}); m.to(dev).eval()
eng = TileBatchEngine(m, GraphBuilder(device=dev), dev)
n, h, w = 3, 40, 56
batches = [synthetic.synthetic_tile_batch(n, h, w, 500 + 10 * i, "V1", True) for i in range(12)]
refs = [eng.infer(list(d), list(mk), list(u), [(0.5, 1.0)] * n) for d, mk, u in batches]
refs2 = [eng.infer(list(d), list(mk), list(u), [(0.5, 1.0)] * n) for d, mk, u in batches]
bad_ref = sum(not np.array_equal(a[k][c], b[k][c]) for a, b in zip(refs, refs2) for k in range(n) for c in a[k])
print("blocking path self-consistency mismatches:", bad_ref)
bad = 0
# warm other code paths first, as the test suite does (generic graphs, other shapes)
for hh, ww in ((64, 64), (17, 23), (256, 256)):
    dd, mm, uu = synthetic.synthetic_tile_batch(2, hh, ww, 7, "V1", True)
    eng.infer(list(dd), list(mm), list(uu), [(0.5, 0.5)] * 2)
for rep in range(60):
    hp = HostTilePipeline(eng, n, h, w, with_uncertainty=True, resolution=(0.5, 1.0))
    got = []
    for i, (d, mk, u) in enumerate(batches):
        r = hp.submit(d, mk, u, tag=i)
        if r is not None:
            got.append((r[0], {k: v.copy() for k, v in r[1].items()}))
    got += [(t, {k: v.copy() for k, v in r.items()}) for t, r in hp.drain()]
    after = [eng.infer(list(d), list(mk), list(u), [(0.5, 1.0)] * n) for d, mk, u in batches[:5]]
    for t in range(5):
        for k in range(n):
            for ch in ("classification", "confidence", "correction"):
                if not np.array_equal(after[t][k][ch], refs[t][k][ch]):
                    bad += 1; print("rep", rep, "BLOCKING path after pipeline differs: batch", t, "tile", k, ch)
    for (t, r) in got:
        for k in range(n):
            for ch in ("classification", "confidence", "correction"):
                if not np.array_equal(r[ch][k].view(np.uint32), refs[t][k][ch].view(np.uint32)):
                    bad += 1
                    # which tile does it look like?
                    who = [(tt, kk) for tt in range(len(refs)) for kk in range(n) if np.array_equal(r[ch][k], refs[tt][kk][ch])]
                    print("rep", rep, "batch", t, "tile", k, ch, "mismatch; equals", who[:3])
print("pipeline mismatches:", bad)
