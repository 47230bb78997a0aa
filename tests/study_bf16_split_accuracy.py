#!/usr/bin/env python3
"""How accurate would the forward be if the dense products ran on the bf16 matrix pipe through operand splitting?
(Design study for the next round: on gfx950 the f32 MFMA runs at the VALU rate and blocks its SIMD neighbour --
tools/mfma_overlap_probe.hip -- while the bf16 MFMA is 16x faster per instruction and co-executes with VALU work.)

Every GATConv / head / extractor matrix product x @ W^T is replaced by a sum of bf16 x bf16 products with float32
accumulation (bf16 x bf16 is exact in float32, as on the matrix cores):
  x3: x = xh + xl, W = Wh + Wl          -> xh Wh + xh Wl + xl Wh                 (3 MFMAs per k-step)
  x6: x = xh + xm + xl, W likewise      -> hh + hm + mh + hl + lh + mm           (6 MFMAs per k-step)
and the class logits are compared with the float64 forward, next to the plain float32 forward."""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bathymetric_gnn_amd import synthetic
from oracle import gat_cpu, graph_cpu

def split(t, parts):
    out, r = [], t.clone()
    for _ in range(parts):
        h = r.to(torch.bfloat16).to(torch.float32)
        out.append(h); r = r - h
    return out

MODE = {"n": 0}
_orig_linear = F.linear
def linear(x, w, b=None):
    n = MODE["n"]
    if n == 0 or x.dtype != torch.float32:
        return _orig_linear(x, w, b)
    xs, ws = split(x, 2 if n == 3 else 3), split(w, 2 if n == 3 else 3)
    pairs = [(0, 0), (0, 1), (1, 0)] if n == 3 else [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]
    acc = torch.zeros(x.shape[0], w.shape[0], dtype=torch.float32)
    for i, j in reversed(pairs):                       # small terms first
        acc = acc + xs[i] @ ws[j].T
    return acc if b is None else acc + b
F.linear = linear

sd = synthetic.synthetic_state_dict(in_channels=7, seed=1234)
rows = []
for seed, (h, w) in enumerate([(64, 64), (96, 80), (128, 128)]):
    d, m, _ = synthetic.synthetic_tile(h, w, 100 + seed, "V1")
    g = graph_cpu.build_graph(d, m, None, (0.5, 0.5))
    MODE["n"] = 0
    ref = gat_cpu.forward(sd, g.x, g.edge_index, g.edge_attr, dtype=torch.float64)["class_logits"]
    res = {}
    for name, n in (("float32", 0), ("bf16 x3", 3), ("bf16 x6", 6)):
        MODE["n"] = n
        out = gat_cpu.forward(sd, g.x, g.edge_index, g.edge_attr)["class_logits"].double()
        res[name] = float((out - ref).abs().max())
    rows.append((h, w, g.x.shape[0], res))
    print(f"{h}x{w} N={g.x.shape[0]}: max |logit - float64| :", {k: f"{v:.2e}" for k, v in res.items()})
