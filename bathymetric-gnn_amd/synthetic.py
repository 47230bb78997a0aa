"""Seeded synthetic inputs and weights.

There is no network for real surveys or checkpoints, so benchmarks and tests
use tiles of the shape the reference processes and random-init weights of the
reference architecture with the upstream parameter names
(``models/gnn.py`` + torch_geometric ``GATConv``/``BatchNorm`` key names).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional, Tuple

import numpy as np

NODATA = 1.0e6


def synthetic_tile(h: int, w: int, seed: int, variant: str = "V0",
                   with_uncertainty: bool = False):
    """One synthetic depth tile.

    depth[r,c] = -20 - 0.01 c - 0.005 r + 0.5 sin(2 pi r/37) cos(2 pi c/53) + 0.05 N(0,1),
    0.5 % of cells spiked by +-U(1,5).  ``V0``: all valid.  ``V1``: 5 % iid invalid
    plus one 16x16 hole (clipped to the tile), nodata 1e6 written into invalid cells.
    Returns (depth f32 [h,w], valid_mask bool [h,w], uncertainty f32 [h,w] | None).
    """
    rng = np.random.default_rng(seed)
    r = np.arange(h, dtype=np.float64)[:, None]
    c = np.arange(w, dtype=np.float64)[None, :]
    depth = (-20.0 - 0.01 * c - 0.005 * r
             + 0.5 * np.sin(2 * np.pi * r / 37.0) * np.cos(2 * np.pi * c / 53.0)
             + 0.05 * rng.standard_normal((h, w)))
    spikes = rng.random((h, w)) < 0.005
    mag = rng.uniform(1.0, 5.0, size=(h, w)) * rng.choice([-1.0, 1.0], size=(h, w))
    depth = np.where(spikes, depth + mag, depth).astype(np.float32)
    mask = np.ones((h, w), dtype=bool)
    if variant == "V1":
        mask &= rng.random((h, w)) >= 0.05
        hh, hw_ = min(16, h), min(16, w)
        r0 = int(rng.integers(0, h - hh + 1)); c0 = int(rng.integers(0, w - hw_ + 1))
        mask[r0:r0 + hh, c0:c0 + hw_] = False
        depth = np.where(mask, depth, np.float32(NODATA)).astype(np.float32)
    elif variant != "V0":
        raise ValueError(variant)
    unc = None
    if with_uncertainty:
        unc = rng.uniform(0.05, 0.3, size=(h, w)).astype(np.float32)
    return depth, mask, unc


def synthetic_tile_batch(n: int, h: int, w: int, seed0: int, variant: str = "V0",
                         with_uncertainty: bool = False):
    ds, ms, us = [], [], []
    for i in range(n):
        d, m, u = synthetic_tile(h, w, seed0 + i, variant, with_uncertainty)
        ds.append(d); ms.append(m); us.append(u)
    depth = np.stack(ds); mask = np.stack(ms)
    unc = np.stack(us) if with_uncertainty else None
    return depth, mask, unc


def synthetic_survey_device(size: int, device, seed: int = 0, band: int = 4096, nodata_corner: bool = True, rows=None):
    """BASELINE config 5's survey, generated ON the device (numpy would need minutes for 3.6 G cells): a ``size`` x ``size``
    float32 depth field by the SURVEY 8(d) formula (noise from torch's generator), in row bands so that no survey-sized
    temporary exists; ``nodata_corner`` writes 1e6 into the top-left tenth x eighth (tiles skipped by ``min_valid_ratio``,
    a ragged edge).  The field is procedural in the ABSOLUTE row band (one generator seed per band of ``band`` rows), so
    ``rows=(lo, hi)`` returns exactly the rows ``[lo, hi)`` of the full survey without generating the rest -- what a rank of
    a row-band sharded run holds.  Returns (depth [hi-lo, S] f32, valid [hi-lo, S] bool), both resident in HBM."""
    import torch
    S = int(size)
    lo, hi = (0, S) if rows is None else (int(rows[0]), int(rows[1]))
    assert 0 <= lo <= hi <= S
    g = torch.Generator(device=device)
    depth = torch.empty((hi - lo, S), dtype=torch.float32, device=device)
    c = torch.arange(S, dtype=torch.float32, device=device)[None, :]
    for b in range(lo // band, (hi + band - 1) // band if hi > lo else 0):
        r0, r1 = b * band, min(S, (b + 1) * band)
        g.manual_seed(seed * 1000003 + b)
        r = torch.arange(r0, r1, dtype=torch.float32, device=device)[:, None]
        d = -20 - 0.01 * c - 0.005 * r + 0.5 * torch.sin(2 * np.pi * r / 37) * torch.cos(2 * np.pi * c / 53)
        d += 0.05 * torch.randn((r1 - r0, S), generator=g, device=device)
        a, e = max(r0, lo), min(r1, hi)
        depth[a - lo:e - lo] = d[a - r0:e - r0]
        del d
    if nodata_corner and lo < S // 10:
        depth[: min(hi, S // 10) - lo, : S // 8] = NODATA
    valid = (depth != NODATA) & torch.isfinite(depth)
    return depth, valid


def vr_grid_stream(n: int, seed0: int = 1000, lo: int = 3, hi: int = 50):
    """BASELINE config 4: refinement grids with dims iid uniform on {lo..hi}^2,
    uncertainty U(0.05,0.3), mask variant V1-like 3 % invalid."""
    out = []
    for i in range(n):
        rng = np.random.default_rng(seed0 + i)
        h = int(rng.integers(lo, hi + 1)); w = int(rng.integers(lo, hi + 1))
        d, m, u = synthetic_tile(h, w, seed0 + i, "V0", True)
        inval = rng.random((h, w)) < 0.03
        d = np.where(inval, np.float32(NODATA), d).astype(np.float32)
        res = float(rng.choice([0.5, 1.0, 2.0, 4.0]))
        out.append((d, u, (res, res)))
    return out


def synthetic_vr_bag(base_rows: int, base_cols: int, seed: int = 1000, lo: int = 3, hi: int = 50,
                     refined_fraction: float = 0.85, empty_fraction: float = 0.02, sparse_fraction: float = 0.03):
    """A synthetic VR BAG as its two HDF5 arrays (``data/vr_bag.py`` formats): ``varres_metadata``
    [base_rows, base_cols] and ``varres_refinements`` (1, N).  A base cell is refined with probability
    ``refined_fraction``; its grid is a ``vr_grid_stream`` grid (dims iid on {lo..hi}^2, 3 % nodata), entirely
    nodata with probability ``empty_fraction`` and 99.5 % nodata with ``sparse_fraction`` (what
    ``min_valid_ratio`` filters).  Records are laid out in base-grid row-major order, as BAG writers do."""
    from .data.vr_bag import VARRES_METADATA_DTYPE, VARRES_REFINEMENT_DTYPE
    rng = np.random.default_rng(seed)
    md = np.zeros((base_rows, base_cols), VARRES_METADATA_DTYPE)
    md["index"] = 4294967295
    recs = []
    pos = 0
    k = 0
    for r in range(base_rows):
        for c in range(base_cols):
            if rng.random() >= refined_fraction:
                continue
            (d, u, res), = vr_grid_stream(1, seed0=seed + 7919 * (k + 1), lo=lo, hi=hi)
            k += 1
            roll = rng.random()
            if roll < empty_fraction:
                d = np.full_like(d, np.float32(NODATA))
            elif roll < empty_fraction + sparse_fraction:
                keep = rng.random(d.shape) < 0.005
                d = np.where(keep, d, np.float32(NODATA)).astype(np.float32)
            h, w = d.shape
            md[r, c] = (pos, w, h, res[0], res[1], 0.5 * res[0], 0.5 * res[1])
            rec = np.empty(h * w, VARRES_REFINEMENT_DTYPE)
            rec["depth"] = d.ravel(); rec["depth_uncrt"] = u.ravel()
            recs.append(rec)
            pos += h * w
    ref = (np.concatenate(recs) if recs else np.empty(0, VARRES_REFINEMENT_DTYPE))[None, :]
    return md, ref


def _glorot(rng, shape, fan_in, fan_out):
    a = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-a, a, size=shape).astype(np.float32)


def synthetic_state_dict(in_channels: int = 7, hidden: int = 64, num_layers: int = 4,
                         heads: int = 4, num_classes: int = 3, edge_dim: int = 3,
                         predict_correction: bool = True, seed: int = 1234,
                         legacy_lin_src: bool = False, gnn_type: str = "GAT") -> "OrderedDict[str, np.ndarray]":
    """Random-init weights under the reference's state_dict key names
    (``training/trainer.py:809-829`` saves ``model.state_dict()``).

    glorot-uniform lin / lin_edge / att_*, small random biases, NON-trivial BatchNorm
    running statistics so that BN folding is exercised (SURVEY 8(c) 'Weights for tests').
    ``legacy_lin_src``: emit ``lin_src.weight`` + ``lin_dst.weight`` (older
    torch_geometric releases) instead of ``lin.weight``.
    """
    rng = np.random.default_rng(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def linear(prefix, out_f, in_f):
        b = 1.0 / np.sqrt(in_f)
        sd[prefix + ".weight"] = rng.uniform(-b, b, size=(out_f, in_f)).astype(np.float32)
        sd[prefix + ".bias"] = rng.uniform(-b, b, size=(out_f,)).astype(np.float32)

    linear("feature_extractor.mlp.0", hidden, in_channels)
    linear("feature_extractor.mlp.3", hidden, hidden)
    def batch_norm(l, width):
        q = f"gnn.norms.{l}.module."
        sd[q + "weight"] = rng.uniform(0.8, 1.2, size=width).astype(np.float32)
        sd[q + "bias"] = (0.05 * rng.standard_normal(width)).astype(np.float32)
        sd[q + "running_mean"] = (0.1 * rng.standard_normal(width)).astype(np.float32)
        sd[q + "running_var"] = rng.uniform(0.5, 1.5, size=width).astype(np.float32)
        sd[q + "num_batches_tracked"] = np.array(100, dtype=np.int64)

    for l in range(num_layers if gnn_type != "GAT" else 0):      # GCN / GraphSAGE / GIN: every layer hidden -> hidden
        p = f"gnn.convs.{l}."
        if gnn_type == "GCN":
            sd[p + "lin.weight"] = _glorot(rng, (hidden, hidden), hidden, hidden)
            sd[p + "bias"] = (0.05 * rng.standard_normal(hidden)).astype(np.float32)
        elif gnn_type == "GraphSAGE":
            linear(p + "lin_l", hidden, hidden)
            sd[p + "lin_r.weight"] = _glorot(rng, (hidden, hidden), hidden, hidden)
        elif gnn_type == "GIN":
            linear(p + "nn.0", hidden, hidden)
            linear(p + "nn.2", hidden, hidden)
        else:
            raise ValueError(f"Unknown GNN type: {gnn_type}")
        batch_norm(l, hidden)
    for l in range(num_layers if gnn_type == "GAT" else 0):
        last = l == num_layers - 1
        H = 1 if last else heads
        d_in = hidden if l == 0 else hidden * heads
        p = f"gnn.convs.{l}."
        w = _glorot(rng, (H * hidden, d_in), d_in, H * hidden)
        if legacy_lin_src:
            sd[p + "lin_src.weight"] = w
            sd[p + "lin_dst.weight"] = w.copy()
        else:
            sd[p + "lin.weight"] = w
        for nm in ("att_src", "att_dst", "att_edge"):
            sd[p + nm] = _glorot(rng, (1, H, hidden), H, hidden)
        sd[p + "lin_edge.weight"] = _glorot(rng, (H * hidden, edge_dim), edge_dim, H * hidden)
        width = hidden if last else H * hidden
        sd[p + "bias"] = (0.05 * rng.standard_normal(width)).astype(np.float32)
        q = f"gnn.norms.{l}.module."
        sd[q + "weight"] = rng.uniform(0.8, 1.2, size=width).astype(np.float32)
        sd[q + "bias"] = (0.05 * rng.standard_normal(width)).astype(np.float32)
        sd[q + "running_mean"] = (0.1 * rng.standard_normal(width)).astype(np.float32)
        sd[q + "running_var"] = rng.uniform(0.5, 1.5, size=width).astype(np.float32)
        sd[q + "num_batches_tracked"] = np.array(100, dtype=np.int64)
    half = hidden // 2
    linear("classification_head.mlp.0", half, hidden)
    linear("classification_head.mlp.3", num_classes, half)
    linear("confidence_head.mlp.0", half, hidden)
    linear("confidence_head.mlp.3", 1, half)
    if predict_correction:
        linear("correction_head.mlp.0", half, hidden)
        linear("correction_head.mlp.3", 1, half)
    return sd
