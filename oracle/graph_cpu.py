"""CPU oracle: grid -> graph (restates reference ``data/graph_construction.py``).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Vectorised numpy/scipy; the
reference's two Python loops (``_build_edges`` list building ``:223`` and the
per-edge feature loop ``:342-369``) are replaced by array expressions that
perform the *same* scalar operations in the same dtypes, so results are
expected bit-identical to the reference (pinned by ``tests/golden``).

Every function cites the reference lines it follows
(``/root/reference/data/graph_construction.py``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
from scipy import ndimage

DEFAULT_NODE_FEATURES = [  # graph_construction.py:60-68
    "depth", "local_mean", "local_std", "gradient_x", "gradient_y",
    "gradient_magnitude", "curvature",
]
DEFAULT_EDGE_FEATURES = ["distance", "depth_difference", "slope"]  # :71-75

OFFSETS = {  # graph_construction.py:78-87
    "4-connected": [(-1, 0), (1, 0), (0, -1), (0, 1)],
    "8-connected": [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)],
}
# NOT in the reference (BASELINE config 3 "k=16" has no counterpart there): the build-side
# dilated extension = the 8 base offsets followed by the same offsets x2.  No golden vectors can exist.
OFFSETS["16-dilated"] = OFFSETS["8-connected"] + [(2 * r, 2 * c) for r, c in OFFSETS["8-connected"]]


@dataclass
class GraphCPU:
    """Attribute bag with the fields the reference attaches to ``Data`` (:150-167)."""
    x: np.ndarray                 # [N, F] float32
    edge_index: np.ndarray        # [2, E] int64
    edge_attr: np.ndarray         # [E, n_edge_feat] float32
    pos: np.ndarray               # [N, 2] float32  (col, row)
    local_std: np.ndarray         # [N] float32
    grid_shape: Optional[Tuple[int, int]] = None
    valid_rows: Optional[np.ndarray] = None   # int64
    valid_cols: Optional[np.ndarray] = None   # int64
    num_valid_cells: int = 0

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])


def masked_local_stats(depth: np.ndarray, valid_mask: np.ndarray, size: int = 5):
    """graph_construction.py:378-432 -- 5x5 zero-padded masked mean/std in float64."""
    depth_masked = np.where(valid_mask, depth, 0.0).astype(np.float64)
    valid_float = valid_mask.astype(np.float64)
    kernel_area = float(size * size)
    sum_vals = ndimage.uniform_filter(depth_masked, size=size, mode="constant", cval=0.0) * kernel_area
    count = ndimage.uniform_filter(valid_float, size=size, mode="constant", cval=0.0) * kernel_area
    safe_count = np.maximum(count, 1.0)
    local_mean = (sum_vals / safe_count).astype(np.float32)
    depth_sq_masked = np.where(valid_mask, depth.astype(np.float64) ** 2, 0.0)
    sum_sq = ndimage.uniform_filter(depth_sq_masked, size=size, mode="constant", cval=0.0) * kernel_area
    mean_sq = sum_sq / safe_count
    variance = mean_sq - (sum_vals / safe_count) ** 2
    variance = np.maximum(variance, 0.0)
    local_std = np.sqrt(variance).astype(np.float32)
    return local_mean, local_std, count.astype(np.float32)


def curvature(depth_filled: np.ndarray, valid_mask: Optional[np.ndarray]):
    """graph_construction.py:434-458 -- Laplacian (mode='reflect'), zeroed where the
    zero-padded 3x3 valid count is < 3."""
    curv = ndimage.laplace(depth_filled)
    if valid_mask is not None:
        kernel = np.ones((3, 3), dtype=np.float64)
        neighbor_count = ndimage.convolve(valid_mask.astype(np.float64), kernel, mode="constant", cval=0.0)
        curv[neighbor_count < 3] = 0.0
    return curv


def node_features(depth, valid_rows, valid_cols, uncertainty, valid_mask,
                  feature_names: Sequence[str]):
    """graph_construction.py:245-327."""
    if valid_mask is None:
        valid_mask = np.isfinite(depth) & (np.abs(depth) < 1.0e5)
    local_mean, local_std, _ = masked_local_stats(depth, valid_mask, size=5)
    depth_filled = np.where(valid_mask, depth, local_mean)
    depth_filled = np.nan_to_num(depth_filled, nan=0.0)
    grad_y, grad_x = np.gradient(depth_filled)
    grad_mag = np.sqrt(grad_x ** 2 + grad_y ** 2)
    curv = curvature(depth_filled, valid_mask)

    feats = []
    for name in feature_names:
        if name == "depth":
            f = depth[valid_rows, valid_cols]
        elif name == "local_mean":
            f = local_mean[valid_rows, valid_cols]
        elif name == "local_std":
            f = local_std[valid_rows, valid_cols]
        elif name == "gradient_x":
            f = grad_x[valid_rows, valid_cols]
        elif name == "gradient_y":
            f = grad_y[valid_rows, valid_cols]
        elif name == "gradient_magnitude":
            f = grad_mag[valid_rows, valid_cols]
        elif name == "curvature":
            f = curv[valid_rows, valid_cols]
        elif name == "uncertainty" and uncertainty is not None:
            f = uncertainty[valid_rows, valid_cols]
        else:
            continue
        feats.append(np.nan_to_num(f, nan=0.0))
    if uncertainty is not None and "uncertainty" not in feature_names:
        feats.append(np.nan_to_num(uncertainty[valid_rows, valid_cols], nan=0.0))
    x = np.stack(feats, axis=1).astype(np.float32)
    node_local_std = np.nan_to_num(local_std[valid_rows, valid_cols], nan=0.0).astype(np.float32)
    return x, node_local_std


def build_edges(valid_rows, valid_cols, node_index_grid, grid_shape, offsets,
                include_self_loops: bool):
    """graph_construction.py:176-243.  Returns edge_index [2,E] int64 and the
    (src_r, src_c, tgt_r, tgt_c) coordinate arrays (int64) in edge order."""
    height, width = grid_shape
    node_indices = np.arange(len(valid_rows))
    srcs, tgts, sr, sc, tr, tc = [], [], [], [], [], []
    for dr, dc in offsets:
        nr = valid_rows + dr
        nc = valid_cols + dc
        in_bounds = (nr >= 0) & (nr < height) & (nc >= 0) & (nc < width)
        nbr = node_index_grid[np.clip(nr, 0, height - 1), np.clip(nc, 0, width - 1)]
        ok = in_bounds & (nbr >= 0)
        srcs.append(node_indices[ok]); tgts.append(nbr[ok])
        sr.append(valid_rows[ok]); sc.append(valid_cols[ok]); tr.append(nr[ok]); tc.append(nc[ok])
    if include_self_loops:
        srcs.append(node_indices); tgts.append(node_indices)
        sr.append(valid_rows); sc.append(valid_cols); tr.append(valid_rows); tc.append(valid_cols)
    if srcs:
        src = np.concatenate(srcs); tgt = np.concatenate(tgts)
        coords = tuple(np.concatenate(a).astype(np.int64) for a in (sr, sc, tr, tc))
    else:
        src = tgt = np.array([], dtype=np.int64)
        coords = tuple(np.array([], dtype=np.int64) for _ in range(4))
    edge_index = np.stack([src, tgt]).astype(np.int64)
    return edge_index, coords


def edge_features(depth, coords, resolution, feature_names: Sequence[str]):
    """graph_construction.py:329-376.

    dtype trail of the reference's scalar loop: the coordinate
    deltas are ``np.int64``; ``int64 * float`` -> float64, so distance is float64;
    ``depth[t] - depth[s]`` is a float32 subtract for float32 depth;
    ``float32 / float64`` -> float64, so ``arctan``/``degrees`` run in float64;
    the per-feature list goes through ``nan_to_num`` (float64) and the stacked
    matrix is cast to float32 at ``:374``.
    """
    src_r, src_c, tgt_r, tgt_c = coords
    n = len(src_r)
    if n == 0:
        return np.zeros((0, len(feature_names)), dtype=np.float32)
    res_x, res_y = resolution
    dx = (tgt_c - src_c) * res_x
    dy = (tgt_r - src_r) * res_y
    dist = np.sqrt(dx ** 2 + dy ** 2)                      # float64
    dz = depth[tgt_r, tgt_c] - depth[src_r, src_c]         # depth dtype (float32)
    cols = []
    for name in feature_names:
        if name == "distance":
            v = dist
        elif name == "depth_difference":
            v = dz
        elif name == "slope":
            with np.errstate(divide="ignore", invalid="ignore"):
                v = np.degrees(np.arctan(dz / np.where(dist > 0, dist, 1.0)))
            v = np.where(dist > 0, v, 0.0)
        else:
            v = np.zeros(n, dtype=np.float64)
        cols.append(np.nan_to_num(np.asarray(v), nan=0.0))   # keeps float32 for depth_difference
    return np.stack(cols, axis=1).astype(np.float32)


def build_graph(depth: np.ndarray,
                valid_mask: Optional[np.ndarray] = None,
                uncertainty: Optional[np.ndarray] = None,
                resolution: Tuple[float, float] = (1.0, 1.0),
                connectivity: str = "8-connected",
                include_self_loops: bool = False,
                node_feature_names: Optional[List[str]] = None,
                edge_feature_names: Optional[List[str]] = None) -> GraphCPU:
    """graph_construction.py:91-174."""
    if connectivity not in OFFSETS:
        raise ValueError(f"Unknown connectivity: {connectivity}")   # :89
    nfn = node_feature_names or DEFAULT_NODE_FEATURES
    efn = edge_feature_names or DEFAULT_EDGE_FEATURES
    if valid_mask is None:
        valid_mask = np.isfinite(depth)
    valid_rows, valid_cols = np.where(valid_mask)
    n = len(valid_rows)
    if n == 0:                                                      # :119-121, :460-469
        return GraphCPU(
            x=np.zeros((0, len(nfn)), np.float32), edge_index=np.zeros((2, 0), np.int64),
            edge_attr=np.zeros((0, len(efn)), np.float32), pos=np.zeros((0, 2), np.float32),
            local_std=np.zeros(0, np.float32))
    node_index_grid = np.full(depth.shape, -1, dtype=np.int64)
    node_index_grid[valid_rows, valid_cols] = np.arange(n)
    edge_index, coords = build_edges(valid_rows, valid_cols, node_index_grid, depth.shape,
                                     OFFSETS[connectivity], include_self_loops)
    x, node_local_std = node_features(depth, valid_rows, valid_cols, uncertainty, valid_mask, nfn)
    ea = edge_features(depth, coords, resolution, efn)
    pos = np.stack([valid_cols, valid_rows], axis=1).astype(np.float32)
    return GraphCPU(x=x, edge_index=edge_index, edge_attr=ea, pos=pos, local_std=node_local_std,
                    grid_shape=tuple(depth.shape), valid_rows=valid_rows.astype(np.int64),
                    valid_cols=valid_cols.astype(np.int64), num_valid_cells=n)


def graph_to_grid(g: GraphCPU, node_values: np.ndarray, fill_value: float = np.nan) -> np.ndarray:
    """graph_construction.py:471-505."""
    if g.grid_shape is None:
        raise ValueError("Data object missing grid_shape metadata")
    node_values = np.asarray(node_values)
    if node_values.ndim != 1:
        raise ValueError("For multi-channel node values, call graph_to_grid for each channel")
    grid = np.full(g.grid_shape, fill_value, dtype=np.float32)
    grid[g.valid_rows, g.valid_cols] = node_values
    return grid


def batch_graphs(graphs: Sequence[GraphCPU]):
    """PyG ``Batch.from_data_list`` semantics: concatenate node/edge tensors,
    offset edge_index by the cumulative node count, batch[N] = graph id."""
    xs, eis, eas, lss, bs = [], [], [], [], []
    off = 0
    for gi, g in enumerate(graphs):
        xs.append(g.x); eas.append(g.edge_attr); lss.append(g.local_std)
        eis.append(g.edge_index + off)
        bs.append(np.full(g.num_nodes, gi, dtype=np.int64))
        off += g.num_nodes
    return (np.concatenate(xs, 0), np.concatenate(eis, 1), np.concatenate(eas, 0),
            np.concatenate(lss, 0), np.concatenate(bs, 0))
