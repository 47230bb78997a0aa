"""Graph construction -- drop-in for the reference's ``data/graph_construction.py``.

``GraphBuilder`` keeps the reference's constructor and method signatures
(``data/graph_construction.py:34-89, 91-174, 471-505``).  ``build_graph`` uploads the tile and
builds the graph on the GPU (``bgnn_graph_build``); the returned ``GraphData`` exposes the same
attributes the reference attaches to a torch_geometric ``Data`` (``x, edge_index, edge_attr, pos,
grid_shape, valid_rows, valid_cols, num_valid_cells, local_std``), materialised lazily on the
device -- the model never needs the int64 ``edge_index``, only parity checks and foreign code do.
"""
from __future__ import annotations

import ctypes as C
import logging
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import runtime as rt

logger = logging.getLogger(__name__)

DEFAULT_NODE_FEATURES = ["depth", "local_mean", "local_std", "gradient_x", "gradient_y",
                         "gradient_magnitude", "curvature"]
DEFAULT_EDGE_FEATURES = ["distance", "depth_difference", "slope"]


class Data:
    """Minimal attribute bag standing in for ``torch_geometric.data.Data``: used for
    empty graphs, CPU copies and graphs assembled by foreign code."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])

    def _map(self, fn):
        out = Data()
        for k, v in self.__dict__.items():
            setattr(out, k, fn(v) if isinstance(v, torch.Tensor) else v)
        return out

    def to(self, device):
        return self._map(lambda t: t.to(device))

    def cpu(self):
        return self._map(lambda t: t.cpu())


class GraphData:
    """Device-resident graph (one tile, or a batch of tiles in ``Batch.from_data_list`` order)."""

    def __init__(self, ctx: rt.Context, handle, hw: np.ndarray, n_feat: int, edge_dim: int):
        self._ctx = ctx
        self._h = handle
        self._hw = hw                      # int32 [T,2]
        self._cache = {}
        self._sizes = None
        self.num_features = n_feat
        self.edge_dim = edge_dim
        if hw.shape[0] == 1:
            self.grid_shape = (int(hw[0, 0]), int(hw[0, 1]))   # graph_construction.py:158

    @property
    def _handle(self):
        """The library's graph handle.  It dies with the context it was built on (``bgnn_ctx_destroy`` releases the graphs
        still alive on it), so a graph of a closed context refuses to run instead of handing out a dangling pointer."""
        if self._ctx.handle is None:
            raise rt.BgnnError("this graph's library context has been closed; build the graph again on a live context")
        return self._h

    # ---- sizes ---------------------------------------------------------------------------
    def _counts(self):
        if self._sizes is None:
            T = self._hw.shape[0]
            nn, ne = C.c_int64(), C.c_int64()
            noff = (C.c_int64 * (T + 1))()
            eoff = (C.c_int64 * (T + 1))()
            self._ctx.begin()
            rt.check(self._ctx.lib.bgnn_graph_counts(self._handle, C.byref(nn), C.byref(ne), None, None, noff, eoff))
            self._ctx.end()
            self._sizes = (nn.value, ne.value, np.array(noff[:], dtype=np.int64), np.array(eoff[:], dtype=np.int64))
        return self._sizes

    @property
    def num_nodes(self) -> int:
        return self._counts()[0]

    @property
    def num_edges(self) -> int:
        return self._counts()[1]

    @property
    def num_valid_cells(self) -> int:        # graph_construction.py:161
        return self.num_nodes

    @property
    def num_graphs(self) -> int:
        return int(self._hw.shape[0])

    @property
    def ptr(self) -> torch.Tensor:           # PyG Batch.ptr
        return torch.as_tensor(self._counts()[2])

    @property
    def device(self):
        return self._ctx.device

    # ---- lazily exported tensors ------------------------------------------------------------
    def _export(self, names: Sequence[str]):
        need = [n for n in names if n not in self._cache]
        if not need:
            return
        N, E = self.num_nodes, self.num_edges
        dev = self._ctx.device
        F, ED = self.num_features, self.edge_dim
        shapes = {
            "x": ((N, F), torch.float32), "edge_index": ((2, E), torch.int64), "edge_attr": ((E, ED), torch.float32),
            "pos": ((N, 2), torch.float32), "valid_rows": ((N,), torch.int64), "valid_cols": ((N,), torch.int64),
            "local_std": ((N,), torch.float32), "batch": ((N,), torch.int64),
        }
        bufs = {n: torch.empty(shapes[n][0], dtype=shapes[n][1], device=dev) for n in need}
        if "edge_index" in need and "edge_attr" not in bufs and "edge_attr" not in self._cache:
            bufs["edge_attr"] = torch.empty(shapes["edge_attr"][0], dtype=torch.float32, device=dev)
        order = ["x", "edge_index", "edge_attr", "pos", "valid_rows", "valid_cols", "local_std", "batch"]
        args = [rt.ptr(bufs.get(n)) for n in order]
        self._ctx.begin()
        rt.check(self._ctx.lib.bgnn_graph_export(self._handle, *args))
        self._ctx.end()
        self._cache.update(bufs)

    x = property(lambda self: (self._export(["x"]), self._cache["x"])[1])
    edge_index = property(lambda self: (self._export(["edge_index"]), self._cache["edge_index"])[1])
    edge_attr = property(lambda self: (self._export(["edge_attr"]), self._cache["edge_attr"])[1])
    pos = property(lambda self: (self._export(["pos"]), self._cache["pos"])[1])
    valid_rows = property(lambda self: (self._export(["valid_rows"]), self._cache["valid_rows"])[1])
    valid_cols = property(lambda self: (self._export(["valid_cols"]), self._cache["valid_cols"])[1])
    local_std = property(lambda self: (self._export(["local_std"]), self._cache["local_std"])[1])
    batch = property(lambda self: (self._export(["batch"]), self._cache["batch"])[1])

    # ---- Data-like movement ---------------------------------------------------------------
    def to(self, device):
        """The graph already lives on the GPU it was built on; ``.to`` of that device (what the
        reference does at models/pipeline.py:267) is the identity."""
        d = torch.device(device)
        if d.type == "cpu":
            return self.cpu()
        if d.index is not None and d.index != self._ctx.device.index:
            raise rt.BgnnError("a GraphData cannot move between GPUs; build it on the target device")
        return self

    def cuda(self):
        return self

    def cpu(self) -> Data:
        """CPU copy with the reference's attribute set (used with graph_to_grid)."""
        self._export(["x", "edge_index", "edge_attr", "pos", "valid_rows", "valid_cols", "local_std"])
        d = Data(**{k: self._cache[k].cpu() for k in
                    ("x", "edge_index", "edge_attr", "pos", "valid_rows", "valid_cols", "local_std")})
        if hasattr(self, "grid_shape"):
            d.grid_shape = self.grid_shape
        d.num_valid_cells = self.num_nodes
        d._device_graph = self
        return d

    def __del__(self):
        try:
            # a closed context has already released its graphs (bgnn_ctx_destroy): nothing to free, and nothing to touch
            if self._h and self._ctx.handle is not None:
                self._ctx.lib.bgnn_graph_destroy(self._h)
            self._h = None
        except Exception:
            pass


def _as_f32(a, name):
    a = np.asarray(a)
    if a.dtype != np.float32:
        # the reference's loaders hand float32 grids (BAG / GeoTIFF bands); other dtypes would take
        # a different numpy promotion path there.  float32 is what the kernels restate.
        a = a.astype(np.float32)
    if a.ndim != 2:
        raise ValueError(f"{name} must be a 2-D grid, got shape {a.shape}")
    return np.ascontiguousarray(a)


class GraphBuilder:
    """Builds graph structures from gridded bathymetric data on the GPU."""

    def __init__(self, connectivity: str = "8-connected", include_self_loops: bool = False,
                 node_features: Optional[List[str]] = None, edge_features: Optional[List[str]] = None,
                 device=None):
        rt.load_library()   # the reference raises ImportError when its backend is missing (:50-54); so do we
        self.connectivity = connectivity
        self.include_self_loops = include_self_loops
        self.node_features = node_features or list(DEFAULT_NODE_FEATURES)
        self.edge_features = edge_features or list(DEFAULT_EDGE_FEATURES)
        if connectivity == "4-connected":                    # :78-89
            self.neighbor_offsets = [(-1, 0), (1, 0), (0, -1), (0, 1)]
        elif connectivity == "8-connected":
            self.neighbor_offsets = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]
        elif connectivity == "16-dilated":                   # build-side extension (BASELINE config 3, no oracle)
            base = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]
            self.neighbor_offsets = base + [(2 * r, 2 * c) for r, c in base]
        else:
            raise ValueError(f"Unknown connectivity: {connectivity}")
        self._device = device
        self._opts = rt.make_graph_opts(connectivity, include_self_loops, self.node_features, self.edge_features)

    # ---- helpers ---------------------------------------------------------------------------
    def _ctx(self) -> rt.Context:
        return rt.get_context(self._device)

    def n_node_columns(self, has_uncertainty: bool) -> int:
        n = 0
        for name in self.node_features:
            if name not in rt.NODE_FEATURE_IDS:
                continue
            if name == "uncertainty" and not has_uncertainty:
                continue
            n += 1
        if has_uncertainty and "uncertainty" not in self.node_features:
            n += 1
        return n

    def upload_tiles(self, depths, masks, uncs, resolutions):
        """Host grids -> flat device tensors + tile table.  ``masks[i]`` None means
        ``np.isfinite(depth)`` (graph_construction.py:110-111).  Either every tile has an
        uncertainty grid or none has."""
        ctx = self._ctx()
        n = len(depths)
        hw = np.empty((n, 2), np.int32)
        res = np.empty((n, 2), np.float64)
        dl, ml, ul = [], [], []
        has_unc = uncs is not None and any(u is not None for u in uncs)
        for i in range(n):
            d = _as_f32(depths[i], "depth")
            hw[i] = d.shape
            res[i] = (float(resolutions[i][0]), float(resolutions[i][1]))
            m = masks[i] if masks is not None else None
            m = np.isfinite(d) if m is None else np.asarray(m, dtype=bool)
            if m.shape != d.shape:
                raise ValueError(f"valid_mask shape {m.shape} != depth shape {d.shape}")
            dl.append(d.ravel()); ml.append(m.ravel().view(np.uint8))
            if has_unc:
                if uncs[i] is None:
                    raise ValueError("either every grid of a batch has an uncertainty layer or none has")
                u = _as_f32(uncs[i], "uncertainty")
                if u.shape != d.shape:
                    raise ValueError(f"uncertainty shape {u.shape} != depth shape {d.shape}")
                ul.append(u.ravel())
        dev = ctx.device
        depth_t = torch.from_numpy(np.concatenate(dl) if n > 1 else dl[0]).to(dev, non_blocking=False)
        mask_t = torch.from_numpy(np.concatenate(ml) if n > 1 else ml[0]).to(dev)
        unc_t = torch.from_numpy(np.concatenate(ul) if n > 1 else ul[0]).to(dev) if has_unc else None
        return hw, res, depth_t, mask_t, unc_t

    def build_from_device(self, hw: np.ndarray, res: np.ndarray, depth_t: torch.Tensor, mask_t: torch.Tensor,
                          unc_t: Optional[torch.Tensor], ctx: Optional[rt.Context] = None) -> GraphData:
        """Batch entry on device-resident tiles (flat, concatenated row-major).  ``ctx``: build on this library context
        (``rt.new_context``) instead of the device's default one; the graph lives and dies with it."""
        ctx = ctx if ctx is not None else self._ctx()
        cells = int((hw[:, 0].astype(np.int64) * hw[:, 1]).sum())
        if depth_t.numel() != cells or mask_t.numel() != cells or (unc_t is not None and unc_t.numel() != cells):
            raise ValueError("tile table and device buffers disagree on the number of cells")
        assert depth_t.dtype == torch.float32 and mask_t.dtype in (torch.uint8, torch.bool)
        tiles, keep = rt.make_tiles(hw, res, depth_t, mask_t, unc_t)
        h = C.c_void_p()
        ctx.begin()
        rt.check(ctx.lib.bgnn_graph_build(ctx.handle, C.byref(tiles), C.byref(self._opts), C.byref(h)))
        ctx.end()
        g = GraphData(ctx, h, keep[0], self.n_node_columns(unc_t is not None), len(self.edge_features))
        return g

    # ---- reference API ---------------------------------------------------------------------
    def build_graph(self, depth: np.ndarray, valid_mask: Optional[np.ndarray] = None,
                    uncertainty: Optional[np.ndarray] = None,
                    resolution: Tuple[float, float] = (1.0, 1.0)):
        """Build a graph from one grid (reference ``build_graph``, :91-174)."""
        hw, res, d, m, u = self.upload_tiles([depth], [valid_mask], [uncertainty], [resolution])
        g = self.build_from_device(hw, res, d, m, u)
        if g.num_nodes == 0:
            logger.warning("No valid cells in grid")
            return self._create_empty_graph()
        return g

    def build_graphs(self, depths, valid_masks=None, uncertainties=None, resolutions=None) -> GraphData:
        """Batch of grids -> one block-diagonal graph (what ``Batch.from_data_list`` of the
        per-grid graphs would hold, scripts/inference_native.py:312)."""
        n = len(depths)
        if resolutions is None:
            resolutions = [(1.0, 1.0)] * n
        hw, res, d, m, u = self.upload_tiles(depths, valid_masks, uncertainties, resolutions)
        return self.build_from_device(hw, res, d, m, u)

    def _create_empty_graph(self) -> Data:           # :460-469 (note: no grid_shape attribute)
        return Data(x=torch.zeros((0, len(self.node_features)), dtype=torch.float32),
                    edge_index=torch.zeros((2, 0), dtype=torch.long),
                    edge_attr=torch.zeros((0, len(self.edge_features)), dtype=torch.float32),
                    pos=torch.zeros((0, 2), dtype=torch.float32),
                    local_std=torch.zeros(0, dtype=torch.float32))

    def graph_to_grid(self, data, node_values: torch.Tensor, fill_value: float = np.nan) -> np.ndarray:
        """Node values back to grid format (reference :471-505): [h, w] float32 numpy array."""
        if not hasattr(data, "grid_shape"):
            raise ValueError("Data object missing grid_shape metadata")
        if node_values.dim() != 1:
            raise ValueError("For multi-channel node values, call graph_to_grid for each channel")
        g = data if isinstance(data, GraphData) else getattr(data, "_device_graph", None)
        if g is None:
            raise rt.BgnnError("graph_to_grid needs a graph built by this GraphBuilder (device-resident)")
        ctx = g._ctx
        if node_values.shape[0] != g.num_nodes:
            raise ValueError(f"node_values has {node_values.shape[0]} entries, graph has {g.num_nodes} nodes")
        vals = node_values.detach().to(ctx.device, torch.float32).contiguous()
        h, w = data.grid_shape
        grid = torch.empty((h, w), dtype=torch.float32, device=ctx.device)
        ctx.begin()
        rt.check(ctx.lib.bgnn_graph_scatter(g._handle, rt.ptr(vals), C.c_float(fill_value), rt.ptr(grid)))
        ctx.end()
        return grid.cpu().numpy()
