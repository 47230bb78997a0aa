"""Overlapping tiles and seam-free merging -- counterpart of the reference's ``data/tiling.py``.

Same public surface (``Tile``, ``TileSpec``, ``TileManager``, ``TileMerger``) and the same
arithmetic for the tile grid (``:87-138``), the raised-cosine blend ramps (``:313-330``), weighted
accumulation (``:218-294``) and the confidence-arbitrated discrete channel (``:384-428``).  Specs
are kept as an ``[n, 6]`` integer table as well as objects so that batches of tiles can be handed to
the GPU in one call.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Dict, Generator, List, Optional, Tuple

import numpy as np

from .grid import BathymetricGrid

logger = logging.getLogger(__name__)


@dataclass
class TileSpec:
    row_start: int
    col_start: int
    row_end: int
    col_end: int
    tile_row: int
    tile_col: int


@dataclass
class Tile:
    data: np.ndarray
    uncertainty: Optional[np.ndarray]
    row_start: int
    col_start: int
    row_end: int
    col_end: int
    tile_row: int
    tile_col: int
    valid_mask: np.ndarray

    @property
    def shape(self) -> Tuple[int, int]:
        return self.data.shape

    @property
    def valid_ratio(self) -> float:
        return np.sum(self.valid_mask) / self.valid_mask.size


def _axis_spans(extent: int, tile: int, overlap: int) -> List[Tuple[int, int]]:
    """Start/end of every tile along one axis.  Count = ceil((extent - overlap) / stride), at
    least 1; a tile that would run off the end is clipped and then shifted back so it keeps the
    full tile size whenever the axis is long enough (reference :103-122)."""
    stride = tile - overlap
    usable = extent - overlap
    n = max(1, usable // stride + (1 if usable % stride > 0 else 0))
    spans = []
    for i in range(n):
        a = i * stride
        b = min(a + tile, extent)
        if b - a < tile and a > 0:
            a = max(0, b - tile)
        spans.append((a, b))
    return spans


class TileManager:
    def __init__(self, tile_size: int = 1024, overlap: int = 128, min_valid_ratio: float = 0.1):
        self.tile_size = tile_size
        self.overlap = overlap
        self.min_valid_ratio = min_valid_ratio
        self.stride = tile_size - overlap
        if self.stride <= 0:
            raise ValueError("Tile size must be larger than overlap")

    def compute_tile_grid(self, grid_shape: Tuple[int, int]) -> Tuple[int, int, List[TileSpec]]:
        height, width = grid_shape
        rows = _axis_spans(height, self.tile_size, self.overlap)
        cols = _axis_spans(width, self.tile_size, self.overlap)
        specs = [TileSpec(r0, c0, r1, c1, i, j) for i, (r0, r1) in enumerate(rows) for j, (c0, c1) in enumerate(cols)]
        logger.info(f"Grid {height}x{width} -> {len(rows)}x{len(cols)} = {len(specs)} tiles")
        return len(rows), len(cols), specs

    def extract_tile(self, grid: BathymetricGrid, spec: TileSpec, full_valid_mask: Optional[np.ndarray] = None) -> Tile:
        sl = (slice(spec.row_start, spec.row_end), slice(spec.col_start, spec.col_end))
        unc = grid.uncertainty[sl].copy() if grid.uncertainty is not None else None
        vm = grid.valid_mask if full_valid_mask is None else full_valid_mask
        return Tile(data=grid.depth[sl].copy(), uncertainty=unc, row_start=spec.row_start, col_start=spec.col_start,
                    row_end=spec.row_end, col_end=spec.col_end, tile_row=spec.tile_row, tile_col=spec.tile_col,
                    valid_mask=vm[sl].copy())

    def iterate_tiles(self, grid: BathymetricGrid, skip_empty: bool = True) -> Generator[Tile, None, None]:
        _, _, specs = self.compute_tile_grid(grid.shape)
        vm = grid.valid_mask          # computed once (the reference recomputes it per tile, same values)
        for spec in specs:
            tile = self.extract_tile(grid, spec, vm)
            if skip_empty and tile.valid_ratio < self.min_valid_ratio:
                logger.debug(f"Skipping tile ({tile.tile_row}, {tile.tile_col}) - valid ratio {tile.valid_ratio:.2%}")
                continue
            yield tile

    def create_output_grid(self, grid_shape, dtype=np.float32, fill_value: float = np.nan) -> np.ndarray:
        return np.full(grid_shape, fill_value, dtype=dtype)

    # -- blending ---------------------------------------------------------------------------
    def _create_1d_blend(self, size: int) -> np.ndarray:
        """1 in the middle, raised-cosine ramps of length min(overlap, size//4) at both ends; the
        first and last weights are exactly 0 (reference :313-330)."""
        wts = np.ones(size, dtype=np.float32)
        ramp = min(self.overlap, size // 4)
        if ramp > 0:
            up = 0.5 * (1 - np.cos(np.pi * np.linspace(0, 1, ramp)))
            wts[:ramp] = up
            wts[-ramp:] = 0.5 * (1 - np.cos(np.pi * np.linspace(1, 0, ramp)))
        return wts

    def _create_blend_weights(self, shape: Tuple[int, int]) -> np.ndarray:
        return np.outer(self._create_1d_blend(shape[0]), self._create_1d_blend(shape[1])).astype(np.float32)

    def merge_tile(self, output: np.ndarray, tile_data: np.ndarray, spec: TileSpec,
                   weight_grid: Optional[np.ndarray] = None):
        wts = self._create_blend_weights((spec.row_end - spec.row_start, spec.col_end - spec.col_start))
        sl = (slice(spec.row_start, spec.row_end), slice(spec.col_start, spec.col_end))
        region = output[sl]
        ok = np.isfinite(tile_data)
        if weight_grid is not None:        # weighted-average mode (:242-258)
            wregion = weight_grid[sl]
            region[np.isnan(region) & ok] = 0.0
            wregion[ok] += wts[ok]
            region[ok] += (tile_data * wts)[ok]
        else:                              # overwrite-with-blend mode (:259-272)
            have = np.isfinite(region)
            both = ok & have
            fresh = ok & ~have
            region[both] = region[both] * (1 - wts[both]) + tile_data[both] * wts[both]
            region[fresh] = tile_data[fresh]

    def finalize_output(self, output: np.ndarray, weight_grid: Optional[np.ndarray] = None) -> np.ndarray:
        if weight_grid is not None:
            pos = weight_grid > 0
            output[pos] /= weight_grid[pos]
        return output


class TileMerger:
    DISCRETE_CHANNELS = {"classification"}

    def __init__(self, tile_manager: TileManager):
        self.tile_manager = tile_manager
        self.outputs: Dict[str, np.ndarray] = {}
        self.weights: Dict[str, np.ndarray] = {}
        self._confidence_tracker: Optional[np.ndarray] = None

    def initialize(self, grid_shape: Tuple[int, int], channels: List[str], dtypes: Optional[dict] = None):
        dtypes = dtypes or {}
        for ch in channels:
            self.outputs[ch] = np.full(grid_shape, np.nan, dtype=dtypes.get(ch, np.float32))
            self.weights[ch] = np.zeros(grid_shape, dtype=np.float32)
        logger.info(f"Initialized {len(channels)} output channels for {grid_shape}")
        if any(ch in self.DISCRETE_CHANNELS for ch in channels):
            self._confidence_tracker = np.full(grid_shape, -1.0, dtype=np.float32)

    def add_tile(self, spec: TileSpec, channel_data: dict):
        conf = channel_data.get("confidence", None)
        sl = (slice(spec.row_start, spec.row_end), slice(spec.col_start, spec.col_end))
        for ch, data in channel_data.items():
            if ch not in self.outputs:
                raise ValueError(f"Unknown channel: {ch}")
            if ch in self.DISCRETE_CHANNELS and conf is not None and self._confidence_tracker is not None:
                # labels are not averaged: the tile with the higher confidence wins (:408-420)
                region = self.outputs[ch][sl]
                tracked = self._confidence_tracker[sl]
                take = np.isfinite(data) & ((conf > tracked) | np.isnan(region))
                region[take] = data[take]
                tracked[take] = conf[take]
            else:
                self.tile_manager.merge_tile(self.outputs[ch], data, spec, self.weights[ch])

    def finalize(self) -> dict:
        results = {}
        for ch, arr in self.outputs.items():
            if ch in self.DISCRETE_CHANNELS:
                results[ch] = arr
            else:
                results[ch] = self.tile_manager.finalize_output(arr, self.weights[ch])
        self.outputs, self.weights, self._confidence_tracker = {}, {}, None
        return results
