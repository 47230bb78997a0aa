"""Mirror of the reference's ``data`` package for the hot path: graph construction, tiling and the
``BathymetricGrid`` container (file-format I/O -- GDAL / h5py -- is outside the path)."""
from .graph_construction import GraphBuilder, GraphData, Data
from .grid import BathymetricGrid
from .tiling import Tile, TileSpec, TileManager, TileMerger

__all__ = ["GraphBuilder", "GraphData", "Data", "BathymetricGrid", "Tile", "TileSpec", "TileManager", "TileMerger"]
