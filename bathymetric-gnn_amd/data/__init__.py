"""Mirror of the reference's ``data`` package for the hot path: graph construction, tiling and the
``BathymetricGrid`` container (file-format I/O -- GDAL / h5py -- is outside the path; VR BAGs enter as their two HDF5 arrays)."""
from .graph_construction import GraphBuilder, GraphData, Data
from .grid import BathymetricGrid
from .tiling import Tile, TileSpec, TileManager, TileMerger
from .vr_bag import RefinementGrid, VRBagHandler, VRBagWriter, SRBagHandler, SRBagWriter, detect_bag_type

__all__ = ["GraphBuilder", "GraphData", "Data", "BathymetricGrid", "Tile", "TileSpec", "TileManager", "TileMerger",
           "RefinementGrid", "VRBagHandler", "VRBagWriter", "SRBagHandler", "SRBagWriter", "detect_bag_type"]
