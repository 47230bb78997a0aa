"""``BathymetricGrid`` container (reference ``data/loaders.py:42-90``).  Only the in-memory
dataclass is on the hot path; reading/writing BAG / GeoTIFF needs GDAL and stays with the
reference's loaders."""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Any, Optional, Tuple

import numpy as np


@dataclass
class BathymetricGrid:
    depth: np.ndarray
    uncertainty: Optional[np.ndarray] = None
    nodata_value: float = 1.0e6
    transform: Any = None
    crs: Any = None
    resolution: Tuple[float, float] = (1.0, 1.0)
    bounds: Any = None
    source_path: Optional[Path] = None

    @property
    def shape(self) -> Tuple[int, int]:
        return self.depth.shape

    @property
    def valid_mask(self) -> np.ndarray:
        """Finite and not the nodata sentinel (reference ``data/loaders.py:59``)."""
        d = self.depth
        m = np.isfinite(d)
        if self.nodata_value is not None and not np.isnan(self.nodata_value):
            m &= d != self.nodata_value
        return m

    @property
    def valid_ratio(self) -> float:
        return float(np.sum(self.valid_mask)) / self.depth.size
