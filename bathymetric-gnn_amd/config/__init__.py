"""The slice of the reference's ``config`` package the hot path reads
(reference ``config/config.py``, ``config/constants.py``)."""
from .config import Config, TileConfig, GraphConfig, ModelConfig, InferenceConfig
from .constants import CORRECTION_NORM_FLOOR, CORRECTION_NORM_CAP

__all__ = ["Config", "TileConfig", "GraphConfig", "ModelConfig", "InferenceConfig",
           "CORRECTION_NORM_FLOOR", "CORRECTION_NORM_CAP"]
