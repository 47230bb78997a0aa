from .gnn import (BathymetricGNN, LocalFeatureExtractor, GNNBackbone, ClassificationHead, ConfidenceHead,
                  CorrectionHead)
from .pipeline import BathymetricPipeline, TileBatchEngine

__all__ = ["BathymetricGNN", "LocalFeatureExtractor", "GNNBackbone", "ClassificationHead", "ConfidenceHead",
           "CorrectionHead", "BathymetricPipeline", "TileBatchEngine"]
