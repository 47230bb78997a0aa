from .gnn import (BathymetricGNN, LocalFeatureExtractor, GNNBackbone, ClassificationHead, ConfidenceHead,
                  CorrectionHead)
from .pipeline import BathymetricPipeline, TileBatchEngine, HostTilePipeline, shard_info, exchange_tile_results

__all__ = ["BathymetricGNN", "LocalFeatureExtractor", "GNNBackbone", "ClassificationHead", "ConfidenceHead",
           "CorrectionHead", "BathymetricPipeline", "TileBatchEngine", "HostTilePipeline"]
