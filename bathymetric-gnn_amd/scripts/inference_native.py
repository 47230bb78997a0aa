"""NativeVRProcessor -- drop-in for the class in the reference's ``scripts/inference_native.py``
(``:117-342``): per-refinement-grid inference with node-budget batching.

Where the reference builds one torch_geometric graph per grid on the CPU and concatenates them
with ``Batch.from_data_list`` at flush time, this version only queues the raw grids; ``flush_batch``
hands the whole batch to the fused GPU call (``bgnn_infer_tiles``), which builds the block-diagonal
graph, classifies it and scatters the results back to per-grid arrays.
"""
from __future__ import annotations

import logging
from typing import List, Optional, Tuple

import numpy as np
import torch

from ..config.constants import CORRECTION_NORM_FLOOR
from ..data import GraphBuilder
from ..models.gnn import BathymetricGNN
from ..models.pipeline import TileBatchEngine

logger = logging.getLogger(__name__)

Result = Tuple[np.ndarray, np.ndarray, np.ndarray]


class NativeVRProcessor:
    CLASS_NOISE = 2
    BATCH_NODE_BUDGET = 50000        # nodes to accumulate before a flush (reference :128)

    def __init__(self, model: BathymetricGNN, graph_builder: GraphBuilder, device=None,
                 auto_correct_threshold: float = 0.85):
        self.model = model
        self.graph_builder = graph_builder
        self.device = device
        self.auto_correct_threshold = auto_correct_threshold
        self.model.eval()
        try:
            self.expected_in_channels = model.feature_extractor.mlp[0].in_features
            logger.info(f"Model expects {self.expected_in_channels} input features")
        except (AttributeError, IndexError):
            logger.warning("Could not detect model input channels; will use all available features")
            self.expected_in_channels = None
        self._engine = TileBatchEngine(model, graph_builder, device if (device is not None and torch.device(device).type == "cuda") else None,
                                       auto_correct_threshold, 0.6, CORRECTION_NORM_FLOOR)
        self._batch = []             # (depth, valid_mask, uncertainty|None, resolution)
        self._batch_node_count = 0

    # ---- helpers ---------------------------------------------------------------------------
    def _prepare(self, depth, uncertainty, resolution, nodata):
        valid_mask = (depth != nodata) & np.isfinite(depth)          # :160
        if not np.any(valid_mask):
            return None
        use_unc = None if self.expected_in_channels == 7 else uncertainty   # :165-167
        return (depth, valid_mask, use_unc, resolution)

    @staticmethod
    def _empty(depth) -> Result:
        z = np.zeros(np.shape(depth), dtype=np.float32)
        return (z, z.copy(), z.copy())

    def _run(self, items) -> List[Result]:
        has_unc = any(it[2] is not None for it in items)
        res = self._engine.infer([it[0] for it in items], [it[1] for it in items],
                                 [it[2] for it in items] if has_unc else None, [it[3] for it in items])
        return [(r["classification"], r["confidence"], r["correction"]) for r in res]

    # ---- reference API ---------------------------------------------------------------------
    def process_grid(self, depth: np.ndarray, uncertainty: Optional[np.ndarray], resolution: tuple,
                     nodata: float = 1.0e6) -> Result:
        """One refinement grid, unbatched (:206-247): (classification, confidence, correction)."""
        item = self._prepare(depth, uncertainty, resolution, nodata)
        if item is None:
            return self._empty(depth)
        return self._run([item])[0]

    def add_to_batch(self, depth, uncertainty, resolution, nodata=1.0e6):
        """Queue a grid (:249-269).  Returns None when queued, or the all-zero result tuple
        immediately for a grid with no valid cell."""
        item = self._prepare(depth, uncertainty, resolution, nodata)
        if item is None:
            return self._empty(depth)
        self._batch.append(item)
        self._batch_node_count += int(np.count_nonzero(item[1]))
        return None

    @property
    def batch_ready(self) -> bool:
        return self._batch_node_count >= self.BATCH_NODE_BUDGET

    @property
    def batch_pending(self) -> bool:
        return len(self._batch) > 0

    def flush_batch(self) -> List[Result]:
        """Classify every queued grid in one fused pass (:281-342); results in insertion order."""
        if not self._batch:
            return []
        items, self._batch, self._batch_node_count = self._batch, [], 0
        return self._run(items)


def apply_results(depth: np.ndarray, uncertainty: Optional[np.ndarray], classification: np.ndarray,
                  confidence: np.ndarray, correction: np.ndarray, valid_mask: np.ndarray,
                  auto_correct_threshold: float = 0.85):
    """Write-back arithmetic of the reference's ``main`` (``apply_results``, :480-503), in place:
    cells classified noise, valid and with confidence >= threshold get ``depth -= correction`` and
    ``uncertainty *= (2 - confidence)``.  Returns the boolean mask that was applied."""
    apply = (classification == NativeVRProcessor.CLASS_NOISE) & valid_mask & (confidence >= auto_correct_threshold)
    depth[apply] -= correction[apply]
    if uncertainty is not None:
        uncertainty[apply] *= (2.0 - confidence[apply])
    return apply
