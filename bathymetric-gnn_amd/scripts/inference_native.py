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
        # Two batches in flight (submit_batch / collect_batch): a 50 000-node batch is ONE round of workgroups per kernel, so its
        # twelve launches are a chain of latencies that leaves most of the GPU idle; the next batch runs on a second library
        # context (own HIP stream, own arenas) beside it.  flush_batch() stays the reference's synchronous call.
        self._engines = [self._engine]
        self._inflight = []          # tickets in submission order: {"engine", "hw", "out"}
        self._next_engine = 0

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
        """Classify every queued grid in one fused pass (:281-342); results in insertion order.  Synchronous, like the
        reference's (batches still in flight from ``submit_batch`` are not disturbed: this one queues behind the first
        context's work and their results stay available to ``collect_batch``)."""
        if not self._batch:
            return []
        items, self._batch, self._batch_node_count = self._batch, [], 0
        return self._run(items)

    # ---- two batches in flight (MI355X-first extension of the batching API; run_refinements uses it) --------------------------
    MAX_IN_FLIGHT = 2

    def _engine_for_next(self) -> TileBatchEngine:
        from .. import runtime as rt
        i = self._next_engine % self.MAX_IN_FLIGHT
        self._next_engine += 1
        while len(self._engines) <= i:              # the second library context is only created when batches are pipelined
            ctx = rt.new_context(self._engine.ctx.device)
            for k in ("matrix_path", "fused", "fold_extractor", "ragged_atlas", "fused_front", "features_tiled"):
                ctx.set_option(k, self._engine.ctx.get_option(k))
            self._engines.append(TileBatchEngine(self.model, self.graph_builder, self._engine.ctx.device, self.auto_correct_threshold,
                                                 self._engine.review_threshold, self._engine.norm_floor, ctx=ctx))
        return self._engines[i]

    def submit_batch(self) -> Optional[int]:
        """Start classifying the queued grids WITHOUT waiting for the result: the batch is uploaded and its kernels are queued on
        one of two library contexts, alternately, so that it runs beside the batch submitted before it.  Returns the number of
        batches now in flight (None if nothing was queued).  Results come back, in submission order, from ``collect_batch``.
        At most ``MAX_IN_FLIGHT`` batches may be outstanding: collect the oldest first."""
        if not self._batch:
            return None
        if len(self._inflight) >= self.MAX_IN_FLIGHT:
            raise RuntimeError(f"{self.MAX_IN_FLIGHT} batches are already in flight: collect_batch() the oldest first")
        items, self._batch, self._batch_node_count = self._batch, [], 0
        eng = self._engine_for_next()
        has_unc = any(it[2] is not None for it in items)
        use_unc = [it[2] for it in items] if (has_unc and self.model.in_channels == self.graph_builder.n_node_columns(True)) else None
        hw, res, d, m, u = self.graph_builder.upload_tiles([it[0] for it in items], [it[1] for it in items], use_unc, [it[3] for it in items])
        out = eng.infer_device(hw, res, d, m, u, defer_end=True)     # asynchronous: nothing waits for this batch yet
        self._inflight.append({"engine": eng, "hw": hw, "out": out, "keep": (d, m, u)})
        return len(self._inflight)

    @property
    def batches_in_flight(self) -> int:
        return len(self._inflight)

    def _collect(self, index: int) -> List[Result]:
        t = self._inflight.pop(index)
        t["engine"].ctx.end()                        # the caller's stream now waits for that batch ...
        out = t["out"].cpu().numpy()                 # ... and so does this copy
        results, off = [], 0
        for i in range(t["hw"].shape[0]):
            h, w = int(t["hw"][i, 0]), int(t["hw"][i, 1])
            n = h * w
            results.append((out[0, off:off + n].reshape(h, w).copy(), out[1, off:off + n].reshape(h, w).copy(),
                            out[2, off:off + n].reshape(h, w).copy()))
            off += n
        return results

    def collect_batch(self) -> List[Result]:
        """Results of the OLDEST batch in flight (blocks until it is done); same per-grid tuples as ``flush_batch``."""
        if not self._inflight:
            return []
        return self._collect(0)

    # ---- whole-BAG device path (MI355X-first replacement of the main loop, :445-538) ---------------------
    def process_refinements(self, handler, writer=None, min_valid_ratio: float = 0.0,
                            cell_budget: int = 8 << 20, return_results: bool = False):
        """Classify and correct every refinement grid of a VR BAG with the records resident in HBM.

        ``varres_refinements`` is already the concatenated-grid layout ``bgnn_infer_tiles`` consumes
        (grids row-major, one after another in ``varres_metadata.index`` order), so the records are uploaded
        as they are, in chunks of about ``cell_budget`` cells cut at grid boundaries (the reference batches
        50 000 nodes because PyG materialises per-edge tensors; here the bound is HBM).  Per chunk:
        ``bgnn_vr_unpack`` (planes, valid mask, ``min_valid_ratio`` filter) -> ``bgnn_infer_tiles`` ->
        ``bgnn_vr_apply`` (the write-back arithmetic of ``apply_results``) -> one D2H of the corrected
        records into ``writer``.  Returns the statistics the reference's ``main`` logs (:540-559); with
        ``return_results`` also per-record classification / confidence / correction arrays (what the
        sidecar builder consumes).  Results equal ``run_refinements`` (the grid-by-grid loop) bit for bit."""
        import ctypes as C
        from .. import runtime as rt
        eng = self._engine
        ctx, dev = eng.ctx, eng.ctx.device
        tab = handler.refinement_table()
        n_grids = len(tab["cells"])
        stats = {"grids_processed": 0, "cells_processed": 0, "cells_classified_noise": 0, "cells_corrected": 0,
                 "total_confidence": 0.0, "mean_confidence": 0.0, "grids_skipped": 0}
        total = int(tab["cells"].sum()) if n_grids else 0
        res_all = np.zeros((3, total), np.float32) if return_results else None
        if n_grids == 0:
            return (stats, res_all) if return_results else stats
        ref = handler.varres_refinements[0, :]
        plain = (ref.dtype.itemsize == 8 and ref.dtype.fields["depth"][1] == 0 and ref.dtype.fields["depth_uncrt"][1] == 4
                 and ref.dtype.fields["depth"][0] == np.dtype("<f4"))
        start0 = int(tab["index"][0])
        if tab["contiguous"] and plain:
            rec_all = np.ascontiguousarray(ref[start0:start0 + total]).view(np.float32).reshape(total, 2)
            perm = None
        else:       # records not laid out in iteration order (or foreign record layout): pack on the host once
            perm = np.concatenate([np.arange(i, i + c, dtype=np.int64) for i, c in zip(tab["index"], tab["cells"])])
            rec_all = np.empty((total, 2), np.float32)
            rec_all[:, 0] = ref["depth"][perm]; rec_all[:, 1] = ref["depth_uncrt"][perm]
        use_unc = self.expected_in_channels != 7
        off = np.zeros(n_grids + 1, np.int64); np.cumsum(tab["cells"], out=off[1:])
        hw = np.stack([tab["dims_y"], tab["dims_x"]], 1).astype(np.int32)
        res = np.stack([tab["res_x"], tab["res_y"]], 1).astype(np.float64)
        counts_t = torch.zeros(3, dtype=torch.int64, device=dev)
        csum_t = torch.zeros(1, dtype=torch.float64, device=dev)
        g0 = 0
        while g0 < n_grids:
            g1 = int(np.searchsorted(off, off[g0] + cell_budget, side="right")) - 1
            g1 = min(max(g1, g0 + 1), n_grids)
            lo, hi = int(off[g0]), int(off[g1])
            n = hi - lo
            rec_t = torch.from_numpy(rec_all[lo:hi]).to(dev)
            off_t = torch.from_numpy(off[g0:g1 + 1] - lo).to(dev)
            depth_t = torch.empty(n, dtype=torch.float32, device=dev)
            unc_t = torch.empty(n, dtype=torch.float32, device=dev) if use_unc else None
            mask_t = torch.empty(n, dtype=torch.uint8, device=dev)
            cnt_t = torch.empty(g1 - g0, dtype=torch.int64, device=dev)
            keep_t = torch.empty(g1 - g0, dtype=torch.uint8, device=dev)
            ctx.begin()
            rt.check(ctx.lib.bgnn_vr_unpack(ctx.handle, rt.ptr(rec_t), n, C.c_float(handler.NODATA), g1 - g0, rt.ptr(off_t),
                                            C.c_double(min_valid_ratio), rt.ptr(depth_t), rt.ptr(unc_t), rt.ptr(mask_t),
                                            rt.ptr(cnt_t), rt.ptr(keep_t)))
            ctx.end()
            out = eng.infer_device(hw[g0:g1], res[g0:g1], depth_t, mask_t, unc_t)
            before = counts_t[2].item() if writer is not None else 0
            ctx.begin()
            rt.check(ctx.lib.bgnn_vr_apply(ctx.handle, rt.ptr(rec_t), n, rt.ptr(mask_t), rt.ptr(out[0]), rt.ptr(out[1]),
                                           rt.ptr(out[2]), C.c_float(self.auto_correct_threshold), rt.ptr(counts_t),
                                           rt.ptr(csum_t)))
            ctx.end()
            keep = keep_t.cpu().numpy().astype(bool); cnt = cnt_t.cpu().numpy()
            stats["grids_processed"] += int(keep.sum()); stats["grids_skipped"] += int((~keep).sum())
            stats["cells_processed"] += int(cnt[keep].sum())
            if return_results:
                res_all[:, lo:hi] = out.cpu().numpy()
            if writer is not None:
                rec_np = rec_t.cpu().numpy()
                changed = counts_t[2].item() - before
                if perm is None:
                    writer.write_records(start0 + lo, rec_np, corrections_applied=changed)
                else:
                    for g in range(g0, g1):
                        writer.write_records(int(tab["index"][g]), rec_np[off[g] - lo:off[g + 1] - lo],
                                             corrections_applied=changed if g == g0 else 0)
            g0 = g1
        c = counts_t.cpu().numpy()
        stats["cells_classified_noise"] = int(c[0]); stats["cells_corrected"] = int(c[1])
        stats["total_confidence"] = float(csum_t.item())
        stats["mean_confidence"] = stats["total_confidence"] / stats["cells_processed"] if stats["cells_processed"] else 0
        return (stats, res_all) if return_results else stats


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


def run_refinements(processor: NativeVRProcessor, handler, writer, min_valid_ratio: float = 0.0,
                    auto_correct_threshold: Optional[float] = None, results_sink=None, pipelined: bool = True):
    """The grid-by-grid loop of the reference's ``main`` (:445-538): iterate the refinement grids, queue them
    with ``add_to_batch``, flush when ``batch_ready``, apply each grid's results (``apply_results`` closure,
    :480-503) and write it back with ``update_refinement_batch``.  Kept as the API-level mirror and as the
    statement ``NativeVRProcessor.process_refinements`` (records resident in HBM) is tested against.
    ``results_sink(grid, classification, confidence, correction)`` stands where the sidecar builder is fed.

    ``pipelined`` (default): a full batch is SUBMITTED (``submit_batch``) and the loop goes on queueing the next one; the
    results of the batch before it are collected -- and applied, in the same grid order -- while the new one runs on the
    processor's second library context.  Same results and statistics as the synchronous loop (``pipelined=False``: one
    ``flush_batch`` per full batch, exactly the reference's control flow), grid for grid; the GPU just never waits for the host."""
    thr = processor.auto_correct_threshold if auto_correct_threshold is None else auto_correct_threshold
    stats = {"grids_processed": 0, "cells_processed": 0, "cells_classified_noise": 0, "cells_corrected": 0,
             "total_confidence": 0.0}
    nodata = getattr(handler, "NODATA", 1.0e6)
    pending = []

    def apply_one(grid, classification, confidence, correction):
        if results_sink is not None:
            results_sink(grid, classification, confidence, correction)
        depth = grid.depth.copy(); unc = grid.uncertainty.copy()
        noise = (classification == processor.CLASS_NOISE) & grid.valid_mask
        applied = noise & (confidence >= thr)
        if np.any(applied):
            depth[applied] -= correction[applied]
            stats["cells_corrected"] += int(np.sum(applied))
            unc[applied] *= 2.0 - confidence[applied]
        writer.update_refinement_batch(grid, depth, unc)
        stats["grids_processed"] += 1
        stats["cells_processed"] += grid.num_valid
        stats["cells_classified_noise"] += int(np.sum(noise))
        stats["total_confidence"] += float(np.sum(confidence[grid.valid_mask]))

    def apply_all(plist, batch):
        k = 0
        for grid, immediate in plist:
            if immediate is not None:
                apply_one(grid, *immediate)
            else:
                apply_one(grid, *batch[k]); k += 1

    def flush():
        if not pending:
            return
        apply_all(pending, processor.flush_batch())
        pending.clear()

    submitted = []                                      # pending lists of the batches in flight, oldest first

    def submit():
        if not pending:
            return
        plist = list(pending); pending.clear()
        if processor.batch_pending:
            while processor.batches_in_flight >= processor.MAX_IN_FLIGHT:
                apply_all(submitted.pop(0), processor.collect_batch())
            processor.submit_batch()
            submitted.append(plist)
            while len(submitted) > 1:                   # the batch before this one: collect and apply it while this one runs
                apply_all(submitted.pop(0), processor.collect_batch())
        else:                                           # only grids without a valid cell: nothing to classify
            while submitted:
                apply_all(submitted.pop(0), processor.collect_batch())
            apply_all(plist, [])

    for grid in handler.iterate_refinements(min_valid_ratio):
        pending.append((grid, processor.add_to_batch(grid.depth, grid.uncertainty, grid.resolution, nodata=nodata)))
        if processor.batch_ready:
            submit() if pipelined else flush()
    submit() if pipelined else flush()
    while submitted:
        apply_all(submitted.pop(0), processor.collect_batch())
    stats["mean_confidence"] = stats["total_confidence"] / stats["cells_processed"] if stats["cells_processed"] > 0 else 0
    return stats
