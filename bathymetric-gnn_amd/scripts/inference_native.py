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


class _BatchSlot:
    """Staging of ONE refinement-grid batch: a pinned host slab of {depth, depth_uncrt} records the grids are copied into as they are
    queued (``add_to_batch``), the device buffers the batch is unpacked / classified in, and a pinned result buffer.  A slot is
    taken when the first grid of a batch is queued and returns to the processor's pool when the batch has been collected."""

    def __init__(self, dev, cells: int = 1 << 17):
        self.dev = dev
        self.cap = 0
        self.down = torch.cuda.Event()                   # recorded behind the batch's D2H on the engine's stream
        self.mask_h = None                               # pinned uint8 [cap]: only for batches that mix nodata values
        self._grow(cells)
        self.reset()

    def _grow(self, cells: int):
        cap = max(int(cells), 2 * self.cap)
        rec_h = torch.empty((cap, 2), dtype=torch.float32, pin_memory=True)
        if self.cap:
            rec_h[:self.n] = self.rec_h[:self.n]
            if self.mask_h is not None:
                m = torch.empty(cap, dtype=torch.uint8, pin_memory=True); m[:self.n] = self.mask_h[:self.n]; self.mask_h = m
        self.rec_h, self.rec_np = rec_h, rec_h.numpy()
        # (flat: a batch of n cells uses the first 3 n floats as a CONTIGUOUS [3, n] block, so that the results come back in one copy)
        self.out_h = torch.empty(3 * cap, dtype=torch.float32, pin_memory=True)
        mk = lambda dt, shape: torch.empty(shape, dtype=dt, device=self.dev)
        self.rec_t, self.depth_t, self.unc_t = mk(torch.float32, (cap, 2)), mk(torch.float32, cap), mk(torch.float32, cap)
        self.mask_t, self.out_t = mk(torch.uint8, cap), mk(torch.float32, 3 * cap)
        self.cap = cap

    def reset(self):
        self.n = 0                                       # cells staged
        self.hw, self.res = [], []                       # per grid: (h, w), (rx, ry)
        self.has_unc = None
        self.nodata = None
        self.host_masks = False                          # True: masks were computed on the host (mixed nodata values)
        self.engine = None

    def stage(self, depth: np.ndarray, unc: Optional[np.ndarray], resolution, nodata: float, host_mask: Optional[np.ndarray]):
        h, w = depth.shape
        n = h * w
        if self.n + n > self.cap:
            self._grow(self.n + n)
        if self.has_unc is None:
            self.has_unc, self.nodata = unc is not None, float(nodata)
        elif self.has_unc != (unc is not None):
            raise ValueError("either every grid of a batch has an uncertainty layer or none has")
        if float(nodata) != self.nodata and not self.host_masks:
            # a grid with ANOTHER nodata value joins the batch: the device-side mask (one nodata per batch) no longer covers it --
            # this batch's masks are made on the host from here on (the cells staged so far: one vectorised pass, their own nodata)
            if self.mask_h is None or self.mask_h.numel() < self.cap:
                self.mask_h = torch.empty(self.cap, dtype=torch.uint8, pin_memory=True)
            d0 = self.rec_np[:self.n, 0]
            self.mask_h.numpy()[:self.n] = (d0 != np.float32(self.nodata)) & np.isfinite(d0)
            self.host_masks = True
        rec = self.rec_np[self.n:self.n + n]
        np.copyto(rec[:, 0].reshape(h, w), depth, casting="same_kind")
        if unc is not None:
            np.copyto(rec[:, 1].reshape(h, w), unc, casting="same_kind")
        if self.host_masks:
            m = host_mask if host_mask is not None else ((depth != nodata) & np.isfinite(depth))
            self.mask_h.numpy()[self.n:self.n + n] = m.reshape(-1)
        self.hw.append((h, w)); self.res.append((float(resolution[0]), float(resolution[1])))
        self.n += n


class NativeVRProcessor:
    CLASS_NOISE = 2
    BATCH_NODE_BUDGET = 50000        # nodes to accumulate before a flush (reference :128)
    MAX_GRIDS_PER_BATCH = 32768      # the library takes at most 60 000 grids per batch (a BAG of 3 x 3 grids reaches that before any cell budget)

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
        # The queued grids live in a pinned host slab (a _BatchSlot), not in a Python list of arrays: add_to_batch copies a grid's
        # depth / uncertainty straight into the slab (the {depth, depth_uncrt} record layout bgnn_vr_unpack reads), the valid mask
        # (depth != nodata and finite) is made ON THE DEVICE by that kernel, and a flush is one H2D, the kernels and one D2H.
        self._fill: Optional[_BatchSlot] = None       # the batch being queued
        self._free_slots: List[_BatchSlot] = []
        self._batch_node_count = 0
        # Two batches in flight (submit_batch / collect_batch): a 50 000-node batch is ONE round of workgroups per kernel, so its
        # twelve launches are a chain of latencies that leaves most of the GPU idle; the next batch runs on a second library
        # context (own HIP stream, own arenas) beside it.  flush_batch() stays the reference's synchronous call.
        self._engines = [self._engine]
        self._inflight: List[_BatchSlot] = []         # submitted batches, oldest first
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

    def _uses_uncertainty(self, uncertainty) -> bool:
        """Does a queued grid's uncertainty layer reach the model?  (:165-167: dropped for a 7-channel model; and only a model
        whose width is the builder's column count WITH the uncertainty column can take it.)"""
        if uncertainty is None:
            return False
        t = self.__dict__.get("_takes_unc")
        if t is None:
            t = self.__dict__["_takes_unc"] = (self.expected_in_channels != 7 and
                                               self.model.in_channels == self.graph_builder.n_node_columns(True))
        return t

    # ---- reference API ---------------------------------------------------------------------
    def process_grid(self, depth: np.ndarray, uncertainty: Optional[np.ndarray], resolution: tuple,
                     nodata: float = 1.0e6) -> Result:
        """One refinement grid, unbatched (:206-247): (classification, confidence, correction)."""
        item = self._prepare(depth, uncertainty, resolution, nodata)
        if item is None:
            return self._empty(depth)
        return self._run([item])[0]

    def add_to_batch(self, depth, uncertainty, resolution, nodata=1.0e6, valid_count: Optional[int] = None):
        """Queue a grid (:249-269).  Returns None when queued, or the all-zero result tuple immediately for a grid with no valid
        cell.  The grid is copied into the batch's pinned staging slab here; nothing of it is kept by reference.

        ``valid_count`` (optional, not in the reference's signature): the number of valid cells, when the caller already knows it
        (``RefinementGrid.num_valid``) -- the per-grid mask / count pass over the array is then skipped, the mask itself being
        made on the device at flush time."""
        depth = np.asarray(depth)
        if depth.dtype != np.float32:
            depth = depth.astype(np.float32)
        mask = None
        if valid_count is None:
            mask = (depth != nodata) & np.isfinite(depth)            # :160
            valid_count = int(np.count_nonzero(mask))
        if valid_count == 0:
            return self._empty(depth)
        if self._fill is None:
            self._fill = self._free_slots.pop() if self._free_slots else _BatchSlot(self._engine.ctx.device)
            self._fill.reset()
        unc = uncertainty if self._uses_uncertainty(uncertainty) else None
        if unc is not None:
            unc = np.asarray(unc)
            if unc.shape != depth.shape:
                raise ValueError(f"uncertainty shape {unc.shape} != depth shape {depth.shape}")
        self._fill.stage(depth, unc, resolution, nodata, mask)
        self._batch_node_count += int(valid_count)
        return None

    @property
    def batch_ready(self) -> bool:
        return (self._batch_node_count >= self.BATCH_NODE_BUDGET or
                (self._fill is not None and len(self._fill.hw) >= self.MAX_GRIDS_PER_BATCH))     # (a raised budget and tiny grids)

    @property
    def batch_pending(self) -> bool:
        return self._fill is not None and self._fill.n > 0

    def flush_batch(self) -> List[Result]:
        """Classify every queued grid in one fused pass (:281-342); results in insertion order.  Synchronous, like the
        reference's (batches still in flight from ``submit_batch`` are not disturbed: this one queues behind the first
        context's work and their results stay available to ``collect_batch``)."""
        if not self.batch_pending:
            return []
        slot = self._launch(self._engine)
        return self._results_of(slot, *self._finish(slot))

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

    def _launch(self, eng: TileBatchEngine) -> _BatchSlot:
        """Everything of the queued batch that runs on the GPU, asynchronously: H2D of the record slab, unpack + valid mask
        (``bgnn_vr_unpack``), the fused classification (``bgnn_infer_tiles``), D2H of the three result planes into the slot's
        pinned buffer.  The caller's torch stream is not involved: nothing it does later waits for this."""
        import ctypes as C
        from .. import runtime as rt
        slot, self._fill, self._batch_node_count = self._fill, None, 0
        n, ctx = slot.n, eng.ctx
        slot.engine = eng
        slot.hw_np = np.array(slot.hw, np.int32).reshape(-1, 2)
        slot.res_np = np.array(slot.res, np.float64).reshape(-1, 2)
        # Everything goes onto the ENGINE's stream, in order -- H2D of the record slab, unpack + mask, the fused classification, D2H of
        # the results: one stream switch per batch.  (Batches alternate between two engines; a copy stream per slot bought nothing
        # measurable -- the upload is ~0.5 MB -- and cost two more stream waits and an event per batch on the host.)
        with torch.cuda.stream(ctx.stream):
            slot.rec_t[:n].copy_(slot.rec_h[:n], non_blocking=True)
            if slot.host_masks:
                slot.mask_t[:n].copy_(slot.mask_h[:n], non_blocking=True)
            unc_t = slot.unc_t[:n] if slot.has_unc else None
            # (host masks -- a batch that mixes nodata values: the kernel's own mask goes to a scratch plane and is not used)
            mask_out = slot.mask_t[:n]
            if slot.host_masks:
                if getattr(slot, "scratch_t", None) is None or slot.scratch_t.numel() < n:
                    slot.scratch_t = torch.empty(slot.cap, dtype=torch.uint8, device=slot.dev)
                mask_out = slot.scratch_t[:n]
            rt.check(ctx.lib.bgnn_vr_unpack(ctx.handle, rt.ptr(slot.rec_t), n, C.c_float(slot.nodata), 0, None, C.c_double(0.0),
                                            rt.ptr(slot.depth_t), rt.ptr(unc_t), rt.ptr(mask_out), None, None))
            # (begin=False: the inputs were made on the engine's own stream -- nothing of the caller's torch stream to wait for)
            out_t = slot.out_t[:3 * n].view(3, n)
            eng.infer_device(slot.hw_np, slot.res_np, slot.depth_t[:n], slot.mask_t[:n], unc_t, out=out_t, defer_end=True, begin=False)
            # (ONE contiguous copy = one plain hipMemcpyAsync.  A strided [3, n] view of a [3, cap] buffer would go through a temporary
            #  and a CPU-side at::parallel_for: synchronous, and its OpenMP team -- 128 threads on the GPU box, spinning after the
            #  region -- eats the container's CPU quota: the whole process then stalled ~90 ms at a time, anywhere)
            slot.out_h[:3 * n].copy_(slot.out_t[:3 * n], non_blocking=True)
            slot.down.record(ctx.stream)
        return slot

    def _finish(self, slot: _BatchSlot, copy: bool = True):
        """Wait for a launched batch; returns (flat results [3, cells], hw) and puts the slot back into the pool.  ``copy`` (default):
        the results are a FRESH array (one copy per batch); ``copy=False``: a view of the slot's pinned result buffer, valid until
        the next ``submit_batch`` / ``flush_batch`` (the next D2H into that buffer cannot be queued earlier)."""
        slot.down.synchronize()
        flat = slot.out_h.numpy()[:3 * slot.n].reshape(3, slot.n)
        if copy:
            flat = np.array(flat)
        hw = slot.hw
        slot.reset()
        self._free_slots.append(slot)
        return flat, hw

    @staticmethod
    def _results_of(slot, flat, hw) -> List[Result]:
        """Per-grid (classification, confidence, correction): views of the batch's own result array (no per-grid copies)."""
        results, off = [], 0
        c0, c1, c2 = flat[0], flat[1], flat[2]
        for h, w in hw:
            e = off + h * w
            results.append((c0[off:e].reshape(h, w), c1[off:e].reshape(h, w), c2[off:e].reshape(h, w)))
            off = e
        return results

    def submit_batch(self) -> Optional[int]:
        """Start classifying the queued grids WITHOUT waiting for the result: the batch is uploaded and its kernels are queued on
        one of two library contexts, alternately, so that it runs beside the batch submitted before it.  Returns the number of
        batches now in flight (None if nothing was queued).  Results come back, in submission order, from ``collect_batch``.
        At most ``MAX_IN_FLIGHT`` batches may be outstanding: collect the oldest first."""
        if not self.batch_pending:
            return None
        if len(self._inflight) >= self.MAX_IN_FLIGHT:
            raise RuntimeError(f"{self.MAX_IN_FLIGHT} batches are already in flight: collect_batch() the oldest first")
        self._inflight.append(self._launch(self._engine_for_next()))
        return len(self._inflight)

    @property
    def batches_in_flight(self) -> int:
        return len(self._inflight)

    def collect_batch_flat(self, copy: bool = True):
        """Results of the OLDEST batch in flight as ONE array: (flat [3, cells] float32 -- classification, confidence, correction
        of the batch's grids back to back, row-major --, [(h, w) per grid]); (None, []) when nothing is in flight.  ``copy=False``: a
        view of the pinned result buffer, valid only until the next ``submit_batch`` / ``flush_batch``."""
        if not self._inflight:
            return None, []
        return self._finish(self._inflight.pop(0), copy=copy)

    def collect_batch(self) -> List[Result]:
        """Results of the OLDEST batch in flight (blocks until it is done); same per-grid tuples as ``flush_batch``."""
        if not self._inflight:
            return []
        slot = self._inflight.pop(0)
        return self._results_of(slot, *self._finish(slot))

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
            g1 = min(max(g1, g0 + 1), n_grids, g0 + self.MAX_GRIDS_PER_BATCH)
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
    processor's second library context.  The write-back arithmetic of a collected batch then runs ONCE over the batch's cells
    (the same float32 operations as ``apply_one``, element for element, on the concatenated grids) and the corrected values go
    back through the writer's bulk entry (``update_refinements_bulk``) where it has one: per grid the host only iterates and
    queues.  Same records, same sink calls in the same order and the same counts as the synchronous loop
    (``pipelined=False``: one ``flush_batch`` per full batch, exactly the reference's control flow, grid for grid);
    ``total_confidence`` is summed per batch in float64 instead of per grid in numpy's float32 pairwise order (it feeds the
    logged mean only): equal to ~1e-7 relative, not bit for bit."""
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

    bulk = getattr(writer, "update_refinements_bulk", None)

    def flat_of(gl, name):
        """The grids' arrays back to back.  Grids of one handler are views of ONE plane (iterate_refinements) and a batch is usually
        a run of consecutive grids: then this is a slice of that plane, not a concatenation."""
        a0, a1 = getattr(gl[0], name), getattr(gl[-1], name)
        base = a0.base
        if base is not None and a1.base is base and base.ndim == 1 and base.flags.c_contiguous:
            lo = gl[0].start_index
            hi = gl[-1].start_index + a1.size
            if hi - lo == sum(g.depth.size for g in gl) and 0 <= lo and hi <= base.shape[0] and \
                    a0.ctypes.data == base.ctypes.data + lo * base.itemsize:
                return base[lo:hi]
        return np.concatenate([getattr(g, name).reshape(-1) for g in gl])

    def apply_flat(plist, flat, hw):
        """apply_one over a whole collected batch at once: ``flat`` [3, cells] holds the batch's grids back to back.  Element for
        element the float32 operations of apply_one (masked ufuncs instead of boolean gathers / scatters)."""
        gl = [g for g, imm in plist if imm is None]
        if gl:
            assert len(gl) == len(hw)
            depth, unc = flat_of(gl, "depth"), flat_of(gl, "uncertainty")
            cls, conf, corr = flat[0], flat[1], flat[2]
            valid = (depth != 1.0e6) & np.isfinite(depth)              # RefinementGrid.valid_mask
            noise = (cls == processor.CLASS_NOISE) & valid
            applied = noise & (conf >= thr)
            new_depth, new_unc = depth.copy(), unc.copy()
            np.subtract(new_depth, corr, out=new_depth, where=applied)             # depth[applied] -= correction[applied]
            np.multiply(new_unc, 2.0 - conf, out=new_unc, where=applied)           # unc[applied] *= 2.0 - confidence[applied]
            stats["cells_corrected"] += int(np.count_nonzero(applied))
            stats["cells_processed"] += int(np.count_nonzero(valid))
            stats["cells_classified_noise"] += int(np.count_nonzero(noise))
            stats["total_confidence"] += float(np.sum(conf, where=valid, dtype=np.float64))
            if bulk is not None:
                bulk(gl, new_depth, new_unc, changed=int(np.count_nonzero((new_depth != depth) & valid)))
        stats["grids_processed"] += len(plist)
        if results_sink is None and bulk is not None:
            return                                       # (grids without a valid cell: nothing changes, nothing to write)
        k, off = 0, 0
        for grid, immediate in plist:
            if immediate is not None:
                if results_sink is not None:
                    results_sink(grid, *immediate)
                if bulk is None:
                    writer.update_refinement_batch(grid, grid.depth.copy(), grid.uncertainty.copy())
                continue
            h, w = hw[k]
            e = off + h * w
            if results_sink is not None:
                results_sink(grid, flat[0, off:e].reshape(h, w), flat[1, off:e].reshape(h, w), flat[2, off:e].reshape(h, w))
            if bulk is None:
                writer.update_refinement_batch(grid, new_depth[off:e].reshape(h, w), new_unc[off:e].reshape(h, w))
            k += 1; off = e

    def flush():
        if not pending:
            return
        apply_all(pending, processor.flush_batch())
        pending.clear()

    submitted = []                                      # pending lists of the batches in flight, oldest first

    def collect_oldest():
        plist = submitted.pop(0)
        # (a view of the pinned result buffer is enough: the batch is applied here, before anything else is submitted -- unless a
        #  sink may keep the arrays)
        apply_flat(plist, *processor.collect_batch_flat(copy=results_sink is not None))

    def submit():
        if not pending:
            return
        plist = list(pending); pending.clear()
        if processor.batch_pending:
            while processor.batches_in_flight >= processor.MAX_IN_FLIGHT:
                collect_oldest()
            processor.submit_batch()
            submitted.append(plist)
            while len(submitted) > 1:                   # the batch before this one: collect and apply it while this one runs
                collect_oldest()
        else:                                           # only grids without a valid cell: nothing to classify
            while submitted:
                collect_oldest()
            apply_flat(plist, None, [])

    if pipelined:
        for grid in handler.iterate_refinements(min_valid_ratio):
            # (the iterator already knows the valid-cell count: no second mask pass in add_to_batch)
            pending.append((grid, processor.add_to_batch(grid.depth, grid.uncertainty, grid.resolution, nodata=nodata,
                                                         valid_count=grid.num_valid if nodata == 1.0e6 else None)))
            if processor.batch_ready:
                submit()
        submit()
        while submitted:
            collect_oldest()
    else:
        for grid in handler.iterate_refinements(min_valid_ratio):
            pending.append((grid, processor.add_to_batch(grid.depth, grid.uncertainty, grid.resolution, nodata=nodata)))
            if processor.batch_ready:
                flush()
        flush()
    stats["mean_confidence"] = stats["total_confidence"] / stats["cells_processed"] if stats["cells_processed"] > 0 else 0
    return stats
