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
        self.scratch_t = None
        # The buffers are used on the ENGINES' streams (either of two), not on the stream torch's caching allocator made them on: a
        # block it hands out may still be read by work queued on its old stream.  Growing is rare -- wait the device out once.
        torch.cuda.synchronize(self.dev)
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


class _BagSlot:
    """Staging of ONE chunk of a whole-BAG pass (``process_refinements``): a pinned record slab (the chunk's records on their way up,
    the corrected ones on their way down), the device buffers the chunk is unpacked / classified / corrected in, a pinned result
    buffer, and one small "tail" -- counters {noise, corrected, changed} int64, confidence sum float64, per-grid valid counts
    int64, per-grid keep flags uint8 -- that comes back in one copy.  Bound to one engine: everything of a chunk runs on that engine's
    stream."""

    def __init__(self, eng):
        self.eng, self.dev = eng, eng.ctx.device
        self.cap = self.gcap = self.res_cap = 0
        self.done = torch.cuda.Event()
        self.res_h = None

    def ensure(self, cells: int, grids: int, results: bool):
        pin = lambda dt, shape: torch.empty(shape, dtype=dt, pin_memory=True)
        mk = lambda dt, shape: torch.empty(shape, dtype=dt, device=self.dev)
        grown = False
        if cells > self.cap:
            cap = max(int(cells), self.cap + self.cap // 2)
            self.rec_h = pin(torch.float32, (cap, 2)); self.rec_np = self.rec_h.numpy()
            self.out_rec_h = pin(torch.float32, (cap, 2))      # (its own slab: chunk k + S is staged while chunk k is written back)
            self.rec_t, self.depth_t, self.unc_t = mk(torch.float32, (cap, 2)), mk(torch.float32, cap), mk(torch.float32, cap)
            self.mask_t, self.out_t = mk(torch.uint8, cap), mk(torch.float32, 3 * cap)
            self.cap, grown = cap, True
        if grids > self.gcap:
            gcap = max(int(grids), 2 * self.gcap, 256)
            self.off_h = pin(torch.int64, gcap + 1); self.off_np = self.off_h.numpy()
            self.off_t = mk(torch.int64, gcap + 1)
            self.tail_h = pin(torch.uint8, 32 + 9 * gcap); self.tail_t = mk(torch.uint8, 32 + 9 * gcap)
            self.gcap, grown = gcap, True
        if grown:       # (used on the engine's stream, allocated on the caller's: see _BatchSlot._grow)
            torch.cuda.synchronize(self.dev)
        if results and self.res_cap < self.cap:
            self.res_h = pin(torch.float32, 3 * self.cap); self.res_cap = self.cap

    def tail_views(self, g: int):
        """Device views of the tail for a chunk of ``g`` grids: (valid counts int64 [g], keep flags uint8 [g])."""
        return self.tail_t[32:32 + 8 * g].view(torch.int64), self.tail_t[32 + 8 * g:32 + 9 * g]


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

    PIPELINE_COALESCE = 4            # reference-size batches per pipelined submission (run_refinements)

    @property
    def submit_ready(self) -> bool:
        """``batch_ready`` for a PIPELINED submission: ``PIPELINE_COALESCE`` node budgets' worth of grids -- a 50 000-node batch is one
        round of workgroups per kernel; four of them in one submission run at 1.3x the device rate and cost one host round instead of
        four.  The results of a grid do not depend on the batch it travels in."""
        return (self._batch_node_count >= self.PIPELINE_COALESCE * self.BATCH_NODE_BUDGET or
                (self._fill is not None and len(self._fill.hw) >= self.MAX_GRIDS_PER_BATCH))

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

    def _engine_at(self, i: int) -> TileBatchEngine:
        """The processor's i-th engine (0: the device's default library context; 1: a second context -- own HIP stream, own arenas --
        created when batches are first pipelined, with the first one's run-time switches)."""
        from .. import runtime as rt
        while len(self._engines) <= i:
            ctx = rt.new_context(self._engine.ctx.device)
            for k in ("matrix_path", "fused", "fold_extractor", "ragged_atlas", "fused_front", "features_tiled"):
                ctx.set_option(k, self._engine.ctx.get_option(k))
            self._engines.append(TileBatchEngine(self.model, self.graph_builder, self._engine.ctx.device, self.auto_correct_threshold,
                                                 self._engine.review_threshold, self._engine.norm_floor, ctx=ctx))
        return self._engines[i]

    def _second_engine(self) -> TileBatchEngine:
        return self._engine_at(1)

    def _engine_for_next(self) -> TileBatchEngine:
        i = self._next_engine % self.MAX_IN_FLIGHT
        self._next_engine += 1
        return self._engine_at(i)

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
                if slot.scratch_t is None or slot.scratch_t.numel() < n:
                    slot.scratch_t = torch.empty(slot.cap, dtype=torch.uint8, device=slot.dev)
                    torch.cuda.synchronize(slot.dev)                                              # (as in _grow)
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
    BAG_SLOTS = 4                    # chunks in flight (two per library context)
    BAG_CHUNK_MIN, BAG_CHUNK_MAX = 256 << 10, 512 << 10   # automatic chunk size: half of the BAG, within these cell counts

    def _bag_threads(self):
        """The two helper threads of the whole-BAG path (staging copies / collection + write-back), made on first use."""
        t = self.__dict__.get("_bag_pool")
        if t is None:
            from concurrent.futures import ThreadPoolExecutor
            t = self.__dict__["_bag_pool"] = (ThreadPoolExecutor(1, thread_name_prefix="bgnn-stage"),
                                              ThreadPoolExecutor(1, thread_name_prefix="bgnn-collect"))
        return t

    def _bag_slot(self, i: int, cells: int, grids: int, results: bool) -> "_BagSlot":
        slots = self.__dict__.setdefault("_bag_slots", {})
        s = slots.get(i)
        if s is None:
            eng = self._engine if i % 2 == 0 else self._second_engine()
            s = slots[i] = _BagSlot(eng)
        s.ensure(cells, grids, results)
        return s

    def process_refinements(self, handler, writer=None, min_valid_ratio: float = 0.0,
                            cell_budget: Optional[int] = None, return_results: bool = False, results_sink=None,
                            auto_correct_threshold: Optional[float] = None):
        """Classify and correct every refinement grid of a VR BAG with the records resident in HBM.

        ``varres_refinements`` is already the concatenated-grid layout ``bgnn_infer_tiles`` consumes
        (grids row-major, one after another in ``varres_metadata.index`` order), so the records are uploaded
        as they are, in chunks cut at grid boundaries (the reference batches 50 000 nodes because PyG materialises
        per-edge tensors; here a chunk is sized to keep the copies of one chunk under the kernels of another:
        ``cell_budget`` cells, default half of the BAG within ``BAG_CHUNK_MIN`` .. ``BAG_CHUNK_MAX``; measured flat between 256 k and 1 M cells).  Per chunk, all on
        one library context's stream and asynchronous to the host: H2D of the records from a pinned slab ->
        ``bgnn_vr_unpack`` (planes, valid mask, ``min_valid_ratio`` filter) -> ``bgnn_infer_tiles`` -> ``bgnn_vr_apply`` (the
        write-back arithmetic of ``apply_results``) -> D2H of the corrected records, the per-grid valid counts / keep flags
        and the counters (and, when asked for, the three result planes).  Up to ``BAG_SLOTS`` chunks are in flight on two
        contexts; a finished chunk goes into ``writer`` with one slice assignment and, per grid in iteration order, to
        ``results_sink(grid, classification, confidence, correction)`` (where the sidecar builder is fed).

        Returns the statistics the reference's ``main`` logs (:540-559); with ``return_results`` also per-record
        classification / confidence / correction arrays.  Results equal ``run_refinements(pipelined=False)`` (the
        reference's grid-by-grid loop) bit for bit; ``total_confidence`` is a float64 sum in device order."""
        import ctypes as C
        from .. import runtime as rt
        thr = self.auto_correct_threshold if auto_correct_threshold is None else auto_correct_threshold
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
                 and ref.dtype.fields["depth"][0] == np.dtype("<f4") and ref.dtype.fields["depth_uncrt"][0] == np.dtype("<f4"))
        start0 = int(tab["index"][0])
        index = tab["index"]
        off = np.zeros(n_grids + 1, np.int64); np.cumsum(tab["cells"], out=off[1:])
        if tab["contiguous"] and plain:
            rec_all = np.ascontiguousarray(ref[start0:start0 + total]).view(np.float32).reshape(total, 2)   # (a view when it can be)
            depth_f = unc_f = None
        else:       # records not laid out in iteration order (or a foreign record layout): gathered chunk by chunk
            rec_all = None
            depth_f, unc_f = ref["depth"], ref["depth_uncrt"]
        use_unc = self._uses_uncertainty(ref)
        hw = np.stack([tab["dims_y"], tab["dims_x"]], 1).astype(np.int32)
        res = np.stack([tab["res_x"], tab["res_y"]], 1).astype(np.float64)
        want_res = return_results or results_sink is not None
        if cell_budget is None:
            cell_budget = min(max(total // 2 + 1, self.BAG_CHUNK_MIN), self.BAG_CHUNK_MAX)
        nodata = C.c_float(getattr(handler, "NODATA", 1.0e6))
        grids_it = iter(handler.iterate_refinements(min_valid_ratio)) if results_sink is not None else None
        # ---- the chunks, and one staging slot per chunk in flight (sized once, while nothing is in flight) ----
        chunks, g0 = [], 0
        while g0 < n_grids:
            g1 = int(np.searchsorted(off, off[g0] + cell_budget, side="right")) - 1
            g1 = min(max(g1, g0 + 1), n_grids, g0 + self.MAX_GRIDS_PER_BATCH)
            chunks.append((g0, g1)); g0 = g1
        nch, S = len(chunks), self.BAG_SLOTS
        max_cells = max(int(off[b] - off[a]) for a, b in chunks); max_grids = max(b - a for a, b in chunks)
        slots = [self._bag_slot(i, max_cells, max_grids, want_res) for i in range(min(S, nch))]

        def stage(k):
            """Host copy of chunk k's records into its slot's pinned slab (on the staging thread, a chunk ahead of the launches)."""
            if k >= S:
                collected[k - S].wait()                       # the slot's previous chunk has been collected
            g0, g1 = chunks[k]
            lo, hi = int(off[g0]), int(off[g1])
            n, slot = hi - lo, slots[k % S]
            if rec_all is not None:
                np.copyto(slot.rec_np[:n], rec_all[lo:hi])
            else:
                idx = np.repeat(index[g0:g1] - (off[g0:g1] - lo), tab["cells"][g0:g1]) + np.arange(n, dtype=np.int64)
                slot.rec_np[:n, 0] = depth_f[idx]; slot.rec_np[:n, 1] = unc_f[idx]
            np.subtract(off[g0:g1 + 1], lo, out=slot.off_np[:g1 - g0 + 1])

        def launch(k):
            g0, g1 = chunks[k]
            lo, hi = int(off[g0]), int(off[g1])
            n, g, slot = hi - lo, g1 - g0, slots[k % S]
            eng, ctx = slot.eng, slot.eng.ctx
            cnt_t, keep_t = slot.tail_views(g)
            # (all on the engine's stream.  A copy stream per slot -- upload / download either side of the kernels, ordered by events --
            #  was measured SLOWER here: 1.30 instead of 1.14 ms per 256 k-cell chunk; the cross-stream waits cost more than the copies)
            with torch.cuda.stream(ctx.stream):
                slot.rec_t[:n].copy_(slot.rec_h[:n], non_blocking=True)
                slot.off_t[:g + 1].copy_(slot.off_h[:g + 1], non_blocking=True)
                slot.tail_t[:32].zero_()
                unc_t = slot.unc_t[:n] if use_unc else None
                rt.check(ctx.lib.bgnn_vr_unpack(ctx.handle, rt.ptr(slot.rec_t), n, nodata, g, rt.ptr(slot.off_t),
                                                C.c_double(min_valid_ratio), rt.ptr(slot.depth_t), rt.ptr(unc_t), rt.ptr(slot.mask_t),
                                                rt.ptr(cnt_t), rt.ptr(keep_t)))
                out = slot.out_t[:3 * n].view(3, n)
                eng.infer_device(hw[g0:g1], res[g0:g1], slot.depth_t[:n], slot.mask_t[:n], unc_t, out=out, defer_end=True, begin=False)
                rt.check(ctx.lib.bgnn_vr_apply(ctx.handle, rt.ptr(slot.rec_t), n, rt.ptr(slot.mask_t), rt.ptr(out[0]), rt.ptr(out[1]),
                                               rt.ptr(out[2]), C.c_float(thr), rt.ptr(slot.tail_t), rt.ptr(slot.tail_t[24:])))
                if writer is not None:
                    slot.out_rec_h[:n].copy_(slot.rec_t[:n], non_blocking=True)
                slot.tail_h[:32 + 9 * g].copy_(slot.tail_t[:32 + 9 * g], non_blocking=True)
                if want_res:
                    slot.res_h[:3 * n].copy_(slot.out_t[:3 * n], non_blocking=True)
                slot.done.record(ctx.stream)

        def collect(k):
            try:
                g0, g1 = chunks[k]
                lo, hi = int(off[g0]), int(off[g1])
                n, g, slot = hi - lo, g1 - g0, slots[k % S]
                slot.done.synchronize()
                tail = slot.tail_h.numpy()
                counts = tail[:24].view(np.int64)
                cnt = tail[32:32 + 8 * g].view(np.int64); keep = tail[32 + 8 * g:32 + 9 * g].view(bool)
                kept = int(np.count_nonzero(keep))
                stats["grids_processed"] += kept; stats["grids_skipped"] += g - kept
                stats["cells_processed"] += int(cnt.sum()) if kept == g else int(cnt[keep].sum())
                stats["cells_classified_noise"] += int(counts[0]); stats["cells_corrected"] += int(counts[1])
                stats["total_confidence"] += float(tail[24:32].view(np.float64)[0])
                if writer is not None:
                    changed, rec_o = int(counts[2]), slot.out_rec_h.numpy()
                    if rec_all is not None:
                        writer.write_records(start0 + lo, rec_o[:n], corrections_applied=changed)
                    else:
                        for j in range(g0, g1):
                            writer.write_records(int(index[j]), rec_o[off[j] - lo:off[j + 1] - lo],
                                                 corrections_applied=changed if j == g0 else 0)
                if want_res:
                    r = slot.res_h.numpy()[:3 * n].reshape(3, n)
                    if return_results:
                        res_all[:, lo:hi] = r
                        r = res_all[:, lo:hi]
                    elif results_sink is not None:
                        r = np.array(r)                           # (the sink may keep the arrays; the pinned buffer is reused)
                    if results_sink is not None:
                        for j in np.nonzero(keep)[0].tolist():    # the grids iterate_refinements yields, in its order
                            grid = next(grids_it)
                            a, b = int(off[g0 + j]) - lo, int(off[g0 + j + 1]) - lo
                            assert grid.start_index == int(index[g0 + j])
                            shp = grid.depth.shape
                            results_sink(grid, r[0, a:b].reshape(shp), r[1, a:b].reshape(shp), r[2, a:b].reshape(shp))
            finally:
                collected[k].set()

        # Three roles: THIS thread queues the GPU work of chunk k; a staging thread copies chunk k + 1 into its pinned slab
        # meanwhile; a collector thread waits for finished chunks and writes them back (array-backed writers only: a sink, or a
        # file-backed writer, is served from this thread).  The copies are plain memcpy's that release the GIL -- per 256 k-cell
        # chunk the host otherwise spends as long copying (2 x 2 MB) and issuing as the GPU spends classifying.
        import threading
        collected = [threading.Event() for _ in range(nch)]
        own_collect = results_sink is not None or not (writer is None or getattr(writer, "_file", 0) is None)
        stager, collector = self._bag_threads()
        f_stage, f_coll, next_collect = [None] * nch, [], 0
        try:
            f_stage[0] = stager.submit(stage, 0)
            for k in range(nch):
                if k + 1 < nch:
                    if own_collect and k + 1 >= S:            # the slot chunk k + 1 is staged into must have been collected
                        while next_collect <= k + 1 - S:
                            collect(next_collect); next_collect += 1
                    f_stage[k + 1] = stager.submit(stage, k + 1)
                f_stage[k].result()
                launch(k)
                if not own_collect:
                    f_coll.append(collector.submit(collect, k))
            if own_collect:
                while next_collect < nch:
                    collect(next_collect); next_collect += 1
            for f in f_coll:
                f.result()
        finally:                                               # (also after an exception: nothing of ours stays queued or blocked)
            for e in collected:
                e.set()
            for f in [f for f in f_stage if f is not None] + f_coll:
                try:
                    f.result()
                except Exception:
                    pass
            for slot in slots:
                slot.done.synchronize()
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
                    auto_correct_threshold: Optional[float] = None, results_sink=None, pipelined: bool = True,
                    records_resident: Optional[bool] = None):
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
    logged mean only): equal to ~1e-7 relative, not bit for bit.

    ``records_resident`` (default: decided here): when the handler exposes the BAG's two arrays (``refinement_table()`` /
    ``varres_refinements``: ``VRBagHandler``) and the writer takes records in bulk (``write_records``), the pipelined call does not
    walk the grids in Python at all: the records go to the GPU as they are and come back corrected
    (``NativeVRProcessor.process_refinements``); the sink is still fed grid by grid, in iteration order.  Any other handler /
    writer (e.g. ``SRBagHandler``, the reference's own classes) takes the loop below, where a pipelined submission gathers
    ``processor.PIPELINE_COALESCE`` reference-size batches (the results do not depend on how grids are batched)."""
    thr = processor.auto_correct_threshold if auto_correct_threshold is None else auto_correct_threshold
    can_route = (hasattr(handler, "refinement_table") and hasattr(handler, "varres_refinements") and hasattr(writer, "write_records")
                 and hasattr(processor, "process_refinements"))
    if records_resident is None:
        records_resident = pipelined and can_route
    if records_resident:
        if not can_route:
            raise ValueError("records_resident=True needs a handler with refinement_table() / varres_refinements and a writer with write_records()")
        st = processor.process_refinements(handler, writer, min_valid_ratio, results_sink=results_sink, auto_correct_threshold=thr)
        st.pop("grids_skipped", None)                      # (the loop's statistics, key for key)
        return st
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
        a run of consecutive grids: then this is a slice of that plane, not a concatenation.  Only when EVERY grid starts where the
        one before it ends (the test update_refinements_bulk makes too): a BAG whose metadata index is not monotonic in iteration
        order can fill the same total span with its grids in another order."""
        a0 = getattr(gl[0], name)
        base = a0.base
        if base is not None and base.ndim == 1 and base.flags.c_contiguous:
            lo = gl[0].start_index
            pos, ok = lo, 0 <= lo and a0.ctypes.data == base.ctypes.data + lo * base.itemsize
            if ok:
                for g in gl:
                    a = getattr(g, name)
                    if g.start_index != pos or a.base is not base:
                        ok = False
                        break
                    pos += a.size
            if ok and pos <= base.shape[0]:
                return base[lo:pos]
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
            if processor.submit_ready:
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
