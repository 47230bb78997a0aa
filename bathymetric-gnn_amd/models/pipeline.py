"""Tile-batch inference -- drop-in for the reference's ``models/pipeline.py``.

``TileBatchEngine`` is the MI355X-native core: a batch of tiles goes through ONE fused call
(``bgnn_infer_tiles``: graph build -> forward -> node-to-grid scatter + correction
de-normalisation) and comes back as three [h, w] grids per tile.  ``BathymetricPipeline`` keeps the
reference's class surface (``__init__``, ``load_model``, ``process``, ``_process_tile``,
``_apply_corrections``) on top of it; ``process_grid`` is the in-memory entry (file I/O needs GDAL,
which is outside the path).
"""
from __future__ import annotations

import ctypes as C
import logging
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .. import runtime as rt
from ..config import Config
from ..config.constants import CORRECTION_NORM_FLOOR
from ..data import BathymetricGrid, GraphBuilder, TileManager, TileMerger
from .gnn import BathymetricGNN

logger = logging.getLogger(__name__)


def shard_info():
    """(rank, world_size) of the tile-parallel job: torch.distributed if initialised, else (0, 1)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


# Channel order of a packed tile-result block: ONE order on every rank, whether or not the rank holds tiles.
TILE_RESULT_CHANNELS = ("classification", "cleaned_depth", "confidence", "correction")


def exchange_tile_results(mine: dict) -> dict:
    """Union of every rank's {tile index: result dict} for the host-stitch path (every rank merges, so every rank needs
    every tile).  The grids travel as ONE float32 tensor per rank (``all_gather`` of [count, channels, h, w] blocks padded
    to the largest count; the tile indices as an int64 tensor) -- no pickling.  The data path itself (graph build,
    forward, scatter) has no collective.  For large surveys use the device path (row bands + halo rows), which moves
    ~0.5 GB per band boundary instead of every tile to every rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return mine
    world = dist.get_world_size()
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    keys = []
    if mine:
        have = set(next(iter(mine.values())).keys())
        keys = [k for k in TILE_RESULT_CHANNELS if k in have] + sorted(have - set(TILE_RESULT_CHANNELS))
    shape = tuple(next(iter(mine.values()))[keys[0]].shape) if mine else (0, 0)
    # agree on (channel names are fixed by the caller) count, tile shape
    # (last entry: are this rank's channels the standard ones in the standard order?  A rank WITHOUT tiles can only unpack those.)
    std = int(keys == list(TILE_RESULT_CHANNELS)[:len(keys)])
    meta = torch.tensor([len(mine), len(keys), shape[0], shape[1], std], dtype=torch.int64, device=dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = [m.cpu().numpy() for m in metas]
    nmax = int(max(m[0] for m in metas))
    nk, th, tw = (int(max(m[i] for m in metas)) for i in (1, 2, 3))
    if nmax == 0:
        return mine
    # Every check below reads only `metas`, which all ranks hold alike, and runs BEFORE the next collective: the ranks fail
    # together instead of one raising while the others wait in the block all_gather.
    if not all(m[0] == 0 or (m[1], m[2], m[3]) == (nk, th, tw) for m in metas):
        raise ValueError("exchange_tile_results: tiles of one survey share one shape and one channel count")
    if any(m[0] == 0 for m in metas) and (nk > len(TILE_RESULT_CHANNELS) or any(m[0] > 0 and m[4] == 0 for m in metas)):
        raise ValueError(f"exchange_tile_results: a rank without tiles can only unpack the standard channels {TILE_RESULT_CHANNELS} "
                         "in that order, and another rank packed something else")
    if not keys:                                           # this rank holds no tile: it still takes part in the collective,
        keys = list(TILE_RESULT_CHANNELS)[:nk]             # and unpacks in the same channel order the packing ranks used
    idx = torch.full((nmax,), -1, dtype=torch.int64)
    blk = torch.zeros((nmax, nk, th, tw), dtype=torch.float32)
    for j, (i, r) in enumerate(sorted(mine.items())):
        idx[j] = i
        for c, k in enumerate(keys):
            blk[j, c] = torch.from_numpy(np.ascontiguousarray(r[k], dtype=np.float32))
    idx, blk = idx.to(dev), blk.to(dev)
    idxs = [torch.empty_like(idx) for _ in range(world)]
    blks = [torch.empty_like(blk) for _ in range(world)]
    dist.all_gather(idxs, idx)
    dist.all_gather(blks, blk)
    out = {}
    for ii, bb in zip(idxs, blks):
        ii = ii.cpu().numpy(); bb = bb.cpu().numpy()
        for j, i in enumerate(ii):
            if i >= 0:
                out[int(i)] = {k: bb[j, c] for c, k in enumerate(keys)}
    return out


def gather_bands_to_rank0(plan, rank: int, band: Optional[torch.Tensor], width: int, channels: int = 4, device=None):
    """The sharded survey path's last step: every rank's stitched band [channels, rows, width] (float32) goes to rank 0 by
    point-to-point ``send`` / ``recv`` of the tensor itself (RCCL under ``nccl``; staged through the host under ``gloo``),
    one message per band -- not an all-gather of pickled arrays to everybody.  Rank 0 returns the assembled
    [channels, H, width] host array, the other ranks None."""
    import torch.distributed as dist
    via_host = dist.get_backend() == "gloo"
    H = max(p["cell_rows"][1] for p in plan)
    if rank != 0:
        R0, R1 = plan[rank]["cell_rows"]
        if R1 > R0:
            t = band.contiguous()
            dist.send(t.cpu() if via_host else t, dst=0)
        return None
    host = np.empty((channels, H, width), np.float32)
    R0, R1 = plan[0]["cell_rows"]
    if R1 > R0:
        host[:, R0:R1] = band.cpu().numpy()
    if via_host:
        recv_dev = torch.device("cpu")
    elif device is not None:
        recv_dev = torch.device(device)
    elif band is not None:
        recv_dev = band.device
    else:
        recv_dev = torch.device("cuda", torch.cuda.current_device())
    for k in range(1, len(plan)):
        R0, R1 = plan[k]["cell_rows"]
        if R1 == R0:
            continue
        buf = torch.empty((channels, R1 - R0, width), dtype=torch.float32, device=recv_dev)
        dist.recv(buf, src=k)
        host[:, R0:R1] = buf.cpu().numpy()
    return host


def survey_shard_plan(row_start, row_end, height: int, world: int):
    """Row-band ownership of a tiled survey over ``world`` GPUs (contiguous row bands localise the
    stitch).  Tile rows are split into contiguous blocks; rank k owns tile rows ``[a_k, b_k)`` and the survey cell
    rows ``[R_k, R_{k+1})`` with ``R_k = row_start[a_k]`` (``R_0 = 0``, ``R_world = height``).  The cells of a band
    are covered by the rank's own tile rows plus the earlier tile rows that reach into it (``row_end > R_k``):
    those are the only results that cross GPUs.  Returns a list of dicts
    ``{"tile_rows": (a, b), "cell_rows": (R0, R1), "need": [(src_rank, tile_row), ...]}``."""
    rs = np.asarray(row_start, np.int64); re = np.asarray(row_end, np.int64)
    ntr = len(rs)
    cuts = [int(x) for x in np.linspace(0, ntr, world + 1).round()]
    owner = np.empty(ntr, np.int64)
    for k in range(world):
        owner[cuts[k]:cuts[k + 1]] = k
    plan = []
    for k in range(world):
        a, b = cuts[k], cuts[k + 1]
        if a == b:                                     # more GPUs than tile rows: nothing to own
            plan.append({"tile_rows": (a, b), "cell_rows": (height, height), "need": []})
            continue
        R0 = 0 if a == 0 else int(rs[a])
        nxt = next((cuts[j] for j in range(k + 1, world) if cuts[j] < cuts[j + 1]), ntr)
        R1 = height if nxt >= ntr else int(rs[nxt])
        need = [(int(owner[t]), int(t)) for t in range(0, a) if re[t] > R0]
        plan.append({"tile_rows": (a, b), "cell_rows": (R0, R1), "need": need})
    return plan


def exchange_halo_tile_rows(plan, rank: int, local_rows: Dict[int, torch.Tensor]) -> Dict[int, torch.Tensor]:
    """Point-to-point exchange of the tile-row result blocks ``survey_shard_plan`` lists under ``need``: every rank
    sends each of its tile rows that a later band needs and receives the ones its own band needs.  ``local_rows``
    maps tile row -> tensor (all of one shape and dtype); returns {tile row: tensor} for the received rows.  This
    is the path's only inter-GPU traffic (RCCL send/recv over xGMI under the ``nccl`` backend; staged through the
    host under ``gloo``)."""
    import torch.distributed as dist
    if not plan[rank]["need"] and not any(src == rank for p in plan for src, _ in p["need"]):
        return {}
    via_host = dist.get_backend() == "gloo"
    ops, recv, keep = [], {}, []
    for dst, p in enumerate(plan):                       # same deterministic order on every rank
        for src, t in p["need"]:
            if src == rank and dst != rank:
                buf = local_rows[t].contiguous()
                buf = buf.cpu() if via_host else buf
                keep.append(buf)
                ops.append(dist.P2POp(dist.isend, buf, dst, tag=t))
            elif dst == rank and src != rank:
                shape, dtype, dev = plan[rank]["_halo_shape"], plan[rank]["_halo_dtype"], plan[rank]["_halo_device"]
                buf = torch.empty(shape, dtype=dtype, device="cpu" if via_host else dev)
                recv[t] = buf
                ops.append(dist.P2POp(dist.irecv, buf, src, tag=t))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if via_host:
        recv = {t: b.to(plan[rank]["_halo_device"]) for t, b in recv.items()}
    return recv


class TileBatchEngine:
    """Fused per-batch inference on one GPU (one engine per worker / device)."""

    def __init__(self, model: BathymetricGNN, graph_builder: GraphBuilder, device=None,
                 auto_correct_threshold: float = 0.85, review_threshold: float = 0.6,
                 norm_floor: float = CORRECTION_NORM_FLOOR, ctx: Optional[rt.Context] = None):
        self.model = model
        self.graph_builder = graph_builder
        # ``ctx``: an extra library context (``rt.new_context``) -- engines on different contexts overlap their batches
        self.ctx = ctx if ctx is not None else rt.get_context(device if device is not None else graph_builder._device)
        self.auto_correct_threshold = auto_correct_threshold
        self.review_threshold = review_threshold
        self.norm_floor = norm_floor

    def infer_device(self, hw: np.ndarray, res: np.ndarray, depth_t: torch.Tensor, mask_t: torch.Tensor,
                     unc_t: Optional[torch.Tensor], out=None,
                     n_nodes_out: Optional[torch.Tensor] = None, defer_end: bool = False, begin: bool = True):
        """Device-resident tiles in, device-resident grids out: returns float32 [3, cells]
        (classification, confidence, correction), same cell layout as ``depth_t``.  Asynchronous
        with respect to the host.  ``defer_end``: do not order the caller's torch stream behind this batch yet (the
        caller calls ``engine.ctx.end()`` before it reads ``out``): batches given to engines on different contexts
        then run concurrently.  ``begin=False``: do not order this batch behind the caller's current torch stream either (the
        inputs were produced on the engine's own stream, or behind an event it already waits for)."""
        ctx = self.ctx
        cells = depth_t.numel()
        if out is None:
            out = torch.empty((3, cells), dtype=torch.float32, device=ctx.device)
        # `out`: a [3, cells] tensor, or three 1-D [cells] tensors (classification, confidence, correction)
        assert all(o.is_contiguous() and o.numel() == cells for o in (out[0], out[1], out[2]))
        tiles, keep = rt.make_tiles(hw, res, depth_t, mask_t, unc_t)
        model_h = self.model.native(ctx, int(self.graph_builder._opts.n_edge_features))
        if begin:
            ctx.begin()
        rt.check(ctx.lib.bgnn_infer_tiles(
            ctx.handle, model_h, C.byref(tiles), C.byref(self.graph_builder._opts),
            C.c_float(self.auto_correct_threshold), C.c_float(self.review_threshold), C.c_float(self.norm_floor),
            rt.ptr(out[0]), rt.ptr(out[1]), rt.ptr(out[2]), rt.ptr(n_nodes_out)))
        if not defer_end:
            ctx.end()
        return out

    def infer(self, depths: Sequence[np.ndarray], masks: Sequence[Optional[np.ndarray]],
              uncs: Optional[Sequence[Optional[np.ndarray]]], resolutions) -> List[Dict[str, np.ndarray]]:
        """Host grids in, per-tile dicts of host grids out."""
        if len(depths) == 0:
            return []
        use_unc = uncs if (uncs is not None and self.model.in_channels == self.graph_builder.n_node_columns(True)
                           and any(u is not None for u in uncs)) else None
        hw, res, d, m, u = self.graph_builder.upload_tiles(depths, masks, use_unc, resolutions)
        out = self.infer_device(hw, res, d, m, u).cpu().numpy()
        results, off = [], 0
        for i in range(hw.shape[0]):
            h, w = int(hw[i, 0]), int(hw[i, 1])
            n = h * w
            results.append({"classification": out[0, off:off + n].reshape(h, w).copy(),
                            "confidence": out[1, off:off + n].reshape(h, w).copy(),
                            "correction": out[2, off:off + n].reshape(h, w).copy()})
            off += n
        return results


class HostTilePipeline:
    """Host tiles in, host grids out, with both PCIe crossings overlapped with compute.

    The reference's ``_process_tile`` hands host arrays across the boundary one tile at a time
    (models/pipeline.py:253-262 ``.to(device)``, :288-295 ``.cpu()``).  Here a batch of equally sized tiles
    is staged in pinned host memory, copied on a dedicated H2D stream, classified on the engine's stream and
    copied back on a D2H stream; ``depth`` slots (default 2) of pinned + device buffers let batch i+1 upload
    and batch i-1 download while batch i computes.  ``submit`` enqueues a batch and returns the result of the
    batch submitted ``depth`` calls earlier (or None); ``drain`` yields what is still in flight, in order.
    Result arrays are views of a pinned output buffer that stays untouched until the NEXT call of ``submit`` /
    ``drain`` step (there is one output buffer more than there are slots): use or copy them before that."""

    def __init__(self, engine: TileBatchEngine, n_tiles: int, h: int, w: int, with_uncertainty: bool = False,
                 resolution=(1.0, 1.0), depth: int = 2):
        self.eng, self.ctx = engine, engine.ctx
        dev = self.ctx.device
        self.n, self.h, self.w = int(n_tiles), int(h), int(w)
        self.cells = self.n * self.h * self.w
        self.hw = np.tile(np.array([[h, w]], np.int32), (self.n, 1))
        self.res = np.tile(np.array([[float(resolution[0]), float(resolution[1])]], np.float64), (self.n, 1))
        self.h2d, self.d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        mk = lambda dt, shape=None: torch.empty(shape or self.cells, dtype=dt, device=dev)
        pin = lambda dt, shape=None: torch.empty(shape or self.cells, dtype=dt, pin_memory=True)
        self.slots = []
        for _ in range(depth):
            self.slots.append({
                "h_depth": pin(torch.float32), "h_mask": pin(torch.uint8), "h_unc": pin(torch.float32) if with_uncertainty else None,
                "d_depth": mk(torch.float32), "d_mask": mk(torch.uint8), "d_unc": mk(torch.float32) if with_uncertainty else None,
                "d_out": mk(torch.float32, (3, self.cells)), "h_out": None,
                "up": torch.cuda.Event(), "done": torch.cuda.Event(), "down": torch.cuda.Event(), "busy": False, "tag": None})
        self._free_out = [pin(torch.float32, (3, self.cells)) for _ in range(depth + 1)]
        self._lent = None            # the output buffer whose views the caller currently holds
        self.i = 0

    def _recycle(self):
        if self._lent is not None:
            self._free_out.append(self._lent)
            self._lent = None

    def _result(self, slot):
        slot["down"].synchronize()
        slot["busy"] = False
        self._lent, slot["h_out"] = slot["h_out"], None
        o = self._lent.numpy().reshape(3, self.n, self.h, self.w)
        return slot["tag"], {"classification": o[0], "confidence": o[1], "correction": o[2]}

    def submit(self, depth: np.ndarray, mask: np.ndarray, unc: Optional[np.ndarray] = None, tag=None):
        self._recycle()                                            # views handed out by the previous call expire now
        slot = self.slots[self.i % len(self.slots)]
        self.i += 1
        ready = self._result(slot) if slot["busy"] else None       # also: its device / pinned input buffers are free again
        slot["h_depth"].numpy()[:] = np.asarray(depth, np.float32).reshape(-1)
        slot["h_mask"].numpy()[:] = np.asarray(mask).reshape(-1).view(np.uint8)
        if slot["h_unc"] is not None:
            slot["h_unc"].numpy()[:] = np.asarray(unc, np.float32).reshape(-1)
        slot["h_out"] = self._free_out.pop()                       # never the buffer just lent to the caller
        with torch.cuda.stream(self.h2d):
            slot["d_depth"].copy_(slot["h_depth"], non_blocking=True)
            slot["d_mask"].copy_(slot["h_mask"], non_blocking=True)
            if slot["h_unc"] is not None:
                slot["d_unc"].copy_(slot["h_unc"], non_blocking=True)
            slot["up"].record(self.h2d)
        with torch.cuda.stream(self.ctx.stream):                   # the engine orders its work behind the current stream
            self.ctx.stream.wait_event(slot["up"])
            self.eng.infer_device(self.hw, self.res, slot["d_depth"], slot["d_mask"], slot["d_unc"], out=slot["d_out"])
            slot["done"].record(self.ctx.stream)
        with torch.cuda.stream(self.d2h):
            self.d2h.wait_event(slot["done"])
            slot["h_out"].copy_(slot["d_out"], non_blocking=True)
            slot["down"].record(self.d2h)
        slot["busy"], slot["tag"] = True, tag
        return ready

    def drain(self):
        for k in range(len(self.slots)):
            slot = self.slots[(self.i + k) % len(self.slots)]
            if slot["busy"]:
                self._recycle()
                yield self._result(slot)


class BathymetricPipeline:
    """Complete inference pipeline (reference ``BathymetricPipeline``, models/pipeline.py:36-382)."""

    def __init__(self, config: Config, vr_bag_mode: str = "resampled", tile_batch: int = 16):
        self.config = config
        self.vr_bag_mode = vr_bag_mode
        self.tile_batch = max(1, int(tile_batch))
        self.tile_manager = TileManager(tile_size=config.tile.tile_size, overlap=config.tile.overlap,
                                        min_valid_ratio=config.tile.min_valid_ratio)
        # like the reference (:64-67), include_self_loops is not forwarded
        self.graph_builder = GraphBuilder(connectivity=config.graph.connectivity,
                                          edge_features=config.graph.edge_features)
        self.model: Optional[BathymetricGNN] = None
        self._engine: Optional[TileBatchEngine] = None
        self.host_stitch = False      # True: numpy TileMerger on the host even on one GPU (the multi-rank path)
        if config.device != "cuda" or not torch.cuda.is_available():
            raise rt.BgnnError(f"config.device={config.device!r}: this pipeline runs on an MI355X only "
                               "(there is no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device())
        logger.info(f"Pipeline initialized, device: {self.device}")

    # ---- model -----------------------------------------------------------------------------
    def set_model(self, model: BathymetricGNN):
        self.model = model.to(self.device).eval()
        self._engine = TileBatchEngine(self.model, self.graph_builder, self.device,
                                       self.config.inference.auto_correct_threshold,
                                       self.config.inference.review_threshold)

    def load_model(self, model_path: Union[str, Path], trust_pickle: bool = False):
        """Checkpoint -> model (reference :92-132).  Accepts the trainer's dict
        (``model_state_dict`` + ``in_channels`` / ``edge_dim`` / optional ``model_config`` or
        pickled ``config``) and a plain ``{state_dict, meta}`` form.  Loaded with
        ``weights_only=True``.  A checkpoint that pickles the reference's ``Config`` object
        (``training/trainer.py:809-829``) cannot be read that way: unpickling executes code from the file, so it
        is an explicit opt-in (``trust_pickle=True``; the reference does it unconditionally, :105) and needs the
        reference's ``config`` package importable."""
        model_path = Path(model_path)
        if not model_path.exists():
            raise FileNotFoundError(f"Model not found: {model_path}")
        try:
            ckpt = torch.load(model_path, map_location="cpu", weights_only=True)
        except Exception as e:
            if not trust_pickle:
                raise RuntimeError(
                    f"{model_path} cannot be loaded with weights_only=True ({type(e).__name__}: it holds pickled Python "
                    "objects, e.g. the trainer's Config). Unpickling executes code from the file: pass "
                    "load_model(path, trust_pickle=True) only for a checkpoint you trust, or re-save it as "
                    "{'model_state_dict', 'in_channels', 'edge_dim', 'model_config': dict}") from e
            ckpt = torch.load(model_path, map_location="cpu", weights_only=False)
        mc = ckpt.get("model_config", None)
        if mc is None and ckpt.get("config", None) is not None:
            mc = getattr(ckpt["config"], "model", None)
        if mc is None:
            mc = self.config.model
        get = (lambda k, dflt=None: mc.get(k, dflt)) if isinstance(mc, dict) else (lambda k, dflt=None: getattr(mc, k, dflt))
        sd = ckpt.get("model_state_dict", ckpt.get("state_dict"))
        model = BathymetricGNN(
            in_channels=ckpt.get("in_channels", 7), hidden_channels=get("gnn_hidden_channels", 64),
            num_gnn_layers=get("gnn_num_layers", 4), gnn_type=get("gnn_type", "GAT"), heads=get("gnn_heads", 4),
            num_classes=get("num_classes", 3), predict_correction=get("predict_correction", True), dropout=0.0,
            edge_dim=ckpt.get("edge_dim", 3))
        model.load_state_dict(sd)
        self.set_model(model)
        logger.info(f"Model loaded from {model_path}")

    # ---- per tile ----------------------------------------------------------------------------
    def _tile_uncertainty(self, tile):
        # The reference passes tile.uncertainty unconditionally (:253), which makes x 8 columns wide
        # and fails for a 7-channel model; like NativeVRProcessor we drop the band when the model
        # does not take it.
        return tile.uncertainty if self.model.in_channels == 8 else None

    def _process_tiles(self, tiles, grid: BathymetricGrid) -> List[Dict[str, np.ndarray]]:
        if self.model is None:
            raise RuntimeError("Model not loaded. Call load_model() first.")
        res = self._engine.infer([t.data for t in tiles], [t.valid_mask for t in tiles],
                                 [self._tile_uncertainty(t) for t in tiles], [grid.resolution] * len(tiles))
        for r, t in zip(res, tiles):
            r["cleaned_depth"] = t.data       # original; corrections are applied after stitching (:310)
        return res

    def _process_tile(self, tile, grid: BathymetricGrid) -> Dict[str, np.ndarray]:
        """One tile -> {'cleaned_depth', 'classification', 'confidence', 'correction'} (reference :243-314)."""
        return self._process_tiles([tile], grid)[0]

    # ---- whole grid ----------------------------------------------------------------------------
    def process_survey_device(self, depth_t: torch.Tensor, valid_t: torch.Tensor, unc_t: Optional[torch.Tensor],
                              resolution, shard: Optional[Tuple[int, int]] = None,
                              survey_shape: Optional[Tuple[int, int]] = None, row_offset: int = 0):
        """Survey resident in HBM in, ``[4, H, W]`` float32 device tensor out (classification, confidence,
        correction, cleaned depth): tiles are cut (``bgnn_cut_tiles``), filtered by ``min_valid_ratio``
        (``bgnn_tile_valid_counts``), classified batch by batch into three long per-tile result arrays and
        stitched / post-processed by ``bgnn_stitch_tiles``.  Nothing crosses PCIe in between; sized for one
        GPU's 288 GB (a 60000 x 60000 survey @512/128 holds 14 GB of depth, 77 GB of per-tile results and
        58 GB of outputs).

        ``shard=(rank, world)`` (one process per GPU, ``torch.distributed`` initialised, every rank holding the
        survey): the rank classifies the tile rows ``survey_shard_plan`` gives it, receives the few earlier tile
        rows that reach into its band of survey rows (``exchange_halo_tile_rows``, the only inter-GPU traffic)
        and stitches that band; returns ``(row0, row1, [4, row1-row0, W])``.  Every cell sees the same tiles in
        the same ascending order as on one GPU, so the bands concatenate to the single-GPU result bit for bit.

        A rank does not need the whole survey: with ``survey_shape=(H, W)`` and ``row_offset``, ``depth_t`` / ``valid_t`` /
        ``unc_t`` hold only the survey rows ``[row_offset, row_offset + depth_t.shape[0])`` -- what the rank's own tile rows
        span (``survey_rows_of_rank``)."""
        if self.model is None:
            raise RuntimeError("Model not loaded. Call load_model() first.")
        eng, ctx, dev = self._engine, self._engine.ctx, self._engine.ctx.device
        tm = self.tile_manager
        LH, W = (int(v) for v in depth_t.shape)              # rows held locally
        H = int(survey_shape[0]) if survey_shape is not None else LH
        assert survey_shape is None or int(survey_shape[1]) == W
        row_offset = int(row_offset)
        assert 0 <= row_offset and row_offset + LH <= H
        assert depth_t.dtype == torch.float32 and depth_t.is_contiguous() and valid_t.is_contiguous() and valid_t.shape == depth_t.shape
        valid_u8 = valid_t.view(torch.uint8) if valid_t.dtype == torch.bool else valid_t
        ntr, ntc, specs = tm.compute_tile_grid((H, W))
        sa = np.array([[s.row_start, s.col_start, s.row_end, s.col_end] for s in specs], np.int64)
        th, tw = int(sa[0, 2] - sa[0, 0]), int(sa[0, 3] - sa[0, 1])     # every tile has this extent (shift-back rule)
        cells = th * tw
        assert np.all((sa[:, 2] - sa[:, 0]) * (sa[:, 3] - sa[:, 1]) == cells)
        rs = sa[::ntc, 0].copy(); re = sa[::ntc, 2].copy(); cs = sa[:ntc, 1].copy(); ce = sa[:ntc, 3].copy()
        rank, world = shard if shard is not None else (0, 1)
        plan = survey_shard_plan(rs, re, H, world)
        me = plan[rank]
        ta, tb = me["tile_rows"]; R0, R1 = me["cell_rows"]
        own = np.arange(ta * ntc, tb * ntc)                              # my tiles, ascending spec order
        if len(own):
            assert rs[ta] >= row_offset and re[tb - 1] <= row_offset + LH, "the local rows do not cover this rank's tile rows"
        assert R0 == R1 or (R0 >= row_offset and R1 <= row_offset + LH)
        # ---- min_valid_ratio filter (iterate_tiles, tiling.py:203-209), exact integer counts ----
        keep = np.zeros(0, bool)
        if len(own):
            org = np.ascontiguousarray(sa[own, :2], dtype=np.int32)
            org[:, 0] -= row_offset                                      # origins in local rows
            org_t = torch.from_numpy(org).to(dev)
            cnt_t = torch.empty(len(own), dtype=torch.int64, device=dev)
            ctx.begin()
            rt.check(ctx.lib.bgnn_tile_valid_counts(ctx.handle, LH, W, rt.ptr(valid_u8), len(own), rt.ptr(org_t), th, tw, rt.ptr(cnt_t)))
            ctx.end()
            keep = ~((cnt_t.cpu().numpy() / cells) < tm.min_valid_ratio)    # same float64 test as iterate_tiles
        proc = np.nonzero(keep)[0]                                       # positions in `own`
        n_proc = len(proc)
        halo_rows = [t for _, t in me["need"]]
        n_slots = n_proc + len(halo_rows) * ntc
        total = max(n_slots, 1) * cells
        r_cls = torch.empty(total, dtype=torch.float32, device=dev)
        r_conf = torch.empty(total, dtype=torch.float32, device=dev)
        r_corr = torch.empty(total, dtype=torch.float32, device=dev)
        resol = np.array([[float(resolution[0]), float(resolution[1])]], np.float64)
        if n_proc:
            nbmax = min(self.tile_batch, n_proc)
            d_b = torch.empty(nbmax * cells, dtype=torch.float32, device=dev)
            m_b = torch.empty(nbmax * cells, dtype=torch.uint8, device=dev)
            u_b = torch.empty(nbmax * cells, dtype=torch.float32, device=dev) if unc_t is not None else None
            proc_t = torch.from_numpy(proc).to(dev)
            for b0 in range(0, n_proc, self.tile_batch):
                nb = min(self.tile_batch, n_proc - b0)
                o_b = org_t[proc_t[b0:b0 + nb]].contiguous()
                ctx.begin()
                rt.check(ctx.lib.bgnn_cut_tiles(ctx.handle, LH, W, rt.ptr(depth_t), rt.ptr(valid_u8), rt.ptr(unc_t), nb, rt.ptr(o_b),
                                                th, tw, rt.ptr(d_b), rt.ptr(m_b), rt.ptr(u_b)))
                ctx.end()
                lo, hi, n = b0 * cells, (b0 + nb) * cells, nb * cells
                eng.infer_device(np.tile(np.array([[th, tw]], np.int32), (nb, 1)), np.tile(resol, (nb, 1)), d_b[:n], m_b[:n],
                                 u_b[:n] if u_b is not None else None, out=(r_cls[lo:hi], r_conf[lo:hi], r_corr[lo:hi]))
            del d_b, m_b, u_b
        # ---- tile offsets of my own tiles, by (tile row, tile col) ----
        off_own = np.full(len(own), -1, np.int64)
        off_own[proc] = np.arange(n_proc, dtype=np.int64) * cells
        # ---- the only inter-GPU step: earlier tile rows that reach into my band ----
        off_halo = np.full((len(halo_rows), ntc), -1, np.int64)
        if world > 1:
            blk = 3 * ntc * cells
            for p in plan:
                p["_halo_shape"], p["_halo_dtype"], p["_halo_device"] = (blk + ntc,), torch.float32, dev
            wanted = sorted({t for p in plan for src, t in p["need"] if src == rank})
            local = {}
            for t in wanted:                                             # pack [3][ntc][cells] + keep flags
                k0 = (t - ta) * ntc
                kept = np.nonzero(keep[k0:k0 + ntc])[0]
                buf = torch.zeros(blk + ntc, dtype=torch.float32, device=dev)
                if len(kept):
                    cols = torch.from_numpy(kept).to(dev)
                    p0 = int(off_own[k0 + kept[0]]); p1 = p0 + len(kept) * cells     # kept tiles of a row are consecutive slots
                    for c, r in enumerate((r_cls, r_conf, r_corr)):
                        buf[c * ntc * cells:(c + 1) * ntc * cells].view(ntc, cells)[cols] = r[p0:p1].view(len(kept), cells)
                    buf[blk + cols] = 1.0
                local[t] = buf
            torch.cuda.synchronize(dev)
            got = exchange_halo_tile_rows(plan, rank, local)
            for j, t in enumerate(halo_rows):
                buf = got[t]
                base = (n_proc + j * ntc) * cells
                for c, r in enumerate((r_cls, r_conf, r_corr)):
                    r[base:base + ntc * cells] = buf[c * ntc * cells:(c + 1) * ntc * cells]
                kp = buf[blk:].cpu().numpy() > 0
                off_halo[j, kp] = base + np.nonzero(kp)[0].astype(np.int64) * cells
        # ---- stitch my band: tile rows involved = halo rows (earlier, ascending) then my own ----
        rows_inv = np.array(halo_rows + list(range(ta, tb)), np.int64)
        o = torch.empty((4, R1 - R0, W), dtype=torch.float32, device=dev)
        if R1 > R0:
            rs_b = (rs[rows_inv] - R0).astype(np.int32); re_b = (re[rows_inv] - R0).astype(np.int32)
            pitch = max(th, tw)
            roww = np.zeros((len(rows_inv), pitch), np.float32); colw = np.zeros((ntc, pitch), np.float32)
            for i, t in enumerate(rows_inv):                             # numpy, exactly TileManager._create_1d_blend
                roww[i, :re[t] - rs[t]] = tm._create_1d_blend(int(re[t] - rs[t]))
            for j in range(ntc):
                colw[j, :ce[j] - cs[j]] = tm._create_1d_blend(int(ce[j] - cs[j]))
            tile_off = np.concatenate([off_halo.reshape(-1), off_own])
            dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            rs_t, re_t, cs_t, ce_t = dv(rs_b), dv(re_b), dv(cs.astype(np.int32)), dv(ce.astype(np.int32))
            rw_t, cw_t, off_t = dv(roww), dv(colw), dv(tile_off)
            ctx.begin()
            rt.check(ctx.lib.bgnn_stitch_tiles(
                ctx.handle, R1 - R0, W, len(rows_inv), ntc, rt.ptr(rs_t), rt.ptr(re_t), rt.ptr(cs_t), rt.ptr(ce_t), rt.ptr(rw_t),
                rt.ptr(cw_t), pitch, rt.ptr(off_t), rt.ptr(r_cls), rt.ptr(r_conf), rt.ptr(r_corr),
                rt.ptr(depth_t[R0 - row_offset:R1 - row_offset]), rt.ptr(valid_u8[R0 - row_offset:R1 - row_offset]),
                C.c_float(self.config.inference.auto_correct_threshold),
                rt.ptr(o[0]), rt.ptr(o[1]), rt.ptr(o[2]), rt.ptr(o[3])))
            ctx.end()
        self.last_tile_counts = (n_proc, len(own) - n_proc)
        logger.info(f"Processed {n_proc} tiles ({len(own) - n_proc} skipped below min_valid_ratio)")
        return o if shard is None else (R0, R1, o)

    def survey_rows_of_rank(self, shape, rank: int, world: int):
        """Survey rows ``[lo, hi)`` rank ``rank`` of ``world`` must hold for a survey of ``shape``: the span of its own tile
        rows (its band of cell rows lies inside it).  (0, 0) for a rank without tile rows."""
        ntr, ntc, specs = self.tile_manager.compute_tile_grid(tuple(int(v) for v in shape))
        rs = np.array([specs[i * ntc].row_start for i in range(ntr)], np.int64)
        re = np.array([specs[i * ntc].row_end for i in range(ntr)], np.int64)
        plan = survey_shard_plan(rs, re, int(shape[0]), world)
        ta, tb = plan[rank]["tile_rows"]
        return (int(rs[ta]), int(re[tb - 1])) if tb > ta else (0, 0), plan

    # ---- host grid in, host grids out, streamed (one GPU) -----------------------------------------------------------
    STREAM_MIN_CELLS = 1 << 24        # surveys from 16 M cells on take the streamed form (below: one upload, one download)
    STREAM_BAND_TILE_ROWS = 2         # tile rows classified between two band stitches
    STREAM_UPLOAD_ROWS_BYTES = 128 << 20

    def process_grid_streamed(self, grid: BathymetricGrid, band_tile_rows: Optional[int] = None) -> Dict[str, np.ndarray]:
        """``process_grid_device`` for a survey that lives in HOST memory, with both PCIe crossings and the host-side copies under the
        kernels (reference :163-211 is host arrays in, host arrays out).  Same tiles, same kernels, same stitch as
        ``process_survey_device`` -- the results are bit-identical to it -- in three overlapped roles:

        * an UPLOAD thread copies the survey into pinned slabs and H2D's them chunk by chunk on its own stream; the valid mask is
          made on the device from the depth (``BathymetricGrid.valid_mask``: finite and not the nodata value; a foreign grid
          object's own mask is uploaded instead), and as soon as a tile row's rows are resident its tiles' valid counts are taken
          (``bgnn_tile_valid_counts``) -- the ``min_valid_ratio`` filter never makes the classifying stream wait for the host;
        * THIS thread cuts and classifies the kept tiles, tile row by tile row, into a ring of per-tile result slots that holds
          just the tile rows a stitch can still reach; after every ``band_tile_rows`` tile rows the survey rows NO later tile
          touches are final: they are stitched (``bgnn_stitch_tiles`` on a row band -- the stitch is a per-cell gather, so a band is
          the same arithmetic as the whole survey) together with the valid mask as float32, and queued for download;
        * a DOWNLOAD thread waits for a band's D2H (own stream, pinned slabs) and copies it into the five result arrays -- fresh
          pageable memory, whose first touch costs as much as the copy itself: that runs beside the kernels too."""
        import threading
        if self.model is None:
            raise RuntimeError("Model not loaded. Call load_model() first.")
        eng, ctx, dev = self._engine, self._engine.ctx, self._engine.ctx.device
        tm = self.tile_manager
        H, W = (int(v) for v in grid.shape)
        depth_h = np.ascontiguousarray(grid.depth, dtype=np.float32)
        plain_mask = type(grid) is BathymetricGrid
        valid_h = None if plain_mask else np.ascontiguousarray(grid.valid_mask).view(np.uint8)
        nodata = grid.nodata_value if plain_mask else None
        use_unc = self.model.in_channels == 8 and grid.uncertainty is not None
        unc_h = np.ascontiguousarray(grid.uncertainty, dtype=np.float32) if use_unc else None
        ntr, ntc, specs = tm.compute_tile_grid((H, W))
        sa = np.array([[s.row_start, s.col_start, s.row_end, s.col_end] for s in specs], np.int64)
        th, tw = int(sa[0, 2] - sa[0, 0]), int(sa[0, 3] - sa[0, 1])
        cells = th * tw
        assert np.all((sa[:, 2] - sa[:, 0]) * (sa[:, 3] - sa[:, 1]) == cells)
        rs = sa[::ntc, 0].copy(); re = sa[::ntc, 2].copy(); cs = sa[:ntc, 1].copy(); ce = sa[:ntc, 3].copy()
        nb = max(1, int(band_tile_rows or self.STREAM_BAND_TILE_ROWS))
        # tile rows a later band's stitch still reads: those that end past the first row of the band's first tile row
        reach = max(int(np.count_nonzero(re[:t] > rs[t])) for t in range(ntr))
        K = min(ntr, nb + reach)                                  # ring slots (tile rows)
        bands = [(a, min(a + nb, ntr)) for a in range(0, ntr, nb)]
        band_rows = []                                            # survey rows that become final with each band
        R0 = 0
        for a, b in bands:
            R1 = int(rs[b]) if b < ntr else H
            R1 = max(R1, R0)
            band_rows.append((R0, R1)); R0 = R1
        max_rows = max(r1 - r0 for r0, r1 in band_rows)
        # ---- device / pinned buffers ----
        mk = lambda dt, *shape: torch.empty(shape, dtype=dt, device=dev)
        pins = self.__dict__.setdefault("_stream_pins", {})      # pinned slabs are kept between calls (page-locking ~1 GB costs ~0.3 s)
        sig = (H, W, th, tw, nb, use_unc, plain_mask)
        if self.__dict__.get("_stream_pins_sig") != sig:          # ... for surveys of the same geometry; another one starts afresh
            pins.clear()
            self.__dict__["_stream_pins_sig"] = sig

        def pin(dt, *shape):
            lst = pins.setdefault((dt, shape), [])
            i = taken[(dt, shape)] = taken.get((dt, shape), -1) + 1
            while len(lst) <= i:
                lst.append(torch.empty(shape, dtype=dt, pin_memory=True))
            return lst[i]
        taken: Dict = {}
        depth_t, valid_t = mk(torch.float32, H, W), mk(torch.uint8, H, W)
        unc_t = mk(torch.float32, H, W) if use_unc else None
        r_all = mk(torch.float32, 3, K * ntc * cells)             # ring: [channel][slot][tile][cells]
        o_dev = [mk(torch.float32, 5, max_rows, W) for _ in range(2)]
        o_pin = [pin(torch.float32, 5, max_rows, W) for _ in range(2)]
        up_rows = max(1, min(H, self.STREAM_UPLOAD_ROWS_BYTES // (4 * W)))
        n_up = 2 + (2 if use_unc else 0) + (2 if valid_h is not None else 0)
        up_pin = [pin(torch.float32, up_rows, W) for _ in range(2)]
        up_pin_u = [pin(torch.float32, up_rows, W) for _ in range(2)] if use_unc else None
        up_pin_m = [pin(torch.uint8, up_rows, W) for _ in range(2)] if valid_h is not None else None
        torch.cuda.synchronize(dev)                               # (buffers made on this stream, used on three others)
        # (the upload thread has a library context of its own -- calls on one context are not re-entrant --; its stream is the upload stream)
        up_ctx = self.__dict__.get("_upload_ctx")
        if up_ctx is None or up_ctx.handle is None:
            up_ctx = self.__dict__["_upload_ctx"] = rt.new_context(dev)
        up_stream, down_stream = up_ctx.stream, torch.cuda.Stream(dev)
        out = {k: np.empty((H, W), np.float32) for k in ("classification", "confidence", "correction", "cleaned_depth", "valid_mask")}
        order = ("classification", "confidence", "correction", "cleaned_depth", "valid_mask")
        # ---- tile-row state published by the upload thread ----
        cv = threading.Condition()
        keep_rows: Dict[int, np.ndarray] = {}
        row_ready = [torch.cuda.Event() for _ in range(ntr)]     # tile row t's survey rows (and mask) are resident
        errors: List[BaseException] = []
        org_all = np.ascontiguousarray(sa[:, :2], dtype=np.int32)
        org_t = torch.from_numpy(org_all).to(dev)

        def uploader():
            try:
                done_rows, next_t, slot_ev = 0, 0, [None, None]
                with torch.cuda.stream(up_stream):
                    k = 0
                    while done_rows < H:
                        r0, r1 = done_rows, min(H, done_rows + up_rows)
                        j = k % 2
                        if slot_ev[j] is not None:
                            slot_ev[j].synchronize()               # the slab's previous H2D has left it
                        n = r1 - r0
                        np.copyto(up_pin[j].numpy()[:n], depth_h[r0:r1])
                        depth_t[r0:r1].copy_(up_pin[j][:n], non_blocking=True)
                        if use_unc:
                            np.copyto(up_pin_u[j].numpy()[:n], unc_h[r0:r1])
                            unc_t[r0:r1].copy_(up_pin_u[j][:n], non_blocking=True)
                        if valid_h is not None:
                            np.copyto(up_pin_m[j].numpy()[:n], valid_h[r0:r1])
                            valid_t[r0:r1].copy_(up_pin_m[j][:n], non_blocking=True)
                        else:                                      # BathymetricGrid.valid_mask on the device
                            d = depth_t[r0:r1]
                            m = torch.isfinite(d)
                            if nodata is not None and not np.isnan(nodata):
                                m &= d != nodata
                            valid_t[r0:r1].copy_(m)
                        slot_ev[j] = torch.cuda.Event(); slot_ev[j].record(up_stream)
                        done_rows, k = r1, k + 1
                        # tile rows whose rows are now all resident: valid counts -> min_valid_ratio filter (iterate_tiles, tiling.py:203-209)
                        t0 = next_t
                        while next_t < ntr and re[next_t] <= done_rows:
                            next_t += 1
                        if next_t > t0:
                            n_t = (next_t - t0) * ntc
                            cnt_t = torch.empty(n_t, dtype=torch.int64, device=dev)
                            rt.check(up_ctx.lib.bgnn_tile_valid_counts(up_ctx.handle, H, W, rt.ptr(valid_t), n_t,
                                                                       rt.ptr(org_t[t0 * ntc:next_t * ntc]), th, tw, rt.ptr(cnt_t)))
                            for t in range(t0, next_t):
                                row_ready[t].record(up_stream)
                            keep = ~((cnt_t.cpu().numpy() / cells) < tm.min_valid_ratio)      # (syncs the upload stream only)
                            with cv:
                                for t in range(t0, next_t):
                                    keep_rows[t] = keep[(t - t0) * ntc:(t - t0 + 1) * ntc]
                                cv.notify_all()
            except BaseException as e:                               # noqa: BLE001 -- handed to the caller's thread
                with cv:
                    errors.append(e); cv.notify_all()

        import queue
        down_q: "queue.Queue" = queue.Queue()
        slot_free = [threading.Event(), threading.Event()]
        for e in slot_free:
            e.set()

        def downloader():
            try:
                while True:
                    item = down_q.get()
                    if item is None:
                        return
                    j, r0, r1, ev = item
                    ev.synchronize()
                    src = o_pin[j].numpy()
                    for c, name in enumerate(order):
                        np.copyto(out[name][r0:r1], src[c, :r1 - r0])
                    slot_free[j].set()
            except BaseException as e:                               # noqa: BLE001
                with cv:
                    errors.append(e); cv.notify_all()
                for e2 in slot_free:
                    e2.set()

        t_up = threading.Thread(target=uploader, name="bgnn-survey-up", daemon=True)
        t_dn = threading.Thread(target=downloader, name="bgnn-survey-down", daemon=True)
        t_up.start(); t_dn.start()
        resol = np.array([[float(grid.resolution[0]), float(grid.resolution[1])]], np.float64)
        pitch = max(th, tw)
        colw = np.zeros((ntc, pitch), np.float32)
        for j in range(ntc):
            colw[j, :ce[j] - cs[j]] = tm._create_1d_blend(int(ce[j] - cs[j]))
        dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        cs_t, ce_t, cw_t = dv(cs.astype(np.int32)), dv(ce.astype(np.int32)), dv(colw)
        tile_off = np.full((ntr, ntc), -1, np.int64)
        d_b = mk(torch.float32, min(self.tile_batch, ntc) * cells); m_b = mk(torch.uint8, min(self.tile_batch, ntc) * cells)
        u_b = mk(torch.float32, min(self.tile_batch, ntc) * cells) if use_unc else None
        n_proc = 0
        band_done = [None, None]                                  # D2H events of the band that last used each device / pinned slot
        try:
            for bi, ((ta, tb), (R0, R1)) in enumerate(zip(bands, band_rows)):
                with cv:
                    while not errors and any(t not in keep_rows for t in range(ta, tb)):
                        cv.wait(0.05)
                    if errors:
                        raise errors[0]
                for t in range(ta, tb):
                    kept = np.nonzero(keep_rows[t])[0]
                    slot = t % K
                    tile_off[t, :] = -1
                    tile_off[t, kept] = (slot * ntc + np.arange(len(kept), dtype=np.int64)) * cells
                    if not len(kept):
                        continue
                    n_proc += len(kept)
                    ctx.stream.wait_event(row_ready[t])
                    idx_t = torch.from_numpy(t * ntc + kept).to(dev)
                    for b0 in range(0, len(kept), self.tile_batch):
                        nbt = min(self.tile_batch, len(kept) - b0)
                        o_b = org_t[idx_t[b0:b0 + nbt]].contiguous()
                        ctx.begin()
                        rt.check(ctx.lib.bgnn_cut_tiles(ctx.handle, H, W, rt.ptr(depth_t), rt.ptr(valid_t), rt.ptr(unc_t), nbt, rt.ptr(o_b),
                                                        th, tw, rt.ptr(d_b), rt.ptr(m_b), rt.ptr(u_b)))
                        ctx.end()
                        lo = (slot * ntc + b0) * cells; hi = lo + nbt * cells; n = nbt * cells
                        eng.infer_device(np.tile(np.array([[th, tw]], np.int32), (nbt, 1)), np.tile(resol, (nbt, 1)), d_b[:n], m_b[:n],
                                         u_b[:n] if u_b is not None else None, out=(r_all[0, lo:hi], r_all[1, lo:hi], r_all[2, lo:hi]))
                if R1 <= R0:
                    continue
                # ---- stitch the rows that are final now, with the tile rows that reach into them ----
                inv = np.nonzero((rs < R1) & (re > R0))[0]
                assert inv.max() < tb and inv.min() > tb - 1 - K, "a stitch only reads tile rows still in the ring"
                roww = np.zeros((len(inv), pitch), np.float32)
                for i, t in enumerate(inv):
                    roww[i, :re[t] - rs[t]] = tm._create_1d_blend(int(re[t] - rs[t]))
                j = bi % 2
                if band_done[j] is not None:
                    torch.cuda.current_stream(dev).wait_event(band_done[j])      # the device slab's last download has left it
                rs_t, re_t = dv((rs[inv] - R0).astype(np.int32)), dv((re[inv] - R0).astype(np.int32))
                rw_t, off_t = dv(roww), dv(tile_off[inv].reshape(-1))
                o = o_dev[j]
                n_r = R1 - R0
                ctx.begin()
                rt.check(ctx.lib.bgnn_stitch_tiles(
                    ctx.handle, n_r, W, len(inv), ntc, rt.ptr(rs_t), rt.ptr(re_t), rt.ptr(cs_t), rt.ptr(ce_t), rt.ptr(rw_t), rt.ptr(cw_t),
                    pitch, rt.ptr(off_t), rt.ptr(r_all[0]), rt.ptr(r_all[1]), rt.ptr(r_all[2]), rt.ptr(depth_t[R0:R1]), rt.ptr(valid_t[R0:R1]),
                    C.c_float(self.config.inference.auto_correct_threshold),
                    rt.ptr(o[0]), rt.ptr(o[1]), rt.ptr(o[2]), rt.ptr(o[3])))
                ctx.end()
                # (the four stitched grids are [n_r][W] blocks at the head of their [max_rows][W] planes; the mask as float32 beside them)
                o[4, :n_r].copy_(valid_t[R0:R1])
                stitched = torch.cuda.Event(); stitched.record(torch.cuda.current_stream(dev))
                slot_free[j].wait()                                # the pinned slab's last band has been copied out
                with cv:
                    if errors:
                        raise errors[0]
                slot_free[j].clear()
                with torch.cuda.stream(down_stream):
                    down_stream.wait_event(stitched)
                    for c in range(5):                             # (contiguous blocks: a strided copy would go through a host-side temporary)
                        o_pin[j][c, :n_r].copy_(o[c, :n_r], non_blocking=True)
                    ev = torch.cuda.Event(); ev.record(down_stream)
                band_done[j] = ev
                down_q.put((j, R0, R1, ev))
        finally:
            down_q.put(None)
            t_up.join(); t_dn.join()
            torch.cuda.synchronize(dev)
        if errors:
            raise errors[0]
        self.last_tile_counts = (n_proc, ntr * ntc - n_proc)
        logger.info(f"Processed {n_proc} tiles ({ntr * ntc - n_proc} skipped below min_valid_ratio)")
        return out

    def process_grid_device(self, grid: BathymetricGrid) -> Optional[Dict[str, np.ndarray]]:
        """Survey path with everything between the two PCIe crossings on the device: the survey is uploaded once,
        processed by ``process_survey_device`` and the four result grids come back in one copy.  Same results as the host
        merge (``host_stitch = True``), bit for bit, for any number of GPUs.

        Under an initialised ``torch.distributed`` job the survey is row-band sharded: every rank uploads ONLY the rows its
        own tile rows span (not the whole survey), classifies and stitches its band, and the bands go to rank 0 as tensors
        (``gather_bands_to_rank0``).  Rank 0 returns the result dict; the other ranks return None."""
        if self.model is None:
            raise RuntimeError("Model not loaded. Call load_model() first.")
        dev = self._engine.ctx.device
        valid_np = grid.valid_mask
        use_unc = self.model.in_channels == 8 and grid.uncertainty is not None
        up = lambda a, lo, hi, dt: torch.from_numpy(np.ascontiguousarray(a[lo:hi], dtype=dt)).to(dev)
        rank, world = shard_info()
        H, W = (int(v) for v in grid.shape)
        if world == 1 and H * W >= self.STREAM_MIN_CELLS:
            return self.process_grid_streamed(grid)
        if world == 1:
            depth_t = up(grid.depth, 0, H, np.float32)
            valid_t = torch.from_numpy(np.ascontiguousarray(valid_np).view(np.uint8)).to(dev)
            unc_t = up(grid.uncertainty, 0, H, np.float32) if use_unc else None
            host = self.process_survey_device(depth_t, valid_t, unc_t, grid.resolution).cpu().numpy()
        else:               # row bands: each rank holds and stitches its own rows; the bands are gathered once at the end
            (lo, hi), plan = self.survey_rows_of_rank((H, W), rank, world)
            band = None
            if hi > lo:
                depth_t = up(grid.depth, lo, hi, np.float32)
                valid_t = torch.from_numpy(np.ascontiguousarray(valid_np[lo:hi]).view(np.uint8)).to(dev)
                unc_t = up(grid.uncertainty, lo, hi, np.float32) if use_unc else None
                _, _, band = self.process_survey_device(depth_t, valid_t, unc_t, grid.resolution, shard=(rank, world),
                                                        survey_shape=(H, W), row_offset=lo)
            host = gather_bands_to_rank0(plan, rank, band, W, device=dev)
            if host is None:
                return None
        return {"cleaned_depth": host[3], "classification": host[0], "confidence": host[1], "correction": host[2],
                "valid_mask": valid_np.astype(np.float32)}

    def process_grid(self, grid: BathymetricGrid) -> Dict[str, np.ndarray]:
        """The body of ``process`` between load and save (reference :163-211), tiles batched.

        Default: the device path (``process_grid_device``), on one GPU or row-band sharded over the ranks of an
        initialised ``torch.distributed`` job.  With ``host_stitch = True`` (or without a device engine) the
        reference-shaped host merge below runs instead: the tiles that pass the ``min_valid_ratio`` filter are
        dealt round-robin by index to the ranks, each rank classifies its share, the per-tile grids are
        exchanged once (``exchange_tile_results``: tensors, no pickling; tiles are independent, so there is no collective inside the
        data path) and EVERY rank stitches them in ascending spec order -- the float32 blend sums and the
        ``>`` tie rule of the discrete channel therefore do not depend on the number of GPUs."""
        if self.model is None:
            raise RuntimeError("Model not loaded. Call load_model() first.")
        if getattr(self, "_engine", None) is not None and not getattr(self, "host_stitch", False):
            return self.process_grid_device(grid)      # one GPU, or row bands over the ranks' GPUs
        _, _, specs = self.tile_manager.compute_tile_grid(grid.shape)
        by_pos = {(s.tile_row, s.tile_col): s for s in specs}
        tiles = list(self.tile_manager.iterate_tiles(grid, skip_empty=True))
        rank, world = shard_info()
        mine = list(range(rank, len(tiles), world))
        done = {}
        for i0 in range(0, len(mine), self.tile_batch):
            idx = mine[i0:i0 + self.tile_batch]
            for i, r in zip(idx, self._process_tiles([tiles[i] for i in idx], grid)):
                done[i] = r
        done = exchange_tile_results(done)
        merger = TileMerger(self.tile_manager)
        merger.initialize(grid_shape=grid.shape, channels=["cleaned_depth", "classification", "confidence", "correction"])
        for i in range(len(tiles)):                  # ascending spec order
            t = tiles[i]
            merger.add_tile(by_pos[(t.tile_row, t.tile_col)], done[i])
        num_tiles = len(tiles)
        logger.info(f"Processed {num_tiles} tiles ({len(specs) - num_tiles} skipped below min_valid_ratio)")
        results = merger.finalize()
        valid_mask = grid.valid_mask
        results["valid_mask"] = valid_mask.astype(np.float32)
        unprocessed = valid_mask & np.isnan(results["classification"])
        if np.any(unprocessed):                      # :198-207
            logger.info(f"Preserving {int(np.sum(unprocessed)):,} valid cells from unprocessed tiles")
            results["cleaned_depth"][unprocessed] = grid.depth[unprocessed]
            for ch in ("classification", "confidence", "correction"):
                results[ch][unprocessed] = 0.0
        results["cleaned_depth"] = self._apply_corrections(grid, results)
        return results

    def process(self, input_path, output_path, export_extras: bool = True) -> Dict[str, np.ndarray]:
        """File in, file out (reference :134-241).  Reading / writing BAG / GeoTIFF is the
        reference's GDAL code and is not part of this package."""
        if self.model is None:
            raise RuntimeError("Model not loaded. Call load_model() first.")
        raise ImportError("GDAL is required to read/write survey files; load the grid with the reference's "
                          "BathymetricLoader and call process_grid(grid) instead")

    def _apply_corrections(self, grid: BathymetricGrid, results: Dict[str, np.ndarray]) -> np.ndarray:
        """cleaned = depth - correction where class == NOISE, confidence > threshold, valid (:316-349)."""
        cleaned = grid.depth.copy()
        apply = ((results["classification"] == BathymetricGNN.CLASS_NOISE)
                 & (results["confidence"] > self.config.inference.auto_correct_threshold) & grid.valid_mask)
        cleaned[apply] = grid.depth[apply] - results["correction"][apply]
        nvalid = max(int(np.sum(grid.valid_mask)), 1)
        logger.info(f"Applied corrections to {int(np.sum(apply))} cells ({100 * np.sum(apply) / nvalid:.1f}% of valid)")
        return cleaned
