"""Multi-GPU path, rehearsed on CPU: world_size-2 gloo job.  Tiles are dealt round-robin to the ranks,
per-tile results exchanged once, and every rank stitches in ascending spec order -- so the stitched
survey must be BITWISE identical to the single-process result (no data-path collective exists to get
wrong; what is tested is the dealing, the exchange and the order-independence of the merge)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_process_tiles(self, tiles, grid):
    """CPU stand-in for the GPU engine: any deterministic per-tile function will do here."""
    out = []
    for t in tiles:
        d = np.where(t.valid_mask, t.data, 0.0).astype(np.float32)
        out.append({"cleaned_depth": t.data,
                    "classification": np.where(t.valid_mask, np.floor(np.abs(d) * 3) % 3, 0).astype(np.float32),
                    "confidence": np.where(t.valid_mask, np.abs(d * 7) % 1.0, 0).astype(np.float32),
                    "correction": np.where(t.valid_mask, 0.1 * np.sin(d), 0).astype(np.float32)})
    return out


def _make_pipeline_and_grid():
    from bathymetric_gnn_amd.config import Config
    from bathymetric_gnn_amd.data import BathymetricGrid, TileManager
    from bathymetric_gnn_amd.models import pipeline as pl
    from bathymetric_gnn_amd import synthetic
    cfg = Config()
    cfg.tile.tile_size, cfg.tile.overlap, cfg.tile.min_valid_ratio = 64, 16, 0.3
    p = pl.BathymetricPipeline.__new__(pl.BathymetricPipeline)      # (the real ctor insists on a GPU)
    p.config, p.tile_batch, p.model = cfg, 3, object()
    p.tile_manager = TileManager(64, 16, 0.3)
    pl.BathymetricPipeline._process_tiles = _stub_process_tiles
    d, m, _ = synthetic.synthetic_tile(200, 170, 5, "V1")
    d[:70, :80] = 1.0e6                                             # a corner of skipped tiles
    return p, BathymetricGrid(depth=d, nodata_value=1.0e6, resolution=(0.5, 0.5))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p, grid = _make_pipeline_and_grid()
    from bathymetric_gnn_amd.models import shard_info
    assert shard_info() == (rank, world)
    res = p.process_grid(grid)
    q.put((rank, {k: v.copy() for k, v in res.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_stitch_equals_single_process():
    sys.path.insert(0, ROOT)
    p, grid = _make_pipeline_and_grid()
    single = p.process_grid(grid)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for rank in (0, 1):
        for k, v in single.items():
            a, b = got[rank][k], v
            assert np.array_equal(np.isnan(a), np.isnan(b)), (rank, k)
            assert np.array_equal(np.nan_to_num(a), np.nan_to_num(b)), (rank, k)
    # sanity of the stitched result itself
    vm = grid.valid_mask
    assert not np.isnan(single["classification"][vm]).any()
    assert set(np.unique(single["classification"][vm])) <= {0.0, 1.0, 2.0}


def _exchange_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bathymetric_gnn_amd.models.pipeline import exchange_tile_results
    mine = {}
    if rank == 0:          # rank 1 holds NO tile (world larger than the number of kept tiles)
        mine = {5: {"cleaned_depth": np.full((4, 6), 4.0, np.float32), "classification": np.full((4, 6), 1.0, np.float32),
                    "confidence": np.full((4, 6), 0.25, np.float32), "correction": np.full((4, 6), -0.5, np.float32)}}
    got = exchange_tile_results(mine)
    q.put((rank, {i: {k: v.copy() for k, v in r.items()} for i, r in got.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_channel_order_on_a_rank_without_tiles():
    """a rank that holds no tile used to unpack the all-gathered block in another channel order than the
    ranks that packed it (classification and cleaned_depth came back swapped)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    want = {"cleaned_depth": 4.0, "classification": 1.0, "confidence": 0.25, "correction": -0.5}
    for rank in (0, 1):
        assert list(got[rank].keys()) == [5]
        for k, v in want.items():
            assert np.all(got[rank][5][k] == v), (rank, k, got[rank][5][k][0, 0])


def _exchange_bad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bathymetric_gnn_amd.models.pipeline import exchange_tile_results
    mine = {}
    if rank == 0:          # a non-standard channel next to a rank that holds no tile: rank 1 could not name it
        mine = {3: {"classification": np.zeros((4, 6), np.float32), "my_extra_layer": np.ones((4, 6), np.float32)}}
    try:
        exchange_tile_results(mine)
        q.put((rank, "returned"))
    except ValueError as e:
        q.put((rank, "ValueError: " + str(e)[:60]))
    dist.barrier()               # both ranks get here: nobody is left inside a collective
    dist.destroy_process_group()


def test_exchange_refuses_unknown_channels_on_every_rank_together():
    """the channel check used to fire only on the rank WITHOUT tiles, after the metadata all_gather -- the other
    ranks went on into the block all_gather and hung.  Now every rank raises from the same metadata, before the next collective."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_exchange_bad_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert got[0].startswith("ValueError") and got[1].startswith("ValueError"), got
