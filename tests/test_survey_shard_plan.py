"""Row-band sharding of a tiled survey (BASELINE config 5): the ownership plan is pure host logic, and
the halo exchange is rehearsed with world_size-3 gloo on CPU tensors (the GPU run of the full path is
tools/survey_sharded_check.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _rows(H, tile, overlap):
    from bathymetric_gnn_amd.data import TileManager
    ntr, ntc, specs = TileManager(tile, overlap).compute_tile_grid((H, tile))
    rs = np.array([specs[i * ntc].row_start for i in range(ntr)]); re = np.array([specs[i * ntc].row_end for i in range(ntr)])
    return rs, re


@pytest.mark.parametrize("H,tile,overlap,world", [(60000, 512, 128, 8), (1300, 512, 128, 2), (1300, 512, 128, 8), (500, 64, 16, 3),
                                                  (1000, 512, 128, 3), (200, 512, 128, 4), (6000, 512, 128, 5)])
def test_plan_partitions_and_needs(H, tile, overlap, world):
    from bathymetric_gnn_amd.models.pipeline import survey_shard_plan
    rs, re = _rows(H, tile, overlap)
    plan = survey_shard_plan(rs, re, H, world)
    assert len(plan) == world
    # tile rows: contiguous blocks covering 0..ntr; cell rows: bands covering [0, H) without gaps or overlap
    rows = [t for p in plan for t in range(*p["tile_rows"])]
    assert rows == list(range(len(rs)))
    pos = 0
    for p in plan:
        a, b = p["cell_rows"]
        if p["tile_rows"][0] == p["tile_rows"][1]:
            assert a == b
            continue
        assert a == pos and b > a
        pos = b
    assert pos == H
    for k, p in enumerate(plan):
        R0, R1 = p["cell_rows"]
        if R0 == R1:
            assert p["need"] == []
            continue
        a, b = p["tile_rows"]
        covering = [t for t in range(len(rs)) if rs[t] < R1 and re[t] > R0]      # tile rows touching the band
        have = sorted([t for _, t in p["need"]] + list(range(a, b)))
        assert set(covering) <= set(have)
        assert all(t < a for _, t in p["need"]) and [t for _, t in p["need"]] == sorted(t for _, t in p["need"])
        for src, t in p["need"]:
            assert plan[src]["tile_rows"][0] <= t < plan[src]["tile_rows"][1] and src < k
    if world == 1 or len(rs) == 1:
        assert all(p["need"] == [] for p in plan)
    # the shift-back rule can make the last tile row overlap two earlier ones: then a band needs two halo rows
    if (H, tile, overlap, world) == (1000, 512, 128, 3):
        assert len(plan[2]["need"]) == 2


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bathymetric_gnn_amd.models.pipeline import exchange_halo_tile_rows, survey_shard_plan
    rs, re = _rows(1000, 512, 128)                                  # tile rows 0,384,488 -> rank 2 needs rows 0 and 1
    plan = survey_shard_plan(rs, re, 1000, world)
    for p in plan:
        p["_halo_shape"], p["_halo_dtype"], p["_halo_device"] = (1000,), torch.float32, torch.device("cpu")
    mine = {t: torch.full((1000,), float(100 * rank + t)) + torch.arange(1000) for t in range(*plan[rank]["tile_rows"])}
    got = exchange_halo_tile_rows(plan, rank, mine)
    q.put((rank, {t: v.numpy().copy() for t, v in got.items()}, plan[rank]["need"]))
    dist.barrier()
    dist.destroy_process_group()


def test_halo_exchange_gloo_three_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 3, port, q)) for r in range(3)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=180) for _ in range(3)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    n_recv = 0
    for rank, got, need in res:
        assert sorted(got) == sorted(t for _, t in need)
        for src, t in need:
            assert np.array_equal(got[t], 100 * src + t + np.arange(1000, dtype=np.float32))
            n_recv += 1
    assert n_recv == 3            # rank 1 <- row 0; rank 2 <- rows 0 and 1


def _gather_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bathymetric_gnn_amd.models.pipeline import gather_bands_to_rank0, survey_shard_plan
    rs, re = _rows(1000, 512, 128)
    plan = survey_shard_plan(rs, re, 1000, world)          # world 4 > 3 tile rows: one rank owns nothing
    R0, R1 = plan[rank]["cell_rows"]
    W = 37
    band = None
    if R1 > R0:
        rows = torch.arange(R0, R1, dtype=torch.float32)[None, :, None]
        band = (rows * 1000 + torch.arange(W, dtype=torch.float32)[None, None, :] + 0.25 * torch.arange(4, dtype=torch.float32)[:, None, None]).contiguous()
    host = gather_bands_to_rank0(plan, rank, band, W)
    q.put((rank, None if host is None else host.copy(), [list(p["cell_rows"]) for p in plan]))
    dist.barrier()
    dist.destroy_process_group()


def test_band_gather_to_rank0_gloo_four_ranks():
    """The sharded survey's last step: bands travel to rank 0 as tensors (one point-to-point message per band), ranks without
    a band send nothing, rank 0 assembles [channels, H, W]; the other ranks get None."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, 4, port, q)) for r in range(4)]
    for pr in procs:
        pr.start()
    res = {r: (h, bands) for r, h, bands in (q.get(timeout=180) for _ in range(4))}
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert all(res[r][0] is None for r in (1, 2, 3))
    host, bands = res[0]
    assert host.shape == (4, 1000, 37) and sum(b[1] - b[0] for b in bands) == 1000 and any(b[0] == b[1] for b in bands)
    exp = (np.arange(1000, dtype=np.float32)[None, :, None] * 1000 + np.arange(37, dtype=np.float32)[None, None, :]
           + 0.25 * np.arange(4, dtype=np.float32)[:, None, None])
    assert np.array_equal(host, exp)


def _gather_worker_rank0_empty(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bathymetric_gnn_amd.models.pipeline import gather_bands_to_rank0, survey_shard_plan
    rs, re = _rows(400, 512, 128)                           # ONE tile row, three ranks: rank 0 owns no rows
    plan = survey_shard_plan(rs, re, 400, world)
    R0, R1 = plan[rank]["cell_rows"]
    band = torch.full((4, R1 - R0, 9), float(rank), dtype=torch.float32) if R1 > R0 else None
    host = gather_bands_to_rank0(plan, rank, band, 9, device="cpu")
    q.put((rank, None if host is None else host.copy(), [list(p["cell_rows"]) for p in plan]))
    dist.barrier()
    dist.destroy_process_group()


def test_band_gather_when_rank0_owns_no_rows():
    """with more ranks than tile rows rank 0 can be the rank without a band; it still has to receive the others'
    (the receive device comes from the caller, not from rank 0's own band)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker_rank0_empty, args=(r, 3, port, q)) for r in range(3)]
    for pr in procs:
        pr.start()
    res = {r: (h, bands) for r, h, bands in (q.get(timeout=180) for _ in range(3))}
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    host, bands = res[0]
    assert bands[0][0] == bands[0][1], "the case under test: rank 0 without rows"
    owner = [r for r in range(3) if bands[r][1] > bands[r][0]]
    assert len(owner) == 1 and host.shape == (4, 400, 9) and np.all(host == float(owner[0]))
