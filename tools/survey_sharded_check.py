#!/usr/bin/env python3
"""Row-band sharded survey path on real kernels: WORLD ranks (gloo transport, all on cuda:0 of a one-GPU box -- on an
8-GPU node each rank takes its own GPU and the nccl backend) run BathymetricPipeline.process_grid_device
(each rank uploads only the survey rows its own tile rows span, exchanges halo tile rows point to point, stitches its band;
the bands go to rank 0 as tensors); rank 0 also runs the unsharded path and checks that the result equals it bit for bit.
The launcher process itself never touches the GPU (children are started before anything initialises HIP)."""
import argparse, json, os, socket, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, size, tile, overlap, q):
    sys.path.insert(0, ROOT)
    import torch, torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    n_gpu = torch.cuda.device_count()
    dev = torch.device(f"cuda:{rank % n_gpu}")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl" if n_gpu >= world else "gloo", rank=rank, world_size=world)
    from bathymetric_gnn_amd import synthetic
    from bathymetric_gnn_amd.config import Config
    from bathymetric_gnn_amd.models import BathymetricGNN, BathymetricPipeline
    cfg = Config(); cfg.tile.tile_size, cfg.tile.overlap, cfg.tile.min_valid_ratio = tile, overlap, 0.3
    pipe = BathymetricPipeline(cfg, tile_batch=16)
    sd = synthetic.synthetic_state_dict(seed=1234)
    m = BathymetricGNN(in_channels=7, edge_dim=3, dropout=0.0); m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    pipe.set_model(m)
    d, mk, _ = synthetic.synthetic_tile(size, size - 37, 5, "V1")
    d[: size // 3, : size // 4] = 1.0e6
    from bathymetric_gnn_amd.data import BathymetricGrid
    grid = BathymetricGrid(depth=d, nodata_value=1.0e6, resolution=(0.5, 0.5))
    depth = torch.from_numpy(d).to(dev); valid = (depth != 1.0e6) & torch.isfinite(depth)
    pipe.process_survey_device(depth[:tile, :tile].contiguous(), valid[:tile, :tile].contiguous(), None, (0.5, 0.5))   # warm-up
    (lo, hi), plan = pipe.survey_rows_of_rank(d.shape, rank, world)
    dist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
    # the product path: every rank uploads ONLY its own rows, stitches its band, bands go to rank 0 as tensors
    res = pipe.process_grid_device(grid)
    torch.cuda.synchronize(); dist.barrier(); dt = time.perf_counter() - t0
    assert (res is None) == (rank != 0)
    counts = [None] * world
    dist.all_gather_object(counts, (list(plan[rank]["cell_rows"]), [lo, hi], list(getattr(pipe, "last_tile_counts", (0, 0)))))   # (bookkeeping only)
    if rank == 0:
        full = pipe.process_survey_device(depth, valid, None, (0.5, 0.5)).cpu().numpy()
        got = np.stack([res["classification"], res["confidence"], res["correction"], res["cleaned_depth"]])
        same = got.shape == full.shape and np.array_equal(np.nan_to_num(got).view(np.uint32), np.nan_to_num(full).view(np.uint32)) \
            and np.array_equal(np.isnan(got), np.isnan(full))
        q.put({"world": world, "survey": list(d.shape), "tile": [tile, overlap], "bands": [c[0] for c in counts],
               "rows_uploaded_per_rank": [c[1] for c in counts], "tiles_per_rank": [c[2] for c in counts], "sharded_wall_s": dt,
               "bit_identical_to_single_gpu": bool(same)})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2); ap.add_argument("--size", type=int, default=1500)
    ap.add_argument("--tile", type=int, default=256); ap.add_argument("--overlap", type=int, default=64)
    args = ap.parse_args()
    import multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, args.world, port, args.size, args.tile, args.overlap, q)) for r in range(args.world)]
    for p in procs:
        p.start()
    out = q.get(timeout=900)
    for p in procs:
        p.join(timeout=120)
    out["exit_codes"] = [p.exitcode for p in procs]
    print(json.dumps(out))
    sys.exit(0 if out["bit_identical_to_single_gpu"] and all(c == 0 for c in out["exit_codes"]) else 1)
