#!/usr/bin/env python3
"""Headline benchmark: classified tile-nodes/s of the fused hot path on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--tiles B] [--tile-size S]

A step = one pass of the hot path over one batch of synthetic tiles already resident in HBM:
``bgnn_infer_tiles`` = graph build (compaction, 5x5 stats, node features, stencil table, edge
attributes) -> 4-layer GAT forward -> heads -> node-to-grid scatter.  Workload at every N: per GPU
a batch of B=128 tiles of 256x256 (BASELINE config[1]'s tile: k=8 / 8-connected, fp32, 4 layers,
all-valid synthetic depth; batched as the metric's "tile-batch").  Tiles are independent, so ranks
share nothing: weak scaling, no data-path collective (only the timing barrier / max).

Prints ONE JSON line (rank 0).  Extra objects: ``roofline`` (dominant kernel, HIP-event timed on the
library's stream during the timed steps), ``kernels`` (all kernel classes), ``cpu_baseline`` (the CPU
oracle timed on this box's host cores on a bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # exact-f32 MFMA (= vector fp32 peak)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md)


def algorithmic_model(num_layers=4, hidden=64, heads=4, in_ch=7, classes=3, deg=8, edge_dim=3, act_bytes=4):
    """SURVEY.md 8(d) per-node figures (compulsory traffic: each tensor read once + written once).  ``act_bytes``: bytes per
    stored layer activation (4 = fp32, the reference's dtype; 2 = bf16 storage of BASELINE config 3).  ``fused_bytes``
    prices the fused launches (xW in / next xW out at ``act_bytes``; attention dots and edge attributes stay f32)."""
    agg_bytes = 0
    gemm_flops = 2 * (in_ch * hidden)                              # feature extractor layer 1 (layer 2 is folded into lin_0: executed flops)
    gemm_bytes = 4 * (8 + hidden) + 4 * (hidden + hidden)
    for l in range(num_layers):
        last = l == num_layers - 1
        H = 1 if last else heads
        d_in = hidden if l == 0 else hidden * heads
        hc = H * hidden
        # read xW + write out + alpha_src/alpha_dst + edge attributes
        agg_bytes += 4 * hc + 4 * hc + 4 * 2 * H + 4 * deg * edge_dim
        gemm_flops += 2 * d_in * hc
        gemm_bytes += 4 * (d_in + hc)
    nh = 3
    gemm_flops += 2 * hidden * nh * (hidden // 2)
    gemm_bytes += 4 * (hidden + nh * (hidden // 2))
    build_bytes = 5 + 28 + 4 + 4 * deg * edge_dim                   # native-internal graph form (133 B)
    # fused path: launch l = aggregate of layer l + GEMM of layer l+1 (last: + heads + grid scatter);
    # the front GEMMs (feature extractor + lin of layer 0) stay separate
    fused_flops, fused_bytes = 0, 0
    # executed flops: the extractor's second Linear is folded into lin of layer 0 (no activation between them), so the
    # hidden x hidden product of the reference is not run
    front_flops = 2 * (in_ch * hidden + hidden * (hidden * (heads if num_layers > 1 else 1)))
    for l in range(num_layers):
        last = l == num_layers - 1
        H = 1 if last else heads
        hc = H * hidden
        if not last:
            Hn = 1 if l + 1 == num_layers - 1 else heads
            nc = Hn * hidden
            fused_flops += 2 * hc * nc
            fused_bytes += act_bytes * hc + 4 * 2 * H + 4 * deg * edge_dim + act_bytes * nc + 4 * 2 * Hn
        else:
            fused_flops += 2 * hidden * nh * (hidden // 2)
            fused_bytes += act_bytes * hc + 4 * 2 * H + 4 * deg * edge_dim + 4 * 3
    return {"aggregate_bytes": agg_bytes, "gemm_flops": gemm_flops, "gemm_bytes": gemm_bytes, "build_bytes": build_bytes,
            "fused_flops": fused_flops, "fused_bytes": fused_bytes, "front_flops": front_flops}


def cpu_baseline(n_tiles, tile, sd, seed0, connectivity="8-connected"):
    """The CPU oracle (vectorised numpy graph build + fp32 torch forward issuing torch_geometric's op
    sequence + scatter) on `n_tiles` tiles of the same workload.  kind = "port"."""
    from bathymetric_gnn_amd import synthetic
    from oracle import gat_cpu, graph_cpu
    tiles = [synthetic.synthetic_tile(tile, tile, seed0 + i, "V0") for i in range(n_tiles)]
    # warm-up on a small tile (thread pools, allocator)
    d, m, _ = synthetic.synthetic_tile(64, 64, 1, "V0")
    gat_cpu.process_tile(sd, graph_cpu.build_graph(d, m, None, (0.5, 0.5)))
    nodes = 0
    t0 = time.perf_counter()
    tg = 0.0
    k = {"4-connected": 4, "8-connected": 8, "16-dilated": 16}[connectivity]
    for d, m, _ in tiles:
        a = time.perf_counter()
        g = graph_cpu.build_graph(d, m, None, (0.5, 0.5), connectivity=connectivity)
        tg += time.perf_counter() - a
        gat_cpu.process_tile(sd, g)
        nodes += g.num_nodes
    dt = time.perf_counter() - t0
    return {"value": nodes / dt, "unit": "nodes/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_tiles} tiles of {tile}x{tile} (k={k}, 4-layer GAT, fp32): numpy graph build + torch CPU forward "
                      f"+ scatter, {dt:.1f} s wall ({tg:.1f} s of it graph build); os.cpu_count()={os.cpu_count()}"}


def launch_plan(gpus, env, n_devices):
    """What `bench.py --gpus N` does with its process: ("worker", None) when a launcher (torch.distributed.run, or this
    file's own spawn) already set WORLD_SIZE; ("single", None) for N = 1; ("spawn", None) when N > 1 ranks must be
    started from here; ("error", reason) when that cannot work.  Pure function (unit-tested on CPU)."""
    if gpus < 1:
        return "error", f"--gpus {gpus}: need at least one GPU"
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != gpus:
            return "error", f"--gpus {gpus} but WORLD_SIZE={world}"
        return ("worker" if world > 1 else "single"), None
    if gpus == 1:
        return "single", None
    if n_devices < gpus:
        return "error", f"--gpus {gpus} but only {n_devices} GPU(s) visible on this node"
    return "spawn", None


def spawn_ranks(n, argv, env=None, python=None):
    """Start `n` child processes of `argv` (one per GPU: RANK = LOCAL_RANK = i, WORLD_SIZE = n, rendezvous on
    127.0.0.1 at a free port), wait for all of them and return the worst exit status.  The parent never touches the
    GPU, so nothing is exec'd or forked from a process that has initialised HIP."""
    env = dict(os.environ if env is None else env)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for i in range(n):
        e = dict(env, RANK=str(i), LOCAL_RANK=str(i), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([python or sys.executable] + list(argv), env=e))
    rc = 0
    for p in procs:
        r = p.wait()
        if r != 0 and rc == 0:
            rc = r if r > 0 else 1
    if rc != 0:                                            # one rank failed: do not leave the others waiting in a barrier
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tiles", type=int, default=128, help="tiles per batch per GPU")
    ap.add_argument("--tile-size", type=int, default=256)
    ap.add_argument("--variant", default="V0", choices=["V0", "V1"])
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed steps (no PCIe-inclusive / split-path / single-tile side measurements, no CPU baseline): "
                         "what the rocprofv3 passes run, so that a kernel's average duration in the trace is the full-batch one")
    ap.add_argument("--split-f16", action="store_true",
                    help="opt-in matrix path: float16 hi/lo operand split (fp16x3), float32 accumulation (|activations| < 65504)")
    ap.add_argument("--split-bf16", action="store_true",
                    help="opt-in matrix path: bf16 hi/lo operand split (bf16x3) with float32 accumulation instead of exact-f32 MFMAs")
    ap.add_argument("--unfused", action="store_true",
                    help="run K3 / K4 / K5 / K6 as separate kernels (standalone gather-aggregate roofline)")
    ap.add_argument("--cpu-tiles", type=int, default=4)
    ap.add_argument("--workload", default="tiles", choices=["tiles", "vr", "c3"],
                    help="tiles: B uniform tiles per step (headline). vr: BASELINE config 4 -- a stream of 4096 ragged "
                         "refinement grids (3x3..50x50, in=8) packed by the reference's 50 000-node batch budget. "
                         "c3: BASELINE configs[2] -- B tiles of 256x256 with k=16 (the '16-dilated' stencil, a build-side "
                         "extension) and bf16 node features (layer activations stored as bf16, bf16 MFMA, f32 accumulate)")
    ap.add_argument("--connectivity", default=None, choices=["4-connected", "8-connected", "16-dilated"])
    ap.add_argument("--bf16", action="store_true", help="matrix_path = bf16 (bf16 activation storage + bf16 MFMA)")
    ap.add_argument("--vr-grids", type=int, default=4096)
    ap.add_argument("--vr-budget", type=int, default=50000)
    ap.add_argument("--vr-streams", type=int, default=4,
                    help="vr workload: library contexts (HIP streams) the batches are dealt over: the tail of one 50 000-node "
                         "batch's kernels overlaps the head of the next one's")
    args = ap.parse_args()

    # N > 1 without a launcher: start the N ranks here, BEFORE anything touches the GPU (device_count() does not
    # initialise HIP on this image), wait for them and leave with their status.
    mode, why = launch_plan(args.gpus, os.environ, torch.cuda.device_count())
    if mode == "error":
        raise SystemExit(f"bench.py: {why}")
    if mode == "spawn":
        raise SystemExit(spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)     # RCCL; used only for the timing barrier / max

    from bathymetric_gnn_amd import runtime as rt, synthetic
    from bathymetric_gnn_amd.data import GraphBuilder
    from bathymetric_gnn_amd.models import BathymetricGNN
    from bathymetric_gnn_amd.models.pipeline import TileBatchEngine

    B, S = args.tiles, args.tile_size
    in_ch = 8 if args.workload == "vr" else 7
    sd = synthetic.synthetic_state_dict(in_channels=in_ch, num_layers=args.layers, seed=1234)
    model = BathymetricGNN(in_channels=in_ch, num_gnn_layers=args.layers, edge_dim=3, dropout=0.0)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    model.to(dev).eval()
    c3 = args.workload == "c3"
    conn = args.connectivity or ("16-dilated" if c3 else "8-connected")
    deg = {"4-connected": 4, "8-connected": 8, "16-dilated": 16}[conn]
    if c3:
        args.workload, args.bf16 = "tiles", True
    gb = GraphBuilder(connectivity=conn, device=dev)
    eng = TileBatchEngine(model, gb, dev)
    ctx = eng.ctx
    # run-time switches of the library context (the environment is only read when a context is created)
    if args.unfused:
        ctx.set_option("fused", 0)
    if args.split_f16 or args.split_bf16 or args.bf16:
        ctx.set_option("matrix_path", "bf16" if args.bf16 else "fp16x3" if args.split_f16 else "bf16x3")
    split_main = {0: None, 1: "bf16x3", 2: "fp16x3", 3: "bf16"}[ctx.get_option("matrix_path")]
    bf16 = split_main == "bf16"
    unfused = not ctx.get_option("fused")
    nn_dev = torch.zeros(1, dtype=torch.int64, device=dev)

    if args.workload == "tiles":
        # synthetic batch, resident in HBM before the timed region (a few distinct tiles, tiled to B)
        n_distinct = min(B, 8)
        depth, mask, _ = synthetic.synthetic_tile_batch(n_distinct, S, S, 100 + 1000 * rank, args.variant)
        reps = (B + n_distinct - 1) // n_distinct
        depth = np.concatenate([depth] * reps)[:B]; mask = np.concatenate([mask] * reps)[:B]
        d_t = torch.from_numpy(depth).to(dev).reshape(-1)
        m_t = torch.from_numpy(mask.view(np.uint8)).to(dev).reshape(-1)
        hw = np.tile(np.array([[S, S]], np.int32), (B, 1)); res = np.full((B, 2), 0.5)
        out = torch.empty((3, d_t.numel()), dtype=torch.float32, device=dev)
        nodes_per_step = int(mask.sum())
        workload_name = (f"{B} tiles of {S}x{S} per GPU per step, {conn} (k={deg}), {args.layers}-layer GAT "
                         f"(hidden 64, heads 4), mask {args.variant}, inputs resident in HBM"
                         + (", layer activations stored as bf16 (bf16 MFMA, f32 softmax / aggregation / accumulation)" if bf16 else ""))

        def step():
            eng.infer_device(hw, res, d_t, m_t, None, out=out, n_nodes_out=nn_dev)
    else:
        # config 4: ragged refinement grids, packed greedily in stream order until the node budget is reached
        grids = synthetic.vr_grid_stream(args.vr_grids, seed0=1000 + 100000 * rank)
        batches, cur, cur_nodes, nodes_per_step = [], [], 0, 0
        for d, u, r in grids:
            m = (d != synthetic.NODATA) & np.isfinite(d)
            cur.append((d, m, u, r)); cur_nodes += int(m.sum())
            if cur_nodes >= args.vr_budget:
                batches.append(cur); nodes_per_step += cur_nodes; cur, cur_nodes = [], 0
        if cur:
            batches.append(cur); nodes_per_step += cur_nodes
        dev_batches = []
        for b in batches:
            hw_b, res_b, d_b, m_b, u_b = gb.upload_tiles([x[0] for x in b], [x[1] for x in b], [x[2] for x in b], [x[3] for x in b])
            dev_batches.append((hw_b, res_b, d_b, m_b, u_b, torch.empty((3, d_b.numel()), dtype=torch.float32, device=dev)))
        workload_name = (f"{args.vr_grids} ragged refinement grids (3x3..50x50, in=8) per GPU per step in {len(batches)} batches "
                         f"of >= {args.vr_budget} nodes dealt over {max(1, args.vr_streams)} HIP stream(s), 8-connected, {args.layers}-layer GAT, "
                         f"inputs resident in HBM")

        engines = [eng] + [TileBatchEngine(model, gb, dev, ctx=rt.new_context(dev)) for _ in range(max(1, args.vr_streams) - 1)]
        for e in engines[1:]:
            for k in ("fused", "matrix_path"):
                e.ctx.set_option(k, ctx.get_option(k))

        def step():
            for i, (hw_b, res_b, d_b, m_b, u_b, o_b) in enumerate(dev_batches):
                engines[i % len(engines)].infer_device(hw_b, res_b, d_b, m_b, u_b, out=o_b, defer_end=True)
            for e in engines:
                e.ctx.end()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    all_ctx = [ctx] + ([e.ctx for e in engines[1:]] if args.workload == "vr" else [])
    # HIP events around every kernel class, recorded inside the timed region.  The vr workload issues ~700 small launches per
    # step over several streams: there the event pairs (1 400 records a step) are host work that slows the stream itself
    # down, so its timed steps run bare and the same K steps are repeated with events for the kernel breakdown.
    events_in_timed_region = args.workload != "vr"
    if events_in_timed_region:
        for c in all_ctx:
            c.profile(rt.K_NAMES)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if not events_in_timed_region:
        for c in all_ctx:
            c.profile(rt.K_NAMES)
        for _ in range(args.steps):
            step()
        barrier()
    prof = {k: {"ms": 0.0, "launches": 0} for k in rt.K_NAMES}
    for c in all_ctx:
        for k, v in c.profile_read().items():
            prof[k]["ms"] += v["ms"]; prof[k]["launches"] += v["launches"]
        c.profile([])
    if args.workload == "tiles":
        assert int(nn_dev.item()) == nodes_per_step
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        am = algorithmic_model(num_layers=args.layers, deg=deg, act_bytes=2 if bf16 else 4)
        nodes_total = nodes_per_step * args.steps * world
        value = nodes_total / elapsed
        kernels = {}
        for k, v in prof.items():
            if v["launches"]:
                kernels[k] = {"ms_per_step": v["ms"] / args.steps, "launches_per_step": v["launches"] / args.steps}
        n_local = nodes_per_step * args.steps
        def roof(kernel, key, bound, work_per_node, note):
            t = prof[key]["ms"] / 1e3
            if t <= 0:
                return None
            unit, peak, scale = ("GB/s", HBM_PEAK_GBS, 1e9) if bound == "hbm" else ("TFLOP/s", MFMA_F32_PEAK_TFLOPS, 1e12)
            if bound == "mfma_bf16":
                unit, peak, scale, bound = "TFLOP/s", MFMA_BF16_PEAK_TFLOPS, 1e12, "mfma"
            ach = work_per_node * n_local / t / scale
            return {"kernel": kernel, "bound": bound, "unit": unit, "peak": peak, "achieved": ach, "frac": ach / peak,
                    "avg_launch_ms": prof[key]["ms"] / max(prof[key]["launches"], 1),
                    "launches_per_step": prof[key]["launches"] / args.steps,
                    "algorithmic_work_per_node_per_forward": work_per_node, "traffic": None, "note": note}
        roofs = {}
        if prof["fused"]["launches"]:
            if bf16:
                roofs["fused_mfma"] = roof("gat_layer_fused_kernel", "fused", "mfma_bf16", am["fused_flops"],
                                           "bf16 MFMA, f32 accumulate: priced against the dense bf16 MFMA peak (the kernel is not matrix-bound)")
            elif split_main:
                roofs["fused_mfma"] = roof("gat_layer_fused_kernel", "fused", "mfma_bf16", 3 * am["fused_flops"],
                                           f"{split_main}: executed flops = 3 x algorithmic, priced against the dense bf16 / f16 MFMA peak")
            else:
                roofs["fused_mfma"] = roof("gat_layer_fused_kernel", "fused", "mfma", am["fused_flops"],
                                           "K4 gather-softmax-aggregate fused with the next layer's exact-f32 MFMA GEMM (last: heads + scatter)")
            roofs["fused_hbm"] = roof("gat_layer_fused_kernel", "fused", "hbm", am["fused_bytes"],
                                      "same launches priced by compulsory HBM bytes (read xW + attrs, write next xW)")
            roofs["front_gemm"] = roof("gemm_f32_kernel", "gemm", "mfma", am["front_flops"], "gemm_wres64_kernel: feature extractor layer 1 in front of (extractor layer 2 folded into) lin of layer 0, one launch (executed flops)")
        else:
            roofs["aggregate_hbm"] = roof("gat_aggregate_tiled_kernel", "aggregate", "hbm", am["aggregate_bytes"],
                                          "standalone K4 (LDS-tiled gather-softmax-aggregate + BN + ReLU)")
            roofs["gemm_mfma"] = roof("gemm_f32_kernel", "gemm", "mfma", am["gemm_flops"], "all K3 GEMMs, exact f32 MFMA")
        roofs = {k: v for k, v in roofs.items() if v}
        # PMC traffic is collected in separate rocprofv3 --pmc passes (profiles/); attach if present
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        # (the counters were collected on the default workload only: 128 tiles of 256 x 256, 4 layers)
        pmc_key = (":c3" if (deg == 16 and bf16) else None if (deg != 8 or bf16) else ":split" if split_main else "")
        if os.path.exists(pmc) and args.workload == "tiles" and (B, S, args.layers) == (128, 256, 4) and pmc_key is not None:
            try:
                t = json.load(open(pmc))
                for v in roofs.values():
                    e = t.get(v["kernel"] + pmc_key) if pmc_key else None
                    if pmc_key == ":c3" and e is None:
                        continue
                    v["traffic"] = (e or t.get(v["kernel"], {})).get("hbm_bytes_per_launch")
                    if v["traffic"] is not None:
                        v["traffic_source"] = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                               "command, collected separately (not measured by this run)")
            except Exception:
                pass
        if "fused_mfma" in roofs:
            dominant = roofs["fused_hbm"] if split_main else roofs["fused_mfma"]    # bf16x3: the fused kernel is memory-side bound
        else:
            dominant = max(roofs.values(), key=lambda v: v["avg_launch_ms"] * v["launches_per_step"])
        line = {
            "metric": "classified tile-nodes/s (fused graph build + 4-layer GAT forward + scatter)",
            "value": value, "unit": "nodes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("bf16 (layer activations stored as bf16, bf16 MFMA; f32 softmax / aggregation / accumulation / outputs)" if bf16
                      else f"f32 ({split_main} split-operand MFMA, f32 accumulate)" if split_main else "f32"),
            "data": "synthetic",
            "config": {"workload": workload_name,
                       "tiles_per_gpu": B if args.workload == "tiles" else args.vr_grids, "tile": S if args.workload == "tiles" else "3..50",
                       "nodes_per_step_per_gpu": nodes_per_step,
                       "parallelism": f"tile-sharded x{world}, no collective"},
            "roofline": {k: dominant.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                                       "kernel", "avg_launch_ms")},
            "rooflines": roofs, "kernels": kernels,
            "kernel_events": "inside the timed region" if events_in_timed_region else "separate pass of the same steps (timed steps ran bare)",
            "path": "unfused" if unfused else "fused",
            "matrix_path": "bf16 storage + bf16 MFMA (BASELINE configs[2])" if bf16 else f"{split_main} split (opt-in)" if split_main else "exact f32",
        }
        extras = not args.no_extras
        if extras and world == 1 and args.workload == "tiles" and not unfused:
            # The same batch handed over as HOST arrays (the reference's boundary): pinned staging, H2D / compute /
            # D2H on three streams, two slots in flight.  Reported beside `value`, never as it.
            from bathymetric_gnn_amd.models.pipeline import HostTilePipeline
            hp = HostTilePipeline(eng, B, S, S, resolution=(0.5, 0.5))
            for _ in range(2):
                hp.submit(depth, mask)
            list(hp.drain())
            n_pc = max(4, min(args.steps, 10))
            torch.cuda.synchronize(dev); t1 = time.perf_counter()
            got = 0
            for _ in range(n_pc):
                got += hp.submit(depth, mask) is not None
            got += len(list(hp.drain()))
            t_pc = time.perf_counter() - t1
            assert got == n_pc
            line["pcie_inclusive"] = {"value": nodes_per_step * n_pc / t_pc, "unit": "nodes/s", "ms_per_step": t_pc / n_pc * 1e3,
                                      "steps": n_pc, "bytes_per_cell": {"h2d": 5, "d2h": 12},
                                      "note": "host numpy tiles in, host grids out: pinned double-buffered staging, H2D / "
                                              "compute / D2H overlapped on three streams (HostTilePipeline)"}
        if extras and world == 1 and args.workload == "tiles" and not unfused and not split_main:
            # Opt-in matrix path, reported BESIDE the headline (never as it): bf16 hi/lo operand split (bf16x3) on the bf16
            # matrix cores with float32 accumulation.  Same inputs, same timing protocol; the distance of its class logits
            # to the exact-f32 path is measured on one tile of the batch.
            from bathymetric_gnn_amd.data import GraphBuilder as _GB
            g1 = _GB(device=dev).build_graph(depth[0], mask[0], None, (0.5, 0.5))
            lg_exact = model.predict(g1)["class_logits"].clone()
            for key, env, instr in (("split_bf16x3", "bf16x3", "v_mfma_f32_32x32x16_bf16"),
                                    ("split_fp16x3", "fp16x3", "v_mfma_f32_32x32x16_f16")):
                ctx.set_option("matrix_path", env)
                try:
                    lg_split = model.predict(g1)["class_logits"]
                    for _ in range(2):
                        step()
                    n_sp = max(4, min(args.steps, 10))
                    torch.cuda.synchronize(dev); t2 = time.perf_counter()
                    for _ in range(n_sp):
                        step()
                    torch.cuda.synchronize(dev); t_sp = time.perf_counter() - t2
                finally:
                    ctx.set_option("matrix_path", "exact_f32")
                line[key] = {"value": nodes_per_step * n_sp / t_sp, "unit": "nodes/s", "ms_per_step": t_sp / n_sp * 1e3,
                             "steps": n_sp, "max_abs_logit_diff_vs_exact_f32": float((lg_split - lg_exact).abs().max().item()),
                             "note": f"matrix_path={env}: layer GEMMs as hi/lo operand splits on {instr}, float32 accumulate; "
                                     "opt-in, not the headline"}
        if extras and world == 1 and args.workload == "tiles" and not unfused and B > 1:
            # BASELINE configs[1]: ONE 256 x 256 tile per step (latency-bound: 65 536 nodes cannot fill 256 CUs)
            d1 = d_t[: S * S].clone(); m1 = m_t[: S * S].clone()
            hw1 = np.array([[S, S]], np.int32); res1 = np.full((1, 2), 0.5)
            out1 = torch.empty((3, S * S), dtype=torch.float32, device=dev)
            for _ in range(5):
                eng.infer_device(hw1, res1, d1, m1, None, out=out1)
            torch.cuda.synchronize(dev); t3 = time.perf_counter()
            n_one = 50
            for _ in range(n_one):
                eng.infer_device(hw1, res1, d1, m1, None, out=out1)
            torch.cuda.synchronize(dev); t_one = (time.perf_counter() - t3) / n_one
            line["single_tile"] = {"value": int(mask[0].sum()) / t_one, "unit": "nodes/s", "ms_per_tile": t_one * 1e3, "steps": n_one,
                                   "note": f"configs[1]: one {S}x{S} tile per step (back-to-back launches, inputs resident in HBM)"}
        if extras and world == 1 and not args.no_cpu_baseline and args.workload == "tiles":
            line["cpu_baseline"] = cpu_baseline(args.cpu_tiles, S, sd, 100, conn)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
