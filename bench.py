#!/usr/bin/env python3
"""Headline benchmark: classified tile-nodes/s of the fused hot path on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload tiles|c3|vr|survey] ...

A step = one pass of the hot path over one batch of synthetic input already resident in HBM:
``bgnn_infer_tiles`` = graph build (compaction, 5x5 stats, node features, stencil table, edge
attributes) -> 4-layer GAT forward -> heads -> node-to-grid scatter.

Workloads (BASELINE.json ``configs``):

* ``tiles`` (default, the headline; configs[1]'s tile batched as the metric's "tile-batch"): per GPU a batch of B = 128
  tiles of 256 x 256, k = 8 (8-connected), fp32, 4 layers, all-valid synthetic depth.
* ``c3`` (configs[2]): the same batch with k = 16 (the '16-dilated' stencil) and bf16 node features.
* ``vr`` (configs[3]): a stream of 4096 ragged refinement grids (3x3 .. 50x50, in = 8) packed by the reference's
  50 000-node batch budget (``scripts/inference_native.py:128``), on one library context or dealt over several.
* ``survey`` (configs[4]): one synthetic survey resident in HBM, overlapping 512 x 512 tiles (overlap 128) cut, classified
  and stitched on the device (``BathymetricPipeline.process_survey_device``).  Under ``--gpus N`` the ONE survey is split
  into row bands (strong scaling, point-to-point halo tile rows); every other workload is weak scaling, no collective.

Prints ONE JSON line (rank 0).  Extra objects: ``roofline`` (dominant kernel, HIP-event timed on the library's stream
during the timed steps), ``rooflines`` / ``kernels`` (all kernel classes), ``cpu_baseline`` (the CPU oracle timed on this
box's host cores on a bounded sample; rank 0, N = 1 only) and -- in the default run at N = 1 -- ``config3``, ``config4``,
``config5``: the other single-GPU BASELINE configs measured with the same protocol, each with its own ``roofline``.
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
DEG = {"4-connected": 4, "8-connected": 8, "16-dilated": 16}


def algorithmic_model(num_layers=4, hidden=64, heads=4, in_ch=7, classes=3, deg=8, edge_dim=3, act_bytes=4, layer0_af=False):
    """SURVEY.md 8(d) per-node figures (compulsory traffic: each tensor read once + written once).  ``act_bytes``: bytes per
    stored layer activation (4 = fp32, the reference's dtype; 2 = bf16 storage of BASELINE config 3).  ``fused_bytes``
    prices the fused launches (xW in / next xW out at ``act_bytes``; attention dots and edge attributes stay f32);
    ``front_bytes`` the front GEMM launch (32-byte feature row in, xW_0 + attention dots out).  ``layer0_af`` (bf16 path, default
    shape): layer 0 aggregates the extractor's ``hidden``-channel output and applies lin_0 inside the fused launch, so the front launch
    writes and the first fused launch reads ``hidden`` channels per node instead of ``heads * hidden`` -- priced at what that form has to
    move (SURVEY's figure for the reference's operation order would flatter it: 3 772 instead of 3 388 B/node at k = 16)."""
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
    H0 = heads if num_layers > 1 else 1
    front_flops = 2 * (in_ch * hidden + hidden * (hidden * H0))
    front_bytes = 32 + act_bytes * hidden * (1 if layer0_af else H0) + 4 * 2 * H0
    for l in range(num_layers):
        last = l == num_layers - 1
        H = 1 if last else heads
        hc = H * hidden
        if not last:
            Hn = 1 if l + 1 == num_layers - 1 else heads
            nc = Hn * hidden
            fused_flops += 2 * hc * nc
            src = hidden if (layer0_af and l == 0) else hc
            fused_bytes += act_bytes * src + 4 * 2 * H + 4 * deg * edge_dim + act_bytes * nc + 4 * 2 * Hn
        else:
            fused_flops += 2 * hidden * nh * (hidden // 2)
            fused_bytes += act_bytes * hc + 4 * 2 * H + 4 * deg * edge_dim + 4 * 3
    return {"aggregate_bytes": agg_bytes, "gemm_flops": gemm_flops, "gemm_bytes": gemm_bytes, "build_bytes": build_bytes,
            "fused_flops": fused_flops, "fused_bytes": fused_bytes, "front_flops": front_flops, "front_bytes": front_bytes}


def plain_backbone_model(gnn_type, num_layers=4, hidden=64, in_ch=7, deg=8, classes=3):
    """Compulsory HBM bytes and executed flops per node and forward of the non-attention backbones (reference models/gnn.py:120-143).
    Since round 4 a layer is ONE launch of the fused layer kernel in its plain-backbone mode (aggregate -> GEMM -> per-column post-op;
    GIN's second Linear stays a GEMM launch): ``fused_bytes`` = h row in + h row out per layer (+ 4 B of deg^-1/2 for GCN),
    ``fused_flops`` = the layer's GEMM (GraphSAGE: K = 128, the stacked [lin_l ; lin_r]); ``gemm_bytes`` = every plain GEMM launch
    (extractor 2, heads' first layers 1, GIN + 1 per layer).  K = 64 GEMMs have 16 flop per byte -- below the 19.7 flop/B balance of the
    exact-f32 MFMA -- so they are priced by bytes; the fused launches are priced both ways."""
    row = 4 * hidden
    gemm = (32 + row) + (row + row) + (row + 4 * 3 * (hidden // 2))          # extractor layers, heads' first layers
    fused_bytes, fused_flops = 0, 0
    for _ in range(num_layers):
        fused_bytes += row + row + (4 if gnn_type == "GCN" else 0)
        fused_flops += 2 * hidden * hidden * (2 if gnn_type == "GraphSAGE" else 1)
        if gnn_type == "GIN":
            gemm += row + row
    return {"fused_bytes": fused_bytes, "fused_flops": fused_flops, "gemm_bytes": gemm}


def cpu_baseline(n_runs, tile, sd, seed0, connectivity="8-connected"):
    """The CPU oracle (vectorised numpy graph build + fp32 torch forward issuing torch_geometric's op sequence + scatter)
    on ONE tile of the headline workload per run: one untimed warm-up run, then the MEDIAN of ``n_runs`` timed runs
    (warm-up 1, median of >= 5), with the spread.  kind = "port"."""
    from bathymetric_gnn_amd import synthetic
    from oracle import gat_cpu, graph_cpu
    k = DEG[connectivity]

    def one(seed):
        d, m, _ = synthetic.synthetic_tile(tile, tile, seed, "V0")
        a = time.perf_counter()
        g = graph_cpu.build_graph(d, m, None, (0.5, 0.5), connectivity=connectivity)
        b = time.perf_counter()
        gat_cpu.process_tile(sd, g)
        c = time.perf_counter()
        return g.num_nodes, c - a, b - a

    one(seed0)                                                       # warm-up: thread pools, allocator, first-touch
    t0 = time.perf_counter()
    runs = [one(seed0 + 1 + i) for i in range(n_runs)]
    wall = time.perf_counter() - t0
    rates = sorted(n / t for n, t, _ in runs)
    med = float(np.median(rates))
    return {"value": med, "unit": "nodes/s", "cores": torch.get_num_threads(), "kind": "port",
            "min": rates[0], "max": rates[-1], "runs": n_runs,
            "sample_short": f"median of {n_runs} runs (1 warm-up) of one {tile}x{tile} tile (k={k}, fp32): numpy graph build + torch CPU "
                            f"forward + scatter, {wall:.0f} s; os.cpu_count()={os.cpu_count()}",
            "sample": f"median of {n_runs} runs (after 1 warm-up run) of one {tile}x{tile} tile each (k={k}, 4-layer GAT, fp32): "
                      f"numpy graph build + torch CPU forward + scatter; {wall:.1f} s wall for the timed runs "
                      f"({sum(r[2] for r in runs):.1f} s of it graph build); spread {rates[0]:.0f} .. {rates[-1]:.0f} nodes/s; "
                      f"os.cpu_count()={os.cpu_count()}"}


LINE_BUDGET = 4000           # bytes of the ONE stdout line (the driver keeps about 8 KB of stdout's tail)


def kernel_source_sha(csrc=None):
    """sha256 over the kernel sources (csrc/*.hip, *.h, sorted by name): what `bgnn_build_id()` of a library built from this tree
    returns, and what tools/summarise_profiles.py stamps into profiles/pmc_traffic.json."""
    import hashlib
    csrc = csrc or os.path.join(ROOT, "bathymetric-gnn_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")) and name != "build_id.hip":
            h.update(name.encode()); h.update(b"\0")
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic_table(build_id):
    """profiles/pmc_traffic.json (rocprofv3 --pmc passes, collected separately) -- or {} when it was collected on other
    kernels than the ones the loaded library was built from (then roofline.traffic stays null rather than going stale)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except (OSError, ValueError):
        return {}
    if not build_id or t.get("_kernel_source_sha") != build_id:
        return {}
    return t


def compact_line(line, budget=LINE_BUDGET):
    """The stdout line within `budget` bytes: optional keys are dropped (least important first) until it fits; the contract
    keys, roofline and cpu_baseline are never dropped."""
    out = dict(line)
    for k in ("opt_in_split", "matrix_path", "path", "generic_shape", "gnn_types", "pcie_inclusive", "single_tile", "gpu_over_cpu", "survey", "config5", "config4", "config3", "detail"):
        if len(json.dumps(out)) <= budget:
            break
        out.pop(k, None)
    if len(json.dumps(out)) > budget and isinstance(out.get("cpu_baseline"), dict):
        out["cpu_baseline"] = {k: v for k, v in out["cpu_baseline"].items() if k != "sample"} | {"sample": "see detail file"}
    if len(json.dumps(out)) > budget:
        out["config"] = {"workload": str(out["config"].get("workload", ""))[:200]}
    if len(json.dumps(out)) > budget:                    # last resort (cannot happen with the keys bench.py writes): the contract alone
        keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline")
        out = {k: (str(v)[:200] if isinstance(v, str) else v) for k, v in out.items() if k in keep}
    return out


def emit(line, detail, detail_path):
    """Full record (line + detail) to the side file and to stderr; ONE compact JSON line to stdout, last."""
    full = dict(line); full.update(detail)
    try:
        os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
        with open(detail_path, "w") as fh:
            json.dump(full, fh, indent=1)
        line["detail"] = os.path.relpath(detail_path, ROOT) if os.path.abspath(detail_path).startswith(ROOT) else detail_path
    except OSError as e:
        line["detail"] = f"not written ({e.__class__.__name__}); see stderr"
    print("bench detail: " + json.dumps(full), file=sys.stderr, flush=True)
    out = compact_line(line)
    s = json.dumps(out)
    if len(s) > LINE_BUDGET:                             # never lose the line over its size: say so on stderr and print it anyway
        print(f"bench.py: stdout line is {len(s)} bytes (budget {LINE_BUDGET})", file=sys.stderr, flush=True)
    sys.stdout.flush()
    print(s, flush=True)


# ---- launching ------------------------------------------------------------------------------------------------------------
def visible_gpu_count(env=None, sysfs="/sys/class/kfd/kfd/topology/nodes", dev_dir="/dev/dri"):
    """GPUs this process could open, counted WITHOUT touching the HIP runtime (the parent of a multi-rank run must not
    initialise it): KFD topology nodes with SIMDs (CPU nodes have simd_count 0) whose render node is accessible (a
    container's device cgroup hides the others), capped by ROCR_ / HIP_ / CUDA_VISIBLE_DEVICES when set."""
    env = os.environ if env is None else env
    n = 0
    try:
        nodes = sorted(os.listdir(sysfs))
    except OSError:
        nodes = []
    for node in nodes:
        try:
            props = dict(line.split(None, 1) for line in open(os.path.join(sysfs, node, "properties")) if " " in line.strip())
        except OSError:
            continue
        try:
            if int(props.get("simd_count", "0")) <= 0:
                continue
            minor = int(props.get("drm_render_minor", "-1"))
        except ValueError:
            continue
        if dev_dir and minor >= 0 and not os.access(os.path.join(dev_dir, f"renderD{minor}"), os.R_OK | os.W_OK):
            continue
        n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if var in env:
            n = min(n, len([x for x in env[var].split(",") if x.strip()]))
    return n


def launch_plan(gpus, env, n_devices, share_gpu=False):
    """What `bench.py --gpus N` does with its process: ("worker", None) when a launcher (torch.distributed.run, or this
    file's own spawn) already set WORLD_SIZE; ("single", None) for N = 1; ("spawn", None) when N > 1 ranks must be
    started from here; ("error", reason) when that cannot work.  Pure function (unit-tested on CPU).  ``n_devices`` may be
    a callable (only evaluated when ranks would have to be spawned)."""
    if gpus < 1:
        return "error", f"--gpus {gpus}: need at least one GPU"
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != gpus:
            return "error", f"--gpus {gpus} but WORLD_SIZE={world}"
        return ("worker" if world > 1 else "single"), None
    if gpus == 1:
        return "single", None
    if share_gpu:                # rehearsal: the N ranks all use device 0 (gloo transport), whatever the node has
        return "spawn", None
    if callable(n_devices):
        n_devices = n_devices()
    if n_devices < gpus:
        return "error", f"--gpus {gpus} but only {n_devices} GPU(s) visible on this node"
    return "spawn", None


def spawn_ranks(n, argv, env=None, python=None, poll_s=0.2):
    """Start `n` child processes of `argv` (one per GPU: RANK = LOCAL_RANK = i, WORLD_SIZE = n, rendezvous on
    127.0.0.1 at a free port) and return the first non-zero exit status (0 if all succeed).  All children are polled
    together: as soon as one rank fails the others -- which may sit in a barrier or a collective waiting for it -- are
    killed, so the command fails loudly instead of hanging.  The parent never touches the GPU, so nothing is exec'd or
    forked from a process that has initialised HIP."""
    env = dict(os.environ if env is None else env)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for i in range(n):
        e = dict(env, RANK=str(i), LOCAL_RANK=str(i), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        e.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))   # (torchrun sets 1: the ranks share the host's cores)
        procs.append(subprocess.Popen([python or sys.executable] + list(argv), env=e))
    rc = 0
    live = list(procs)
    while live and rc == 0:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r if r > 0 else 1
        if live and rc == 0:
            time.sleep(poll_s)
    for p in live:                                             # a rank failed: the survivors are fresh processes of ours
        p.kill()
    for p in live:
        p.wait()
    return rc


# ---- workloads ------------------------------------------------------------------------------------------------------------
class Bench:
    """Model, graph builders and engines of one rank; builds the workloads and times them with one protocol."""

    def __init__(self, dev, rank, world, layers, dist=None, hidden=64, heads=4):
        from bathymetric_gnn_amd import runtime as rt, synthetic
        self.rt, self.syn = rt, synthetic
        self.dev, self.rank, self.world, self.layers, self.dist = dev, rank, world, layers, dist
        self.hidden, self.heads = hidden, heads
        self._models = {}
        self.ctx = rt.get_context(dev)
        self.nn_dev = torch.zeros(1, dtype=torch.int64, device=dev)

    def model(self, in_ch, gnn_type="GAT"):
        from bathymetric_gnn_amd.models import BathymetricGNN
        key = (in_ch, gnn_type)
        if key not in self._models:
            sd = self.syn.synthetic_state_dict(in_channels=in_ch, num_layers=self.layers, seed=1234, gnn_type=gnn_type,
                                               hidden=self.hidden, heads=self.heads)
            m = BathymetricGNN(in_channels=in_ch, num_gnn_layers=self.layers, gnn_type=gnn_type, edge_dim=3, dropout=0.0,
                               hidden_channels=self.hidden, heads=self.heads)
            m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
            self._models[key] = (m.to(self.dev).eval(), sd)
        return self._models[key]

    def ctx_option(self, name):
        try:
            return int(self.ctx.get_option(name))
        except Exception:
            return 0

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize(self.dev)

    # -- uniform tile batches: the headline (k = 8, f32) and configs[2] (k = 16, bf16 storage) ---------------------------
    def tiles(self, B, S, variant, conn, matrix_path=None, unfused=False, gnn_type="GAT"):
        from bathymetric_gnn_amd.data import GraphBuilder
        from bathymetric_gnn_amd.models.pipeline import TileBatchEngine
        model, sd = self.model(7, gnn_type)
        gb = GraphBuilder(connectivity=conn, device=self.dev)
        eng = TileBatchEngine(model, gb, self.dev)
        ctx = eng.ctx
        n_distinct = min(B, 8)
        depth, mask, _ = self.syn.synthetic_tile_batch(n_distinct, S, S, 100 + 1000 * self.rank, variant)
        reps = (B + n_distinct - 1) // n_distinct
        depth = np.concatenate([depth] * reps)[:B]; mask = np.concatenate([mask] * reps)[:B]
        d_t = torch.from_numpy(depth).to(self.dev).reshape(-1)
        m_t = torch.from_numpy(mask.view(np.uint8)).to(self.dev).reshape(-1)
        hw = np.tile(np.array([[S, S]], np.int32), (B, 1)); res = np.full((B, 2), 0.5)
        out = torch.empty((3, d_t.numel()), dtype=torch.float32, device=self.dev)
        opts = {}
        if matrix_path:
            opts["matrix_path"] = matrix_path
        if unfused:
            opts["fused"] = 0
        bf16 = matrix_path == "bf16"
        nodes = int(mask.sum())

        def step():
            eng.infer_device(hw, res, d_t, m_t, None, out=out, n_nodes_out=self.nn_dev)

        def check():
            assert int(self.nn_dev.item()) == nodes

        return {"kind": "tiles", "step": step, "check": check, "nodes_per_step": nodes, "contexts": [ctx], "options": opts,
                "deg": DEG[conn], "conn": conn, "bf16": bf16, "split": matrix_path if matrix_path in ("bf16x3", "fp16x3") else None,
                "unfused": unfused, "B": B, "S": S, "eng": eng, "gb": gb, "host": (depth, mask), "dev_in": (d_t, m_t),
                "events_in_timed_region": True, "scaling": "weak", "gnn_type": gnn_type,
                "name": (f"{B} tiles of {S}x{S} per GPU per step, {conn} (k={DEG[conn]}), {self.layers}-layer {gnn_type} "
                         f"(hidden {self.hidden}{f', heads {self.heads}' if gnn_type == 'GAT' else ''}), mask {variant}, inputs resident in HBM"
                         + (", layer activations stored as bf16 (bf16 MFMA, f32 softmax / aggregation / accumulation)" if bf16 else ""))}

    # -- configs[3]: ragged refinement grids, packed greedily in stream order until the node budget is reached -----------
    def vr(self, n_grids, budget, streams, grids=None):
        from bathymetric_gnn_amd.data import GraphBuilder
        from bathymetric_gnn_amd.models.pipeline import TileBatchEngine
        rt = self.rt
        model, sd = self.model(8)
        gb = GraphBuilder(connectivity="8-connected", device=self.dev)
        if grids is None:
            grids = self.syn.vr_grid_stream(n_grids, seed0=1000 + 100000 * self.rank)
        batches, cur, cur_nodes, nodes = [], [], 0, 0
        for d, u, r in grids:
            m = (d != self.syn.NODATA) & np.isfinite(d)
            cur.append((d, m, u, r)); cur_nodes += int(m.sum())
            if cur_nodes >= budget:
                batches.append(cur); nodes += cur_nodes; cur, cur_nodes = [], 0
        if cur:
            batches.append(cur); nodes += cur_nodes
        dev_batches = []
        for b in batches:
            hw_b, res_b, d_b, m_b, u_b = gb.upload_tiles([x[0] for x in b], [x[1] for x in b], [x[2] for x in b], [x[3] for x in b])
            dev_batches.append((hw_b, res_b, d_b, m_b, u_b, torch.empty((3, d_b.numel()), dtype=torch.float32, device=self.dev)))
        streams = max(1, streams)
        engines = [TileBatchEngine(model, gb, self.dev)]
        extra = [rt.new_context(self.dev) for _ in range(streams - 1)]
        engines += [TileBatchEngine(model, gb, self.dev, ctx=c) for c in extra]

        def step():
            for i, (hw_b, res_b, d_b, m_b, u_b, o_b) in enumerate(dev_batches):
                engines[i % len(engines)].infer_device(hw_b, res_b, d_b, m_b, u_b, out=o_b, defer_end=True)
            for e in engines:
                e.ctx.end()

        def close():
            for c in extra:
                c.close()

        return {"kind": "vr", "step": step, "check": lambda: None, "close": close, "nodes_per_step": nodes,
                "contexts": [e.ctx for e in engines], "options": {}, "deg": 8, "conn": "8-connected", "bf16": False, "split": None,
                "unfused": False, "n_grids": len(grids), "batches": len(batches), "budget": budget, "streams": streams,
                "events_in_timed_region": False, "scaling": "weak",
                "name": (f"{len(grids)} ragged refinement grids (3x3..50x50, in=8) per GPU per step in {len(batches)} batches "
                         f"of >= {budget} nodes on {streams} library context(s) / HIP stream(s), 8-connected, "
                         f"{self.layers}-layer GAT, inputs resident in HBM")}

    def vr_processor_api(self, budget, base=28, full_base=70):
        """The drop-in API itself (scripts/inference_native.py:445-538 = run_refinements) on a synthetic VR BAG held in HOST arrays:
        H2D of the records, D2H of the corrected ones and the write-back are inside the clock -- everything a caller pays.
          synchronous    : the reference's control flow, one flush_batch per full batch
          pipelined      : run_refinements' default; for a VRBagHandler + VRBagWriter that is the records-resident whole-BAG path
                           (NativeVRProcessor.process_refinements: no per-grid Python, chunks in flight on two contexts)
          pipelined_loop : the grid-by-grid loop with two coalesced submissions in flight (what a foreign handler / writer gets)
          whole_bag      : `pipelined` on a BAG of configs[3]'s size (>= 4096 refinement grids)."""
        from bathymetric_gnn_amd.data import GraphBuilder, VRBagHandler
        from bathymetric_gnn_amd.scripts.inference_native import NativeVRProcessor, run_refinements
        model, _ = self.model(8)
        proc = NativeVRProcessor(model, GraphBuilder(device=self.dev), self.dev)
        proc.BATCH_NODE_BUDGET = budget

        def best_of(h, reps, **kw):
            run_refinements(proc, h, h.copy_and_open_for_writing(), 0.0, **kw)      # warm-up (second context, arenas, pinned slabs)
            best, st = None, None
            for _ in range(reps):                         # best of `reps`: a few-ms measurement on a shared host
                writer = h.copy_and_open_for_writing()    # (the copy of the output BAG is file I/O in the reference: before the clock)
                torch.cuda.synchronize(self.dev)
                t0 = time.perf_counter()
                st = run_refinements(proc, h, writer, 0.0, **kw)
                torch.cuda.synchronize(self.dev)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            return {"value": st["cells_processed"] / best, "unit": "nodes/s", "wall_s": best, "nodes": st["cells_processed"],
                    "grids": st["grids_processed"]}

        h = VRBagHandler.from_arrays(*self.syn.synthetic_vr_bag(base, base, seed=4242))
        out = {"sample": f"synthetic VR BAG, {base}x{base} base cells, {h.num_refinement_cells} refinement grids (3x3..50x50), host arrays in / "
                         f"corrected records out, {budget}-node reference batches; best of 3 runs, output writer opened before the clock"}
        out["synchronous"] = best_of(h, 3, pipelined=False)
        out["pipelined"] = best_of(h, 3)
        out["pipelined_loop"] = best_of(h, 3, records_resident=False)
        hf = VRBagHandler.from_arrays(*self.syn.synthetic_vr_bag(full_base, full_base, seed=4242))
        out["whole_bag"] = best_of(hf, 3)
        out["whole_bag"]["sample"] = (f"synthetic VR BAG, {full_base}x{full_base} base cells, {hf.num_refinement_cells} refinement grids, "
                                      f"{hf.total_refinement_nodes} cells: run_refinements(handler, writer) on host arrays, copies inside the clock")
        for e in proc._engines[1:]:
            e.ctx.close()
        return out

    # -- configs[4]: a survey resident in HBM, cut into overlapping tiles, classified, stitched on the device --------------
    def survey(self, size, tile=512, overlap=128, tile_batch=32):
        from bathymetric_gnn_amd.config import Config
        from bathymetric_gnn_amd.models import BathymetricPipeline
        from bathymetric_gnn_amd.models.pipeline import survey_shard_plan
        model, sd = self.model(7)
        cfg = Config(); cfg.tile.tile_size, cfg.tile.overlap = tile, overlap
        pipe = BathymetricPipeline(cfg, tile_batch=tile_batch)
        pipe.set_model(model)
        S = int(size)
        rank, world = self.rank, self.world
        if world == 1:
            depth, valid = self.syn.synthetic_survey_device(S, self.dev, seed=0)
            lo, hi, shard = 0, S, None
        else:       # row bands: the rank generates (procedurally, by absolute row) only the rows its own tile rows span
            (lo, hi), plan = pipe.survey_rows_of_rank((S, S), rank, world)
            depth, valid = self.syn.synthetic_survey_device(S, self.dev, seed=0, rows=(lo, hi))
            shard = (rank, world)
        state = {}

        def step():
            if hi <= lo:              # a rank without tile rows (more ranks than tile rows): nothing to classify, nothing to exchange
                return
            if shard is None:
                state["out"] = pipe.process_survey_device(depth, valid, None, (0.5, 0.5))
            else:
                state["out"] = pipe.process_survey_device(depth, valid, None, (0.5, 0.5), shard=shard, survey_shape=(S, S),
                                                          row_offset=lo)

        # tiles processed per step by this rank (the min_valid_ratio filter is data dependent): one untimed pass
        step()
        torch.cuda.synchronize(self.dev)
        n_proc, n_skip = pipe.last_tile_counts if hi > lo else (0, 0)
        ntr, ntc, _ = pipe.tile_manager.compute_tile_grid((S, S))
        halo = 0
        if world > 1:                # every halo tile row travels as [3][ntc][cells] results + ntc keep flags, float32
            halo = sum(len(p["need"]) for p in plan) * (3 * ntc * tile * tile + ntc) * 4
        def stitched_sha256():
            """sha256 over the stitched [4, H, W] result of the last step, on rank 0 (None elsewhere): the bands of a sharded run
            are gathered to rank 0 first (gather_bands_to_rank0), so that 1 rank and N ranks can be compared bit for bit."""
            import hashlib
            from bathymetric_gnn_amd.models.pipeline import gather_bands_to_rank0
            if shard is None:
                host = state["out"].cpu().numpy()
            else:
                band = state["out"][2] if hi > lo else None
                host = gather_bands_to_rank0(plan, rank, band, S, device=self.dev)
                if host is None:
                    return None
            return hashlib.sha256(np.ascontiguousarray(host).tobytes()).hexdigest()

        return {"kind": "survey", "step": step, "check": lambda: None, "nodes_per_step": n_proc * tile * tile, "sha256": stitched_sha256,
                "contexts": [pipe._engine.ctx], "options": {}, "deg": 8, "conn": "8-connected", "bf16": False, "split": None,
                "unfused": False, "size": S, "tiles_total": ntr * ntc, "tiles_processed": n_proc, "tiles_skipped": n_skip,
                "halo_bytes_all_ranks": halo, "events_in_timed_region": True, "scaling": "strong" if world > 1 else "weak",
                "name": (f"{S}x{S} synthetic survey @0.5 m resident in HBM, {ntr * ntc} overlapping {tile}x{tile} tiles (overlap "
                         f"{overlap}) cut / classified in batches of {tile_batch} / stitched on the device "
                         f"(process_survey_device), 8-connected, {self.layers}-layer GAT; node evaluations = cells of processed tiles"
                         + (f"; ONE survey row-band sharded over {world} GPUs, halo tile rows point-to-point" if world > 1 else ""))}

    def survey_host_api(self, size, tile=512, overlap=128, tile_batch=32):
        """configs[4] through the drop-in API on HOST arrays: BathymetricPipeline.process_grid(BathymetricGrid) -- what the body of the
        reference's process() is between load and save (models/pipeline.py:163-211): a host depth grid in, the dict of five host
        result grids out; the upload of the survey and the download of the results are inside the clock."""
        from bathymetric_gnn_amd.config import Config
        from bathymetric_gnn_amd.data.grid import BathymetricGrid
        from bathymetric_gnn_amd.models import BathymetricPipeline
        model, _ = self.model(7)
        cfg = Config(); cfg.tile.tile_size, cfg.tile.overlap = tile, overlap
        pipe = BathymetricPipeline(cfg, tile_batch=tile_batch)
        pipe.set_model(model)
        S = int(size)
        depth, _ = self.syn.synthetic_survey_device(S, self.dev, seed=0)
        grid = BathymetricGrid(depth=depth.cpu().numpy(), resolution=(0.5, 0.5))
        del depth
        torch.cuda.empty_cache()
        walls = []
        for _ in range(2):                                # (the first call grows arenas / pinned staging)
            torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
            out = pipe.process_grid(grid)
            torch.cuda.synchronize(self.dev)
            walls.append(time.perf_counter() - t0)
            assert set(out) == {"cleaned_depth", "classification", "confidence", "correction", "valid_mask"} and out["classification"].shape == (S, S)
            del out
        n_proc, n_skip = pipe.last_tile_counts
        evals = n_proc * tile * tile
        return {"value": evals / walls[-1], "unit": "node evaluations/s", "wall_s": walls[-1], "first_call_wall_s": walls[0],
                "cells_per_s": S * S / walls[-1], "tiles_processed": n_proc, "tiles_skipped": n_skip,
                "sample": f"BathymetricPipeline.process_grid on a host BathymetricGrid of {S}x{S} float32 cells (resolution 0.5 m): host depth in, "
                          "five host result grids out, H2D / D2H inside the clock; second of two calls"}

    # -- one timing protocol for all of them ------------------------------------------------------------------------------------
    def measure(self, wl, steps, warmup):
        """W untimed steps, barrier + synchronize, K timed steps, barrier + synchronize; MAX over ranks.  HIP events around
        every kernel class are recorded inside the timed region -- except for the vr workload, which issues ~700 small launches
        per step: there the event pairs (1 400 records a step) are host work that slows the stream itself down, so its timed
        steps run bare and the same K steps are repeated with events for the kernel breakdown."""
        rt = self.rt
        ctxs = wl["contexts"]
        old = [{k: c.get_option(k) for k in wl["options"]} for c in ctxs]
        for c in ctxs:
            for k, v in wl["options"].items():
                c.set_option(k, v)
        try:
            for _ in range(warmup):
                wl["step"]()
            self.barrier()
            if wl["events_in_timed_region"]:
                for c in ctxs:
                    c.profile(rt.K_NAMES)
            t0 = time.perf_counter()
            for _ in range(steps):
                wl["step"]()
            self.barrier()
            elapsed = time.perf_counter() - t0
            if not wl["events_in_timed_region"]:
                for c in ctxs:
                    c.profile(rt.K_NAMES)
                for _ in range(steps):
                    wl["step"]()
                self.barrier()
            prof = {k: {"ms": 0.0, "launches": 0} for k in rt.K_NAMES}
            for c in ctxs:
                for k, v in c.profile_read().items():
                    prof[k]["ms"] += v["ms"]; prof[k]["launches"] += v["launches"]
                c.profile([])
            wl["check"]()
        finally:
            for c, o in zip(ctxs, old):
                for k, v in o.items():
                    c.set_option(k, v)
        nodes_all = wl["nodes_per_step"]
        if self.dist is not None:
            rdev = torch.device("cpu") if self.dist.get_backend() == "gloo" else self.dev      # (gloo: the --share-gpu rehearsal)
            t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
            n = torch.tensor([wl["nodes_per_step"]], dtype=torch.int64, device=rdev)
            self.dist.all_reduce(n, op=self.dist.ReduceOp.SUM)
            nodes_all = int(n.item())
        return {"elapsed": elapsed, "prof": prof, "steps": steps, "warmup": warmup, "nodes_all_ranks_per_step": nodes_all}

    def report(self, wl, m):
        """value / ms_per_step / rooflines / kernels of one measured workload (rank 0's kernel events)."""
        steps, prof = m["steps"], m["prof"]
        af = bool(wl["bf16"] and self.layers >= 2 and (self.hidden, self.heads) == (64, 4) and self.ctx_option("bf16_layer0_af")
                  and self.ctx_option("bf16_two_phase"))
        am = algorithmic_model(num_layers=self.layers, deg=wl["deg"], act_bytes=2 if wl["bf16"] else 4,
                               in_ch=8 if wl["kind"] == "vr" else 7, hidden=self.hidden, heads=self.heads, layer0_af=af)
        n_local = wl["nodes_per_step"] * steps
        kernels = {k: {"ms_per_step": v["ms"] / steps, "launches_per_step": v["launches"] / steps} for k, v in prof.items() if v["launches"]}

        def roof(kernel, key, bound, work_per_node, note):
            t = prof[key]["ms"] / 1e3
            if t <= 0 or n_local <= 0:
                return None
            unit, peak, scale = ("GB/s", HBM_PEAK_GBS, 1e9) if bound == "hbm" else ("TFLOP/s", MFMA_F32_PEAK_TFLOPS, 1e12)
            if bound == "mfma_bf16":
                unit, peak, scale, bound = "TFLOP/s", MFMA_BF16_PEAK_TFLOPS, 1e12, "mfma"
            ach = work_per_node * n_local / t / scale
            return {"kernel": kernel, "bound": bound, "unit": unit, "peak": peak, "achieved": ach, "frac": ach / peak,
                    "avg_launch_ms": prof[key]["ms"] / max(prof[key]["launches"], 1),
                    "launches_per_step": prof[key]["launches"] / steps,
                    "algorithmic_work_per_node_per_forward": work_per_node, "traffic": None, "note": note}

        bf16, split = wl["bf16"], wl["split"]
        roofs = {}
        if wl.get("gnn_type", "GAT") != "GAT":
            pm = plain_backbone_model(wl["gnn_type"], num_layers=self.layers, deg=wl["deg"])
            roofs["fused_hbm"] = roof("gat_layer_fused_kernel", "fused", "hbm", pm["fused_bytes"],
                                      f"{wl['gnn_type']} layers, one fused launch each (aggregate -> GEMM -> post-op): h row in, h row out")
            roofs["fused_mfma"] = roof("gat_layer_fused_kernel", "fused", "mfma", pm["fused_flops"], "same launches priced by the layer GEMM's exact-f32 flops")
            roofs["gemm_hbm"] = roof("gemm_f32_kernel", "gemm", "hbm", pm["gemm_bytes"],
                                     "the plain GEMM launches (extractor, heads' first layers, GIN's second Linear; K = 64: below the f32 MFMA's "
                                     "flop/B balance), priced by compulsory bytes")
        elif prof["fused"]["launches"]:
            if bf16:
                roofs["fused_mfma"] = roof("gat_layer_fused_kernel", "fused", "mfma_bf16", am["fused_flops"],
                                           "bf16 MFMA, f32 accumulate: priced against the dense bf16 MFMA peak (the kernel is not matrix-bound)")
            elif split:
                roofs["fused_mfma"] = roof("gat_layer_fused_kernel", "fused", "mfma_bf16", 3 * am["fused_flops"],
                                           f"{split}: executed flops = 3 x algorithmic, priced against the dense bf16 / f16 MFMA peak")
            else:
                roofs["fused_mfma"] = roof("gat_layer_fused_kernel", "fused", "mfma", am["fused_flops"],
                                           "K4 gather-softmax-aggregate fused with the next layer's exact-f32 MFMA GEMM (last: heads + scatter)")
            roofs["fused_hbm"] = roof("gat_layer_fused_kernel", "fused", "hbm", am["fused_bytes"],
                                      "same launches priced by compulsory HBM bytes (read xW + attrs, write next xW)")
            if bf16 or split:
                # bf16 / 16-bit split MFMAs: the front GEMM's matrix work is negligible against that pipe -- it is a streaming
                # kernel (32-byte feature row in, xW_0 + attention dots out) and is priced by its bytes
                roofs["front_gemm"] = roof("extractor_af_kernel" if af else "gemm_wres64_kernel", "gemm", "hbm", am["front_bytes"],
                                           "extractor layer 1 + layer 0's attention dots (aggregate-first: feature row in, h1 as bf16 + dots out)" if af else
                                           "front GEMM on 16-bit MFMA: HBM-bound, priced by compulsory bytes (feature row in, xW_0 + attention dots out)")
            else:
                roofs["front_gemm"] = roof("gemm_wres64_kernel", "gemm", "mfma", am["front_flops"],
                                           "feature extractor layer 1 in front of (extractor layer 2 folded into) lin of layer 0, one launch (executed flops, exact f32 MFMA)")
                roofs["front_gemm_hbm"] = roof("gemm_wres64_kernel", "gemm", "hbm", am["front_bytes"], "same launch priced by compulsory HBM bytes")
            if prof["features"]["launches"]:
                roofs["features_hbm"] = roof("features_kernel", "features", "hbm", am["build_bytes"],
                                             "K1b + K2: node features, stencil table, edge attributes (native-internal graph form)")
        else:
            roofs["aggregate_hbm"] = roof("gat_aggregate_tiled_kernel", "aggregate", "hbm", am["aggregate_bytes"],
                                          "standalone K4 (LDS-tiled gather-softmax-aggregate + BN + ReLU)")
            roofs["gemm_mfma"] = roof("gemm_f32_kernel", "gemm", "mfma", am["gemm_flops"], "all K3 GEMMs, exact f32 MFMA")
        roofs = {k: v for k, v in roofs.items() if v}
        # PMC traffic is collected in separate rocprofv3 --pmc passes (tools/collect_profiles.sh -> profiles/pmc_traffic.json, stamped
        # with the kernel-source hash it was collected on); attached only to the workload AND the kernels it was collected on
        pmc_key = None
        # (the GAT model only -- a plain backbone runs other instances of the same kernel classes -- and of the operand-split paths
        #  only bf16x3, the one the counter passes ran)
        if wl["kind"] == "tiles" and (wl["B"], wl["S"], self.layers) == (128, 256, 4) and wl.get("gnn_type", "GAT") == "GAT":
            pmc_key = (":c3" if (wl["deg"] == 16 and bf16) else None if (wl["deg"] != 8 or bf16) else
                       ":split" if split == "bf16x3" else None if split else "")
        t = pmc_traffic_table(self.rt.build_id()) if pmc_key is not None else {}
        for v in roofs.values():
            e = t.get(v["kernel"] + pmc_key) if pmc_key else t.get(v["kernel"])
            if not e:
                continue
            v["traffic"] = e.get("hbm_bytes_per_launch")
            if v["traffic"] is not None:
                # HBM bytes by the counters / the launch's own duration, against the 8 TB/s roof: the fraction of the roof the
                # kernel really moves (`frac` prices SURVEY 8(d)'s fixed algorithmic figure, which the compact edge storage undercuts)
                v["traffic_frac"] = v["traffic"] / (v["avg_launch_ms"] * 1e-3) / (HBM_PEAK_GBS * 1e9)
                v["traffic_source"] = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on "
                                       f"kernels {t.get('_kernel_source_sha')} (= this library), collected separately, not by this run")
        if wl.get("gnn_type", "GAT") != "GAT" and roofs:
            # the class that takes the most time; for the fused launches the roof they sit closest to
            dominant = max(roofs.values(), key=lambda v: (v["avg_launch_ms"] * v["launches_per_step"], v["frac"]))
        elif "fused_mfma" in roofs:
            dominant = roofs["fused_hbm"] if (split or bf16) else roofs["fused_mfma"]   # 16-bit MFMA paths: memory-side bound
        elif roofs:
            dominant = max(roofs.values(), key=lambda v: v["avg_launch_ms"] * v["launches_per_step"])
        else:
            dominant = {}
        for v in roofs.values():
            assert v["frac"] <= 1.0, f"roofline fraction above 1 for {v['kernel']}: mis-priced"
        nodes_all = m["nodes_all_ranks_per_step"]
        return {"value": nodes_all * steps / m["elapsed"], "unit": "nodes/s", "ms_per_step": m["elapsed"] / steps * 1e3,
                "steps": steps, "warmup": m["warmup"], "workload": wl["name"], "nodes_per_step_per_gpu": wl["nodes_per_step"],
                "roofline": {k: dominant.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_frac",
                                                           "kernel", "avg_launch_ms")},
                "rooflines": roofs, "kernels": kernels,
                "kernel_events": "inside the timed region" if wl["events_in_timed_region"] else "separate pass of the same steps (timed steps ran bare)"}


def dtype_name(wl):
    if wl["bf16"]:
        return "bf16 (layer activations stored as bf16, bf16 MFMA; f32 softmax / aggregation / accumulation / outputs)"
    if wl["split"]:
        return f"f32 ({wl['split']} split-operand MFMA, f32 accumulate)"
    return "f32"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tiles", type=int, default=128, help="tiles per batch per GPU")
    ap.add_argument("--tile-size", type=int, default=256)
    ap.add_argument("--variant", default="V0", choices=["V0", "V1"])
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--hidden", type=int, default=64, help="gnn_hidden_channels (config/config.py:43): 64 is the fused kernels' shape; "
                    "32 / 128 run the generic GEMM + aggregate kernels, anything else zero-padded to the next of those")
    ap.add_argument("--heads", type=int, default=4, help="gnn_heads (config/config.py:45): 1 / 2 / 4 fused (hidden 64), otherwise generic kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed steps (no config3 / config4 / config5 / PCIe-inclusive / split-path / single-tile side "
                         "measurements, no CPU baseline): what the rocprofv3 passes run, so that a kernel's average duration in the "
                         "trace is the full-batch one")
    ap.add_argument("--split-f16", action="store_true",
                    help="opt-in matrix path: float16 hi/lo operand split (fp16x3), float32 accumulation (|activations| < 65504)")
    ap.add_argument("--split-bf16", action="store_true",
                    help="opt-in matrix path: bf16 hi/lo operand split (bf16x3) with float32 accumulation instead of exact-f32 MFMAs")
    ap.add_argument("--unfused", action="store_true",
                    help="run K3 / K4 / K5 / K6 as separate kernels (standalone gather-aggregate roofline)")
    ap.add_argument("--cpu-runs", type=int, default=5, help="CPU baseline: timed single-tile runs (median reported)")
    ap.add_argument("--workload", default="tiles", choices=["tiles", "vr", "c3", "survey"],
                    help="tiles: B uniform tiles per step (headline). c3: BASELINE configs[2] -- B tiles of 256x256 with k=16 (the "
                         "'16-dilated' stencil, a build-side extension) and bf16 node features. vr: configs[3] -- 4096 ragged refinement "
                         "grids packed by the reference's 50 000-node batch budget. survey: configs[4] -- one survey resident in HBM, "
                         "overlapping 512x512 tiles cut, classified and stitched on the device (row-band sharded under --gpus N)")
    ap.add_argument("--connectivity", default=None, choices=["4-connected", "8-connected", "16-dilated"])
    ap.add_argument("--gnn-type", default="GAT", choices=["GAT", "GCN", "GraphSAGE", "GIN"],
                    help="tiles workload: the backbone (reference models/gnn.py:120-143).  GAT is the hot path (fused kernels); the other "
                         "three run on plain gather + GEMM kernels and are priced by their compulsory HBM bytes")
    ap.add_argument("--bf16", action="store_true", help="matrix_path = bf16 (bf16 activation storage + bf16 MFMA)")
    ap.add_argument("--vr-grids", type=int, default=4096)
    ap.add_argument("--vr-budget", type=int, default=50000)
    ap.add_argument("--vr-streams", type=int, default=2,
                    help="vr workload: library contexts (HIP streams) the batches are dealt over: the tail of one 50 000-node "
                         "batch's kernels overlaps the head of the next one's (two is what NativeVRProcessor keeps in flight, "
                         "and the fastest: 203 M nodes/s against 193 M with four)")
    ap.add_argument("--survey-size", type=int, default=20000, help="survey workload: side of the square survey in cells (config 5: 60000)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="REHEARSAL of the multi-rank path on a one-GPU box: with --gpus N the N ranks all use device 0 and talk over "
                         "gloo (host-staged halo rows); the line says `rehearsal` and is no scaling measurement")
    ap.add_argument("--checksum", action="store_true",
                    help="survey workload: add the sha256 of the stitched result (rank 0; the bands of a sharded run gathered first)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="side file for the full record (rooflines of every kernel class, per-kernel times, the side measurements "
                         "in full); stdout carries ONE compact line")
    ap.add_argument("--extras-survey-size", type=int, default=20000, help="side of the config5 survey measured beside the default headline")
    args = ap.parse_args()

    # N > 1 without a launcher: start the N ranks here, BEFORE anything touches the GPU (the device count comes from the KFD
    # topology in sysfs, not from the HIP runtime), wait for them and leave with their status.
    mode, why = launch_plan(args.gpus, os.environ, visible_gpu_count, share_gpu=args.share_gpu)
    if mode == "error":
        raise SystemExit(f"bench.py: {why}")
    if mode == "spawn":
        raise SystemExit(spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.share_gpu:
            dist.init_process_group("gloo")                # rehearsal: RCCL cannot put two ranks on one device
        else:
            dist.init_process_group("nccl", device_id=dev)     # RCCL: the timing barrier / max; the survey workload's halo rows

    bench = Bench(dev, rank, world, args.layers, dist, hidden=args.hidden, heads=args.heads)
    S, B = args.tile_size, args.tiles
    mp = "bf16" if (args.bf16 or args.workload == "c3") else "fp16x3" if args.split_f16 else "bf16x3" if args.split_bf16 else None
    if args.workload in ("tiles", "c3"):
        conn = args.connectivity or ("16-dilated" if args.workload == "c3" else "8-connected")
        if args.gnn_type != "GAT" and (mp or args.unfused or args.workload == "c3"):
            raise SystemExit("bench.py: --gnn-type other than GAT runs the plain exact-f32 kernels only")
        wl = bench.tiles(B, S, args.variant, conn, matrix_path=mp, unfused=args.unfused, gnn_type=args.gnn_type)
    elif args.workload == "vr":
        wl = bench.vr(args.vr_grids, args.vr_budget, args.vr_streams)
    else:
        wl = bench.survey(args.survey_size)
    m = bench.measure(wl, args.steps, args.warmup)
    sha = wl["sha256"]() if (args.checksum and "sha256" in wl) else None          # (every rank: the band gather is point-to-point)

    if rank == 0:
        rep = bench.report(wl, m)
        line = {
            "metric": ((f"classified tile-nodes/s (graph build + 4-layer {wl.get('gnn_type')} forward + scatter; fused aggregate-GEMM layer launches)"
                        if wl.get("gnn_type", "GAT") != "GAT" else
                        "classified tile-nodes/s (fused graph build + 4-layer GAT forward + scatter)") if wl["kind"] != "survey" else
                       "classified tile-nodes/s (node evaluations of the survey's overlapping tiles: cut + graph build + GAT forward + stitch)"),
            "value": rep["value"], "unit": "nodes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": rep["ms_per_step"], "higher_is_better": True, "scaling": wl["scaling"],
            "vs_baseline": None, "dtype": dtype_name(wl), "data": "synthetic",
            "config": {"workload": wl["name"],
                       "tiles_per_gpu": B if wl["kind"] == "tiles" else wl.get("n_grids", wl.get("tiles_processed")),
                       "tile": S if wl["kind"] == "tiles" else "3..50" if wl["kind"] == "vr" else 512,
                       "nodes_per_step_per_gpu": wl["nodes_per_step"],
                       "parallelism": (f"tile-sharded x{world}, no collective" if wl["kind"] != "survey" else
                                       f"row bands x{world}, halo tile rows point-to-point ({wl['halo_bytes_all_ranks'] / 1e9:.2f} GB per step), no collective")},
            "roofline": rep["roofline"],
            "path": "unfused" if wl["unfused"] else "fused",
            "matrix_path": "bf16 storage + bf16 MFMA (BASELINE configs[2])" if wl["bf16"] else f"{wl['split']} split (opt-in)" if wl["split"] else "exact f32",
        }
        # Everything beyond the contract goes to the DETAIL record (side file + stderr): the stdout line stays small enough for
        # the driver's tail of stdout (round 3's 23 KB line lost its head there).
        detail = {"rooflines": rep["rooflines"], "kernels": rep["kernels"], "kernel_events": rep["kernel_events"]}
        if wl["kind"] == "survey":
            line["survey"] = {k: wl[k] for k in ("size", "tiles_total", "tiles_processed", "tiles_skipped", "halo_bytes_all_ranks")}
            if sha is not None:
                line["survey"]["stitched_sha256"] = sha
        if args.share_gpu and world > 1:
            line["rehearsal"] = True
            line["rehearsal_note"] = (f"{world} ranks share ONE GPU and talk over gloo (host-staged halo rows): a functional rehearsal of "
                                      "the multi-rank code path, NOT a scaling measurement -- `value` earns no credit")
        extras = not args.no_extras and world == 1
        plain = wl.get("gnn_type", "GAT") != "GAT"
        default_headline = wl["kind"] == "tiles" and not wl["unfused"] and not wl["bf16"] and not wl["split"] and not plain
        if plain:
            line["path"], line["matrix_path"] = "fused layers (plain-backbone mode)", "exact f32"

        def guarded(key, fn):
            """One side measurement: its failure is recorded under its key, the headline still prints."""
            try:
                return fn()
            except Exception as e:                                   # noqa: BLE001 -- any failure of an extra must not lose the line
                import traceback
                traceback.print_exc(file=sys.stderr)
                detail[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
                line[key] = {"error": f"{type(e).__name__}: {e}"[:120]}
                torch.cuda.empty_cache()
                return None

        # The CPU baseline first: it is part of the contract and needs nothing of the GPU.
        if extras and not args.no_cpu_baseline and wl["kind"] == "tiles" and not plain:
            def _cpu():
                cb = cpu_baseline(args.cpu_runs, S, bench.model(7)[1], 100, wl["conn"])
                detail["cpu_baseline"] = cb
                line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind")}
                line["cpu_baseline"]["sample"] = cb["sample_short"]
                line["gpu_over_cpu"] = rep["value"] / cb["value"]
            guarded("cpu_baseline", _cpu)
        if extras and wl["kind"] == "tiles" and not wl["unfused"] and not plain:
            def _pcie_and_single():
                eng, (depth, mask), (d_t, m_t) = wl["eng"], wl["host"], wl["dev_in"]
                ctx = eng.ctx
                with ctx.options(**wl["options"]):
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
                    detail["pcie_inclusive"] = {"value": wl["nodes_per_step"] * n_pc / t_pc, "unit": "nodes/s", "ms_per_step": t_pc / n_pc * 1e3,
                                                "steps": n_pc, "bytes_per_cell": {"h2d": 5, "d2h": 12},
                                                "note": "host numpy tiles in, host grids out: pinned double-buffered staging, H2D / "
                                                        "compute / D2H overlapped on three streams (HostTilePipeline)"}
                    line["pcie_inclusive"] = {"value": detail["pcie_inclusive"]["value"], "ms_per_step": detail["pcie_inclusive"]["ms_per_step"]}
                    del hp
                    if B > 1:
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
                        detail["single_tile"] = {"value": int(mask[0].sum()) / t_one, "unit": "nodes/s", "ms_per_tile": t_one * 1e3, "steps": n_one,
                                                 "note": f"configs[1]: one {S}x{S} tile per step (back-to-back launches, inputs resident in HBM)"}
                        line["single_tile"] = {"value": detail["single_tile"]["value"], "ms_per_tile": t_one * 1e3}
            guarded("pcie_inclusive", _pcie_and_single)
        if extras and default_headline:
            # Opt-in matrix paths, reported BESIDE the headline (never as it): hi/lo operand splits on the 16-bit matrix cores with
            # float32 accumulation.  Same inputs, same timing protocol; the distance of their class logits to the exact-f32 path
            # is measured on one tile of the batch.
            def _splits():
                from bathymetric_gnn_amd.data import GraphBuilder as _GB
                model = bench.model(7)[0]
                ctx = wl["eng"].ctx
                g1 = _GB(device=dev).build_graph(wl["host"][0][0], wl["host"][1][0], None, (0.5, 0.5))
                lg_exact = model.predict(g1)["class_logits"].clone()
                for key, env, instr in (("split_bf16x3", "bf16x3", "v_mfma_f32_32x32x16_bf16"),
                                        ("split_fp16x3", "fp16x3", "v_mfma_f32_32x32x16_f16")):
                    ctx.set_option("matrix_path", env)
                    try:
                        lg_split = model.predict(g1)["class_logits"]
                        for _ in range(2):
                            wl["step"]()
                        n_sp = max(4, min(args.steps, 10))
                        torch.cuda.synchronize(dev); t2 = time.perf_counter()
                        for _ in range(n_sp):
                            wl["step"]()
                        torch.cuda.synchronize(dev); t_sp = time.perf_counter() - t2
                    finally:
                        ctx.set_option("matrix_path", "exact_f32")
                    detail[key] = {"value": wl["nodes_per_step"] * n_sp / t_sp, "unit": "nodes/s", "ms_per_step": t_sp / n_sp * 1e3,
                                   "steps": n_sp, "max_abs_logit_diff_vs_exact_f32": float((lg_split - lg_exact).abs().max().item()),
                                   "note": f"matrix_path={env}: layer GEMMs as hi/lo operand splits on {instr}, float32 accumulate; "
                                           "opt-in, not the headline"}
                    # (brief, in the line: what the opt-in operand-split paths give and how far their logits are from the headline path's)
                    line.setdefault("opt_in_split", {"note": "not the headline; max |dlogit| vs the exact-f32 path"})[env] = {
                        "value": detail[key]["value"], "max_dlogit": detail[key]["max_abs_logit_diff_vs_exact_f32"]}
                # distance to the float64 forward: measured by the -m gpu suite against the oracle (test_matrix_paths_distance_to_float64),
                # not by this run -- the committed record of that test is quoted (max over its two tiles)
                rec = os.path.join(ROOT, "profiles", "split_accuracy.json")
                if os.path.exists(rec) and "opt_in_split" in line:
                    acc = json.load(open(rec))
                    worst = lambda k: max(float(v[k]["max"]) for v in acc.values() if isinstance(v, dict) and k in v)
                    line["opt_in_split"]["max_dist_to_float64"] = {"exact_f32": worst("exact_f32_mfma"), "fp16x3": worst("fp16x3"), "bf16x3": worst("bf16x3"),
                                                                  "source": "profiles/split_accuracy.json (GPU test suite vs the float64 oracle)"}
            guarded("split_paths", _splits)
            # ---- the other single-GPU BASELINE configs, same protocol (rank 0, N = 1), each with its own roofline ----------
            k_x, w_x = max(4, min(args.steps, 10)), 2

            def side(w, steps=k_x, warmup=w_x, extra=None):
                try:
                    r = bench.report(w, bench.measure(w, steps, warmup))
                finally:
                    if "close" in w:
                        w["close"]()
                r["dtype"] = dtype_name(w)
                r["higher_is_better"] = True
                if extra:
                    r.update(extra)
                return r

            def brief(r, short_dtype, workload):
                """What the stdout line keeps of a side measurement."""
                rf = r["roofline"]
                return {"value": r["value"], "ms_per_step": r["ms_per_step"], "dtype": short_dtype, "workload": workload,
                        "roofline": {"bound": rf.get("bound"), "frac": rf.get("frac"), "kernel": rf.get("kernel")}}

            def _c3():
                c3 = bench.tiles(128, 256, "V0", "16-dilated", matrix_path="bf16")
                detail["config3"] = side(c3, extra={"config": "BASELINE configs[2]: batch of 128 x 256x256 tiles, k=16, bf16 node features"})
                line["config3"] = brief(detail["config3"], "bf16", "configs[2]: 128 x 256x256 tiles, k=16 (16-dilated), bf16 activations")
            guarded("config3", _c3)
            torch.cuda.empty_cache()

            def _c4():
                grids = bench.syn.vr_grid_stream(args.vr_grids, seed0=1000)
                v1 = side(bench.vr(args.vr_grids, args.vr_budget, 1, grids=grids))
                v2 = side(bench.vr(args.vr_grids, args.vr_budget, 2, grids=grids))
                v4 = side(bench.vr(args.vr_grids, args.vr_budget, 4, grids=grids))
                detail["config4"] = {"config": "BASELINE configs[3]: VR-BAG mixed refinement grids (3x3..50x50), 4096-grid stream, "
                                               f"{args.vr_budget}-node batches (scripts/inference_native.py:128)",
                                     "value": v1["value"], "unit": "nodes/s", "ms_per_step": v1["ms_per_step"], "roofline": v1["roofline"],
                                     "note": "value = ONE library context / HIP stream: what one synchronous NativeVRProcessor.flush_batch after the "
                                             "other gives; two_contexts = two batches in flight, what the processor's submit_batch / collect_batch "
                                             "(run_refinements' default) keeps on the GPU; four_contexts = the same batches dealt over 4 contexts",
                                     "one_context": v1, "two_contexts": v2, "four_contexts": v4}
                line["config4"] = brief(v1, "f32", f"configs[3]: {args.vr_grids} VR refinement grids (3x3..50x50), {args.vr_budget}-node batches, one context")
                line["config4"]["one_context"] = v1["value"]
                line["config4"]["two_contexts"] = v2["value"]
                api = bench.vr_processor_api(args.vr_budget)
                detail["config4"]["processor_api"] = api
                line["config4"]["processor_api"] = {k: api[k]["value"] for k in ("synchronous", "pipelined", "pipelined_loop")}
                line["config4"]["whole_bag"] = api["whole_bag"]["value"]
            guarded("config4", _c4)
            torch.cuda.empty_cache()

            def _generic_shape():
                # a model shape OUTSIDE the fused kernels' (hidden 64, heads <= 4): what leaving the fused path costs.  96 x 2 heads runs
                # zero-padded as 128 x 2 on the generic GEMM + LDS-tiled aggregate kernels.
                b2 = Bench(dev, rank, world, args.layers, dist, hidden=96, heads=2)
                w = b2.tiles(B, S, args.variant, "8-connected")
                r = b2.report(w, b2.measure(w, max(3, min(args.steps, 5)), 1))
                detail["generic_shape"] = dict(r, shape="hidden 96, heads 2 (zero-padded to 128 x 2; generic kernels)")
                line["generic_shape"] = {"hidden": 96, "heads": 2, "value": r["value"], "ms_per_step": r["ms_per_step"]}
                del w, b2
            if (args.hidden, args.heads) == (64, 4):
                guarded("generic_shape", _generic_shape)
                torch.cuda.empty_cache()

            def _backbones():
                detail["gnn_types"], line["gnn_types"] = {}, {}
                for gt in ("GCN", "GraphSAGE", "GIN"):
                    w = bench.tiles(B, S, args.variant, "8-connected", gnn_type=gt)
                    r = side(w, steps=max(3, min(args.steps, 5)), warmup=1)
                    detail["gnn_types"][gt] = r
                    line["gnn_types"][gt] = {"value": r["value"], "frac": r["roofline"].get("frac"), "kernel": r["roofline"].get("kernel")}
                    del w
                    torch.cuda.empty_cache()
            guarded("gnn_types", _backbones)

            def _c5():
                sv = bench.survey(args.extras_survey_size)
                detail["config5"] = side(sv, steps=1, warmup=0, extra={
                    "config": f"BASELINE configs[4] at {args.extras_survey_size}x{args.extras_survey_size} on ONE GPU (the full 60000x60000 survey runs "
                              "in tests/test_gpu_survey.py and with --workload survey --survey-size 60000)",
                    "survey": {k: sv[k] for k in ("size", "tiles_total", "tiles_processed", "tiles_skipped")}})
                line["config5"] = brief(detail["config5"], "f32",
                                        f"configs[4] at {args.extras_survey_size}^2 on one GPU: {sv['tiles_total']} overlapping 512x512 tiles, cut + classify + stitch on device")
            guarded("config5", _c5)
            torch.cuda.empty_cache()

            def _c5_host():
                r = bench.survey_host_api(args.extras_survey_size)
                detail.setdefault("config5", {})["host_api"] = r
                line.setdefault("config5", {})["host_api"] = r["value"]
            guarded("config5_host_api", _c5_host)
            torch.cuda.empty_cache()
        emit(line, detail, args.detail)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
