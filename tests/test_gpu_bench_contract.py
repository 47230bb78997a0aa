"""bench.py's ONE JSON line on a GPU, at small sizes: the driver's contract keys and a roofline whose fraction stays within (0, 1] --
for the tile, configs[2], VR and survey workloads (each as its own child process, as the driver runs it), and the DEFAULT command
shape (extras on: cpu_baseline, config3 / config4 / config5 at small sizes): exactly one stdout line, under 4 KB, that starts with
`{` -- round 3's default line was 23 KB and the driver's tail of stdout lost its head."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"]


def _line(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-500:]
    return json.loads(lines[0])


@pytest.mark.parametrize("args", [
    ["--tiles", "8", "--steps", "2", "--warmup", "1", "--no-extras"],
    ["--workload", "c3", "--tiles", "8", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"],
    ["--workload", "vr", "--vr-grids", "300", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"],
    ["--workload", "survey", "--survey-size", "2048", "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline"],
    ["--gnn-type", "GCN", "--tiles", "8", "--steps", "2", "--warmup", "1", "--no-extras"],
    ["--gnn-type", "GraphSAGE", "--tiles", "8", "--steps", "2", "--warmup", "1", "--no-extras"],
    ["--gnn-type", "GIN", "--tiles", "8", "--steps", "2", "--warmup", "1", "--no-extras"],
], ids=["tiles", "c3", "vr", "survey", "gcn", "sage", "gin"])
def test_bench_line_contract(args, gpu_device):
    j = _line(args)
    for k in CONTRACT:
        assert k in j, k
    assert j["n_gpus"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["value"] > 0 and j["ms_per_step"] > 0 and j["unit"] == "nodes/s" and j["data"].startswith("synthetic")
    assert "workload" in j["config"] and "model" not in j["config"]
    rf = j["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and 0 < rf["frac"] <= 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9


def test_default_command_shape_prints_one_small_line(gpu_device, tmp_path):
    """What the driver runs (`python bench.py --gpus 1 --steps K --warmup W`: every extra on), at small sizes."""
    side = tmp_path / "detail.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--tiles", "8",
                        "--vr-grids", "300", "--extras-survey-size", "2048", "--cpu-runs", "1", "--detail", str(side)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out_lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(out_lines) == 1 and out_lines[0].startswith("{"), r.stdout[-500:]
    assert len(out_lines[0]) < 4096
    j = json.loads(out_lines[0])
    for k in CONTRACT + ["cpu_baseline", "config3", "config4", "config5", "gnn_types"]:
        assert k in j, k
    assert set(j["gnn_types"]) == {"GCN", "GraphSAGE", "GIN"}
    for v in j["gnn_types"].values():
        assert v["value"] > 0 and 0 < v["frac"] <= 1.0 and v["kernel"] in ("gat_layer_fused_kernel", "gemm_f32_kernel")
    for k in ("config3", "config4", "config5"):
        assert "error" not in j[k], j[k]
        assert j[k]["value"] > 0 and j[k]["ms_per_step"] > 0 and j[k]["dtype"] in ("f32", "bf16")
        assert j[k]["roofline"]["bound"] in ("hbm", "mfma") and 0 < j[k]["roofline"]["frac"] <= 1.0
    assert j["config4"]["two_contexts"] > 0 and j["config4"]["processor_api"]["pipelined"] > 0
    cb = j["cpu_baseline"]
    assert cb["value"] > 0 and cb["kind"] == "port" and cb["cores"] >= 1 and cb["unit"] == "nodes/s" and cb["sample"]
    rf = j["roofline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] <= 1.0 and "traffic" in rf and rf["kernel"] == "gat_layer_fused_kernel"
    # the rest is in the side file (and on stderr), not on stdout
    full = json.load(open(side))
    assert "rooflines" in full and "kernels" in full and "one_context" in full["config4"] and "rooflines" in full["config3"]
    assert "bench detail: {" in r.stderr


def test_multi_rank_code_path_rehearsal_two_ranks_share_the_gpu(gpu_device):
    """The multi-rank path (`bench.py --gpus 2`: spawn_ranks, survey_rows_of_rank, the halo tile-row exchange, the band gather) on
    REAL kernels, with both ranks on this box's one GPU and gloo as transport (`--share-gpu`: a clearly labelled rehearsal -- RCCL
    cannot put two ranks on one device, and no 8-GPU node is available to builders).  No scaling claim: asserted are exit 0, one
    JSON line that says `rehearsal`, halo bytes > 0, and a stitched survey bit-identical to the 1-rank run's (sha256 of [4, H, W])."""
    base = ["--workload", "survey", "--survey-size", "2048", "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline", "--checksum"]
    one = _line(base)
    two = _line(base + ["--gpus", "2", "--share-gpu"])
    assert one["n_gpus"] == 1 and "rehearsal" not in one and one["survey"]["halo_bytes_all_ranks"] == 0
    assert two["n_gpus"] == 2 and two["rehearsal"] is True and "NOT a scaling measurement" in two["rehearsal_note"]
    assert two["scaling"] == "strong" and two["survey"]["halo_bytes_all_ranks"] > 0
    assert two["survey"]["tiles_total"] == one["survey"]["tiles_total"] == 25
    assert len(one["survey"]["stitched_sha256"]) == 64
    assert two["survey"]["stitched_sha256"] == one["survey"]["stitched_sha256"]
    # node evaluations of all ranks = the single rank's (every tile is classified exactly once)
    assert two["config"]["nodes_per_step_per_gpu"] < one["config"]["nodes_per_step_per_gpu"]
    assert abs(two["value"] * two["ms_per_step"] - one["value"] * one["ms_per_step"]) < 1e-6 * one["value"] * one["ms_per_step"]


def test_tile_workload_two_rank_rehearsal_counts_both_ranks(gpu_device):
    """The headline workload through the same multi-rank plumbing (rank spawn, barrier, MAX of the elapsed times, SUM of the node counts
    over the ranks) as a `--share-gpu` rehearsal: every rank keeps its full batch (weak scaling), `value` is the nodes of BOTH ranks over
    the slowest rank's time -- so value x ms_per_step is twice the per-rank node count.  Labelled a rehearsal; no scaling claim."""
    base = ["--tiles", "8", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]
    one = _line(base)
    two = _line(base + ["--gpus", "2", "--share-gpu"])
    assert one["n_gpus"] == 1 and "rehearsal" not in one
    assert two["n_gpus"] == 2 and two["rehearsal"] is True and two["scaling"] == "weak"
    per_rank = one["config"]["nodes_per_step_per_gpu"]
    assert two["config"]["nodes_per_step_per_gpu"] == per_rank and "x2" in two["config"]["parallelism"]
    assert abs(one["value"] * one["ms_per_step"] * 1e-3 - per_rank) < 1e-6 * per_rank
    assert abs(two["value"] * two["ms_per_step"] * 1e-3 - 2 * per_rank) < 1e-6 * per_rank
