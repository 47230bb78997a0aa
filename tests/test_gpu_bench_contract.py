"""bench.py's ONE JSON line on a GPU, at small sizes: the driver's contract keys and a roofline whose fraction stays within (0, 1] --
for the tile, configs[2], VR and survey workloads (each as its own child process, as the driver runs it; `--no-extras`: the default
line's extras -- config3 / config4 / config5, cpu_baseline -- are what the round's evidence runs exercise)."""
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
], ids=["tiles", "c3", "vr", "survey"])
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
