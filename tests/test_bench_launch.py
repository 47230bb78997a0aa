"""bench.py --gpus N: argument / launch logic (CPU only, no GPU work).

The driver starts N ranks through ``torch.distributed.run`` (WORLD_SIZE set); a bare ``python bench.py --gpus N`` must
start the N ranks itself or fail loudly -- it must never print an ``n_gpus: 1`` line for N > 1.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_plan_cases():
    assert bench.launch_plan(1, {}, 0) == ("single", None)
    assert bench.launch_plan(1, {"WORLD_SIZE": "1"}, 1) == ("single", None)
    assert bench.launch_plan(8, {"WORLD_SIZE": "8"}, 8) == ("worker", None)           # torch.distributed.run
    assert bench.launch_plan(4, {}, 8) == ("spawn", None)                             # bare python bench.py --gpus 4
    mode, why = bench.launch_plan(2, {}, 1)                                           # one-GPU box
    assert mode == "error" and "only 1 GPU" in why
    mode, why = bench.launch_plan(2, {"WORLD_SIZE": "4"}, 8)
    assert mode == "error" and "WORLD_SIZE=4" in why
    assert bench.launch_plan(0, {}, 8)[0] == "error"


def test_spawn_ranks_sets_the_rendezvous_environment(tmp_path):
    script = tmp_path / "child.py"
    script.write_text(
        "import json, os, sys\n"
        "keys = ['RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY']\n"
        "open(os.path.join(sys.argv[1], 'rank%s.json' % os.environ['RANK']), 'w').write(json.dumps({k: os.environ[k] for k in keys}))\n")
    rc = bench.spawn_ranks(3, [str(script), str(tmp_path)])
    assert rc == 0
    seen = [json.load(open(tmp_path / f"rank{i}.json")) for i in range(3)]
    assert [s["RANK"] for s in seen] == ["0", "1", "2"] and [s["LOCAL_RANK"] for s in seen] == ["0", "1", "2"]
    assert {s["WORLD_SIZE"] for s in seen} == {"3"} and {s["MASTER_ADDR"] for s in seen} == {"127.0.0.1"}
    assert len({s["MASTER_PORT"] for s in seen}) == 1 and {s["HSA_ENABLE_IPC_MODE_LEGACY"] for s in seen} == {"0"}


def test_spawn_ranks_reports_a_failing_rank(tmp_path):
    script = tmp_path / "child.py"
    script.write_text("import os, sys\nsys.exit(7 if os.environ['RANK'] == '1' else 0)\n")
    assert bench.spawn_ranks(2, [str(script)]) == 7


def test_bare_multi_gpu_invocation_fails_loudly_without_the_devices():
    """In this container (and on a 1-GPU box) `python bench.py --gpus 2` must exit non-zero with a one-line reason and
    print no JSON line."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the bare invocation would really run")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0
    assert b"--gpus 2 but only" in r.stderr and b"n_gpus" not in r.stdout
