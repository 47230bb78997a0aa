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


def _fake_kfd(tmp_path, nodes):
    """nodes: list of (simd_count, render_minor, device_file_exists)"""
    sysfs = tmp_path / "nodes"; dev = tmp_path / "dri"
    sysfs.mkdir(); dev.mkdir()
    for i, (simd, minor, present) in enumerate(nodes):
        (sysfs / str(i)).mkdir()
        (sysfs / str(i) / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\ndrm_render_minor {minor}\n")
        if present:
            (dev / f"renderD{minor}").write_text("")
    return str(sysfs), str(dev)


def test_visible_gpu_count_reads_the_kfd_topology_not_the_runtime(tmp_path):
    """the parent of a multi-rank run counts GPUs without torch / HIP: KFD nodes with SIMDs whose render
    node this process can open, capped by the *_VISIBLE_DEVICES lists."""
    sysfs, dev = _fake_kfd(tmp_path, [(0, 0, False), (0, 0, False)] + [(1024, 128 + i, i < 3) for i in range(8)])
    assert bench.visible_gpu_count({}, sysfs, dev) == 3                        # 8 GPUs on the host, 3 render nodes in this container
    assert bench.visible_gpu_count({"HIP_VISIBLE_DEVICES": "0,1"}, sysfs, dev) == 2
    assert bench.visible_gpu_count({"ROCR_VISIBLE_DEVICES": "0,1,2,3,4"}, sysfs, dev) == 3
    assert bench.visible_gpu_count({"CUDA_VISIBLE_DEVICES": ""}, sysfs, dev) == 0
    assert bench.visible_gpu_count({}, str(tmp_path / "missing"), dev) == 0
    # launch_plan takes the count lazily: a launcher-started worker never evaluates it
    boom = lambda: (_ for _ in ()).throw(AssertionError("evaluated"))
    assert bench.launch_plan(8, {"WORLD_SIZE": "8"}, boom) == ("worker", None)
    assert bench.launch_plan(1, {}, boom) == ("single", None)
    assert bench.launch_plan(2, {}, lambda: 2) == ("spawn", None)


def test_spawn_ranks_kills_the_survivors_of_a_failed_rank(tmp_path):
    """rank 1 dies while rank 0 would sit in a barrier for a long time -- the command must come back with rank 1's
    status at once instead of waiting for rank 0."""
    import time
    script = tmp_path / "child.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n    time.sleep(0.5); sys.exit(9)\n"
                      "time.sleep(120)\n")
    t0 = time.perf_counter()
    assert bench.spawn_ranks(2, [str(script)]) == 9
    assert time.perf_counter() - t0 < 30


@pytest.mark.parametrize("workload", ["c3", "vr", "survey"])
def test_every_workload_is_accepted_under_gpus_n(workload):
    """`--workload c3 | vr | survey` under `--gpus N`: parsed, and (no devices here) refused loudly before any GPU work."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the bare invocation would really run")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload, "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and b"--gpus 2 but only" in r.stderr and b"n_gpus" not in r.stdout
    # a mismatching launcher environment is refused as well (the driver's torch.distributed.run form sets WORLD_SIZE)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload],
                       env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and b"WORLD_SIZE=4" in r.stderr


# ---- the ONE stdout line: size budget, what may be dropped, where the rest goes -----------------------------------------------
def _representative_line():
    """A line with every key the default run produces, with worst-case-long strings."""
    rf = {"bound": "mfma", "achieved": 104.71234567890123, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.6656855415060472,
          "traffic": 13474343608.0, "traffic_frac": 0.27012345678901234, "kernel": "gat_layer_fused_kernel", "avg_launch_ms": 6.251234567890123}
    brief = {"value": 812345678.1234567, "ms_per_step": 10.323456789012345, "dtype": "bf16", "workload": "w" * 120,
             "roofline": {"bound": "hbm", "frac": 0.4512345678901234, "kernel": "gat_layer_fused_kernel"}}
    return {"metric": "classified tile-nodes/s (fused graph build + 4-layer GAT forward + scatter)", "value": 303612345.1234567,
            "unit": "nodes/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 27.63123456789012, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "x" * 200, "tiles_per_gpu": 128, "tile": 256, "nodes_per_step_per_gpu": 8388608,
                       "parallelism": "tile-sharded x1, no collective"},
            "roofline": rf, "path": "fused", "matrix_path": "exact f32",
            "cpu_baseline": {"value": 15926.123456789, "unit": "nodes/s", "cores": 128, "kind": "port", "sample": "s" * 200},
            "gpu_over_cpu": 19065.12345678901,
            "pcie_inclusive": {"value": 291234567.1234567, "ms_per_step": 28.712345678901234},
            "single_tile": {"value": 171234567.1234567, "ms_per_tile": 0.39123456789012345},
            "config3": dict(brief), "config5": dict(brief),
            "gnn_types": {k: {"value": 412345678.1234567, "frac": 0.3312345678901234, "kernel": "neighbor_reduce_kernel"} for k in ("GCN", "GraphSAGE", "GIN")},
            "config4": dict(brief, one_context=157123456.12345678, two_contexts=203123456.12345678,
                            processor_api={"synchronous": 15123456.123456789, "pipelined": 16123456.123456789}),
            "opt_in_split": {"note": "not the headline; max |dlogit| vs the exact-f32 path",
                             "bf16x3": {"value": 478901592.1696222, "max_dlogit": 5.364418029785156e-06},
                             "fp16x3": {"value": 473647466.0648265, "max_dlogit": 5.662441253662109e-07}},
            "detail": "gpurun_out/bench_detail.json"}


def test_default_line_fits_the_stdout_budget_with_every_key():
    line = _representative_line()
    out = bench.compact_line(line)
    assert out == line, "nothing had to be dropped"
    assert len(json.dumps(out)) < 4096 and bench.LINE_BUDGET <= 4096


def test_compact_line_drops_optional_keys_but_never_the_contract():
    line = _representative_line()
    line["config3"]["workload"] = "y" * 3000
    out = bench.compact_line(line)
    assert len(json.dumps(out)) <= bench.LINE_BUDGET
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out
    line["config"]["workload"] = "z" * 9000
    out = bench.compact_line(line)
    assert len(json.dumps(out)) <= bench.LINE_BUDGET and "roofline" in out and "cpu_baseline" in out


def test_emit_writes_detail_to_side_file_and_one_line_to_stdout(tmp_path, capsys):
    line = _representative_line()
    detail = {"rooflines": {"a": {"note": "n" * 5000}}, "kernels": {"fused": {"ms_per_step": 1.0}}, "config4": {"one_context": {"big": "b" * 9000}}}
    side = tmp_path / "sub" / "detail.json"
    bench.emit(line, detail, str(side))
    cap = capsys.readouterr()
    lines = [ln for ln in cap.out.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{") and len(lines[0]) <= bench.LINE_BUDGET
    j = json.loads(lines[0])
    assert "rooflines" not in j and j["roofline"]["frac"] == line["roofline"]["frac"] and j["detail"] == str(side)
    full = json.load(open(side))
    assert full["rooflines"]["a"]["note"].startswith("n") and full["value"] == line["value"]
    assert full["config4"]["one_context"]["big"].startswith("b")        # detail's full config4 wins over the line's brief one
    assert "bench detail: {" in cap.err


def test_pmc_traffic_is_refused_for_other_kernels(tmp_path, monkeypatch):
    """profiles/pmc_traffic.json carries the kernel-source hash it was collected on; a library built from other sources gets no
    `traffic` (null) instead of a stale one."""
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "pmc_traffic.json").write_text(json.dumps({"_kernel_source_sha": "abc", "gat_layer_fused_kernel": {"hbm_bytes_per_launch": 1.0}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_traffic_table("abc")["gat_layer_fused_kernel"]["hbm_bytes_per_launch"] == 1.0
    assert bench.pmc_traffic_table("other") == {} and bench.pmc_traffic_table(None) == {}
    (prof / "pmc_traffic.json").write_text(json.dumps({"gat_layer_fused_kernel": {"hbm_bytes_per_launch": 1.0}}))      # unstamped (rounds 1-3)
    assert bench.pmc_traffic_table("abc") == {}
