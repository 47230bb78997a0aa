#!/usr/bin/env python3
"""A/B of two builds of the library on the same box: `bench.py --no-extras` per workload with BGNN_LIB pointing at each .so, the
per-kernel-class times (HIP events, ms per step) side by side.

A variant may also be an environment switch of ONE build: name=path:VAR=value[:VAR2=value2].

usage: python tools/ab_bench.py [--libs base=bathymetric-gnn_amd/libbgnn_hip_base.so new=bathymetric-gnn_amd/libbgnn_hip.so]
                                [--workloads tiles c3] [--steps 10] [--repeat 2]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(lib, workload, steps, extra):
    with tempfile.NamedTemporaryFile(suffix=".json", delete=False) as f:
        side = f.name
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras", "--steps", str(steps), "--warmup", "3", "--detail", side]
    if workload != "tiles":
        args += ["--workload", workload]
    args += extra
    lib, *sets = lib.split(":")
    env = dict(os.environ, BGNN_LIB=os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib)
    env.update(dict(x.split("=", 1) for x in sets))
    r = subprocess.run(args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise SystemExit(f"{lib} {workload}: bench failed\n{r.stderr[-2000:]}")
    d = json.load(open(side))
    os.unlink(side)
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", default=["base=bathymetric-gnn_amd/libbgnn_hip_base.so", "new=bathymetric-gnn_amd/libbgnn_hip.so"])
    ap.add_argument("--workloads", nargs="+", default=["tiles", "c3"])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--extra", nargs="*", default=[])
    a = ap.parse_args()
    libs = [x.split("=", 1) for x in a.libs]
    for wl in a.workloads:
        print(f"== {wl}")
        for rep in range(a.repeat):
            for name, lib in libs:
                d = run(lib, wl, a.steps, a.extra)
                k = d["kernels"]
                print(f"  {name:6s} run {rep}: {d['value'] / 1e6:8.1f} M nodes/s  {d['ms_per_step']:7.3f} ms/step | " +
                      "  ".join(f"{n} {v['ms_per_step']:.3f}" for n, v in k.items()), flush=True)


if __name__ == "__main__":
    main()
