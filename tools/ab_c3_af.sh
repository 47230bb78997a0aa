#!/bin/bash
# configs[2] with layer 0 aggregate-first (default) against the front-GEMM form, same box, same build: bench lines + per-kernel stats.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_c3_af
mkdir -p $O
cd $R
for rep in 1 2; do
  python3 bench.py --workload c3 --no-extras --no-cpu-baseline --detail $O/af_$rep.detail.json > $O/af_$rep.json 2> $O/af_$rep.err
  BGNN_NO_LAYER0_AF=1 python3 bench.py --workload c3 --no-extras --no-cpu-baseline --detail $O/front_$rep.detail.json > $O/front_$rep.json 2> $O/front_$rep.err
done
python3 - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "ab_c3_af")
for f in sorted(glob.glob(O + "/*.detail.json")):
    d = json.load(open(f))
    print(os.path.basename(f), round(d["value"] / 1e6, 1), "M nodes/s", round(d["ms_per_step"], 3), "ms", {k: round(v["ms_per_step"], 3) for k, v in d.get("kernels", {}).items()})
PY
