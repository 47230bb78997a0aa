#!/bin/bash
# MFMA / VALU utilisation counters of the fused layer kernels (exact-f32 headline and configs[2]): one rocprofv3
# --pmc pass per derived metric, kernel trace only (no other trace domains).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_util
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# the kernels these counters are collected on (the summary carries the stamp; DESIGN.md quotes only stamped figures)
python3 -c "import sys; sys.path.insert(0, '$R'); from bathymetric_gnn_amd import runtime; print(runtime.build_id())" > $O/build_id.txt
for mode in exact c3; do
  flag=""; [ $mode = c3 ] && flag="--workload c3"
  for ctr in MfmaUtil VALUBusy; do
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/${mode}_$ctr -- python3 $R/bench.py --no-extras --steps 3 --warmup 1 $flag > /dev/null 2>&1
  done
  echo "$mode done"
done
find $O -name "*kernel_trace.csv" -delete
ls $O
