#!/bin/bash
# MFMA / VALU utilisation counters of the fused layer kernel (exact-f32 and the two split matrix paths): one rocprofv3
# --pmc pass per derived metric, kernel trace only (no other trace domains).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_util
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in exact bf16 f16; do
  flag=""; [ $mode = bf16 ] && flag="--split-bf16"; [ $mode = f16 ] && flag="--split-f16"
  for ctr in MfmaUtil VALUBusy; do
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/${mode}_$ctr -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 $flag > /dev/null 2>&1
  done
  echo "$mode done"
done
find $O -name "*kernel_trace.csv" -delete
ls $O
