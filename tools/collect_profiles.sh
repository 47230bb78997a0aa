#!/bin/bash
# Round-end evidence run (on the GPU box): bench lines, rocprofv3 kernel stats and the two PMC passes (FETCH_SIZE,
# WRITE_SIZE, each alone with --kernel-trace, as MI355X_MICROARCH.md prescribes), for the fused and the unfused path.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_fused.json
python3 $R/bench.py --unfused --no-cpu-baseline > $O/bench_unfused.json
python3 $R/bench.py --split-bf16 --no-cpu-baseline > $O/bench_split.json
python3 $R/bench.py --split-f16 --no-cpu-baseline > $O/bench_split_f16.json
python3 $R/bench.py --workload c3 --no-cpu-baseline > $O/bench_c3.json
python3 $R/bench.py --bf16 --no-cpu-baseline > $O/bench_bf16_k8.json
python3 $R/bench.py --workload vr --vr-budget 50000 --vr-streams 1 > $O/bench_vr_50k_1ctx.json
python3 $R/bench.py --workload vr --vr-budget 50000 --vr-streams 2 > $O/bench_vr_50k_2ctx.json
python3 $R/bench.py --workload vr --vr-budget 50000 --vr-streams 4 > $O/bench_vr_50k.json
python3 $R/bench.py --workload vr --vr-budget 1000000 --vr-streams 1 > $O/bench_vr_1M.json
python3 $R/bench.py --workload survey --survey-size 20000 --steps 2 --warmup 0 > $O/bench_survey_20000.json
for mode in fused unfused split c3; do
  flag=""; [ $mode = unfused ] && flag="--unfused"; [ $mode = split ] && flag="--split-bf16"; [ $mode = c3 ] && flag="--workload c3"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${mode}_stats -- python3 $R/bench.py --no-extras --steps 5 --warmup 1 $flag > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${mode}_fetch -- python3 $R/bench.py --no-extras --steps 5 --warmup 1 $flag > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${mode}_write -- python3 $R/bench.py --no-extras --steps 5 --warmup 1 $flag > /dev/null 2>&1
  echo "$mode profiled"
done
# keep the merge small: drop the per-dispatch traces of the stats passes
find $O -name "*kernel_trace.csv" -path "*_stats*" -delete
ls $O
