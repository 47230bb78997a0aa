#!/bin/bash
# Round-end evidence run (on the GPU box): bench lines, rocprofv3 kernel stats and the two PMC passes (FETCH_SIZE,
# WRITE_SIZE, each alone with --kernel-trace, as MI355X_MICROARCH.md prescribes), for the fused and the unfused path.
# usage: collect_profiles.sh [all|bench|prof]   (two gpurun calls of <= 20 min: `bench`, then `prof`; the files merge under gpurun_out/prof_final)
set -e
STAGE=${1:-all}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# the kernels these profiles are collected on (bench.py refuses PMC traffic stamped with another id)
python3 -c "import sys; sys.path.insert(0, '$R'); from bathymetric_gnn_amd import runtime; print(runtime.build_id())" > $O/build_id.txt
if [ "$STAGE" = all ] || [ "$STAGE" = bench ]; then
python3 $R/bench.py --detail $O/bench_fused.detail.json > $O/bench_fused.json 2> $O/bench_fused.stderr
python3 $R/bench.py --unfused --no-cpu-baseline --detail $O/bench_unfused.detail.json > $O/bench_unfused.json 2> $O/bench_unfused.stderr
python3 $R/bench.py --split-bf16 --no-cpu-baseline --detail $O/bench_split.detail.json > $O/bench_split.json 2> $O/bench_split.stderr
python3 $R/bench.py --split-f16 --no-cpu-baseline --detail $O/bench_split_f16.detail.json > $O/bench_split_f16.json 2> $O/bench_split_f16.stderr
python3 $R/bench.py --workload c3 --no-cpu-baseline --detail $O/bench_c3.detail.json > $O/bench_c3.json 2> $O/bench_c3.stderr
python3 $R/bench.py --bf16 --no-cpu-baseline --detail $O/bench_bf16_k8.detail.json > $O/bench_bf16_k8.json 2> $O/bench_bf16_k8.stderr
python3 $R/bench.py --workload vr --vr-budget 50000 --vr-streams 1 --detail $O/bench_vr_50k_1ctx.detail.json > $O/bench_vr_50k_1ctx.json 2> $O/bench_vr_50k_1ctx.stderr
python3 $R/bench.py --workload vr --vr-budget 50000 --vr-streams 2 --detail $O/bench_vr_50k_2ctx.detail.json > $O/bench_vr_50k_2ctx.json 2> $O/bench_vr_50k_2ctx.stderr
python3 $R/bench.py --workload vr --vr-budget 50000 --vr-streams 4 --detail $O/bench_vr_50k.detail.json > $O/bench_vr_50k.json 2> $O/bench_vr_50k.stderr
python3 $R/bench.py --workload vr --vr-budget 1000000 --vr-streams 1 --detail $O/bench_vr_1M.detail.json > $O/bench_vr_1M.json 2> $O/bench_vr_1M.stderr
python3 $R/bench.py --workload survey --survey-size 20000 --steps 2 --warmup 0 --detail $O/bench_survey_20000.detail.json > $O/bench_survey_20000.json 2> $O/bench_survey_20000.stderr
for t in GCN GraphSAGE GIN; do
  python3 $R/bench.py --gnn-type $t --detail $O/bench_$t.detail.json > $O/bench_$t.json 2> $O/bench_$t.stderr
done
echo "bench stage done"
fi
if [ "$STAGE" = all ] || [ "$STAGE" = prof ]; then
for mode in fused unfused split c3; do
  flag=""; [ $mode = unfused ] && flag="--unfused"; [ $mode = split ] && flag="--split-bf16"; [ $mode = c3 ] && flag="--workload c3"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${mode}_stats -- python3 $R/bench.py --no-extras --steps 5 --warmup 1 $flag > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${mode}_fetch -- python3 $R/bench.py --no-extras --steps 5 --warmup 1 $flag > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${mode}_write -- python3 $R/bench.py --no-extras --steps 5 --warmup 1 $flag > /dev/null 2>&1
  echo "$mode profiled"
done
fi
# keep the merge small: drop the per-dispatch traces of the stats passes
find $O -name "*kernel_trace.csv" -path "*_stats*" -delete
ls $O
