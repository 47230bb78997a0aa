#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py invocation (no extras), per-kernel average durations on stdout.
# usage (on the GPU box): tools/quick_stats.sh <name> [bench.py flags...]      e.g.  tools/quick_stats.sh c3 --workload c3
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
O=$R/gpurun_out/quick/$NAME
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 1 "$@" > $O/bench.json 2> $O/bench.err
find $O -name "*kernel_trace.csv" -delete
F=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    n = re.sub(r"^void bgnn::", "", r["Name"]); n = re.sub(r"\(.*", "", n)
    print(f"{n[:70]:70s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e6:8.4f} ms  total {float(r['TotalDurationNs'])/1e6:9.3f} ms  {float(r['Percentage']):5.1f}%")
PY
