cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_c3m -- python3 /root/repo/bench.py --workload c3 --no-extras --steps 6 --warmup 2 > /dev/null 2>&1
find /root/repo/gpurun_out/prof_c3m -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-200
