cd /root/repo
for s in 1 4; do
timeout -k 10 300 python bench.py --workload vr --vr-streams $s --no-extras 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value']/1e6, j['ms_per_step'], j['config']['workload'][:120], {k:round(v['ms_per_step'],2) for k,v in j['kernels'].items()})"
timeout -k 10 300 python bench.py --workload vr --vr-budget 1000000 --vr-streams $s --no-extras 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value']/1e6, j['ms_per_step'], j['config']['workload'][:120], {k:round(v['ms_per_step'],2) for k,v in j['kernels'].items()})"
done
timeout -k 10 300 python tools/vr_streams_probe.py
