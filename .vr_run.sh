cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k "bf16 or config3 or canvas or c3" 2>&1 | tail -3
timeout -k 10 300 python bench.py --workload c3 --no-extras 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value']/1e6, j['ms_per_step'], {k:round(v['ms_per_step'],2) for k,v in j['kernels'].items()})"
timeout -k 10 300 python bench.py --workload c3 --no-extras 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value']/1e6, j['ms_per_step'], {k:round(v['ms_per_step'],2) for k,v in j['kernels'].items()})"
