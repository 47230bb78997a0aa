cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 300 python bench.py --no-extras 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value']/1e6, j['ms_per_step'], {k:round(v['ms_per_step'],2) for k,v in j['kernels'].items()})"
timeout -k 10 300 python bench.py --no-extras 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value']/1e6, j['ms_per_step'], {k:round(v['ms_per_step'],2) for k,v in j['kernels'].items()})"
