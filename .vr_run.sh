cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py tests/test_gpu_vr_bag.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 300 python bench.py --workload vr --vr-budget 1000000 --no-extras 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value']/1e6, j['ms_per_step'])"
timeout -k 10 300 python bench.py --workload vr --no-extras 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value']/1e6, j['ms_per_step'])"
