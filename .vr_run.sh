set -e
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py tests/test_gpu_vr_bag.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -5
for s in 1 4; do
timeout -k 10 300 python bench.py --workload vr --vr-streams $s --no-extras 2>&1 | tail -1
timeout -k 10 300 python bench.py --workload vr --vr-budget 1000000 --vr-streams $s --no-extras 2>&1 | tail -1
done
