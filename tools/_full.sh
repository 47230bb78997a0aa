set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_suite.log 2>&1 || { tail -40 gpurun_out/gpu_suite.log; exit 1; }
tail -4 gpurun_out/gpu_suite.log
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.stderr
cat gpurun_out/bench_default.json
