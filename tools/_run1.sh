set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k "pair_major or extractor_layer_inside or predict_matches_oracle or full_batch_properties" > gpurun_out/t1.log 2>&1 || { tail -30 gpurun_out/t1.log; exit 1; }
tail -3 gpurun_out/t1.log
timeout -k 10 400 python tools/ab_bench.py --libs base=bathymetric-gnn_amd/libbgnn_hip.so:BGNN_NO_PAIR_MAJOR=1 new=bathymetric-gnn_amd/libbgnn_hip.so --workloads tiles --steps 10 --repeat 2 > gpurun_out/ab1.log 2>&1
cat gpurun_out/ab1.log
