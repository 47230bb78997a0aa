#!/bin/bash
# Where do the waves of the fused layer kernels spend their cycles?  SQ wait / active-instruction / LDS / TA-FIFO counters, a few per
# rocprofv3 --pmc pass (kernel trace only), for one bench mode:   bash tools/collect_sq_counters.sh [c3|exact]
set -e
MODE=${1:-c3}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_sq_$MODE
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$R'); from bathymetric_gnn_amd import runtime; print(runtime.build_id())" > $O/build_id.txt
flag=""; [ $MODE = c3 ] && flag="--workload c3"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "LdsUtil" "MemUnitStalled" "SALUBusy" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -- python3 $R/bench.py --no-extras --steps 3 --warmup 1 $flag > /dev/null 2>&1 || echo "group $i failed: $grp"
  echo "group $i done"
done
find $O -name "*kernel_trace.csv" -delete
