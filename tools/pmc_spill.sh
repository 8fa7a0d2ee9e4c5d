#!/bin/bash
# HBM traffic of the spill backward: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short bench run
cd /tmp && export TMPDIR=/tmp
export SINK_ATTENTION_DS_SPILL_GB=64 SFA_DQ_GEMM_NW=8
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_spill_$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_spill_$c.log 2>&1 || exit 1
done
cd $GRAFT_REPO_ROOT && python tools/pmc_summary.py "gpurun_out/pmc_spill_*/**/*counter_collection.csv"
