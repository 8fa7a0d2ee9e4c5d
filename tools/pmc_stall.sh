#!/bin/bash
# Stall-attribution counters of the MFMA kernels at C3 (separate rocprofv3 --pmc passes, small groups, --kernel-trace
# only).  A group whose counters this build of rocprofv3 does not know is reported and skipped.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_stall
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 60 rocprofv3 -L > $OUT/counters.txt 2>&1
i=0
for grp in "SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INST_LEVEL_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o p -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/g$i.log 2>&1 || echo "group $i ($grp) failed"
done
cd $R && python tools/pmc_summary.py "gpurun_out/pmc_stall/g*/**/*counter_collection.csv"
