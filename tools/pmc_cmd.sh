#!/bin/bash
# Issue / stall counters of whatever kernels one kbench run launches: tools/pmc_cmd.sh <tag> <kbench args...>
# (separate rocprofv3 --pmc passes with --kernel-trace only; summary by tools/pmc_summary.py)
set -u
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o p -- python3 $R/tools/kbench.py "$@" > $OUT/g$i.log 2>&1 || echo "group $i ($grp) failed"
done
cd $R && python tools/pmc_summary.py "gpurun_out/pmc_$TAG/g*/**/*counter_collection.csv" | tee gpurun_out/pmc_$TAG.txt
