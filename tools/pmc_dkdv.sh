#!/bin/bash
# SQ counters of the backward kernels at C3 (separate rocprofv3 --pmc passes, small groups, --kernel-trace only)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_dkdv
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o p -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/g$i.log 2>&1 || echo "group $i ($grp) failed"
done
cd $R && python tools/pmc_summary.py "gpurun_out/pmc_dkdv/g*/**/*counter_collection.csv" > gpurun_out/pmc_dkdv/summary.txt 2>&1
cat gpurun_out/pmc_dkdv/summary.txt
