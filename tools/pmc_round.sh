#!/bin/bash
# SQ counters of the three MFMA kernels at C3 (separate rocprofv3 --pmc passes, small groups; no trace domains mixed in)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o p -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/g$i.log 2>&1 || echo "group $i ($grp) failed"
done
cd $R && python tools/pmc_summary.py "gpurun_out/pmc_round/g*/**/*counter_collection.csv"
