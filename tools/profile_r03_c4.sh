#!/bin/bash
# Round-3 profile of the short-window shapes (C4, gpt-oss D=64 sliding layer): kernel-trace stats and FETCH / WRITE_SIZE, one
# process per pass (outputs under gpurun_out/r03c4/, summaries copied to profiles/ by the builder)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03c4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in C4 oss_swa; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$cfg -o t -- python3 $R/tools/kbench.py fwd bwd --cfg $cfg --iters 20 > $OUT/kt_$cfg.log 2>&1 || echo "trace $cfg failed"
  find $OUT/kt_$cfg -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${cfg}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${cfg}_$c -o p -- python3 $R/tools/kbench.py fwd bwd --cfg $cfg --iters 4 > $OUT/pmc_${cfg}_$c.log 2>&1 || echo "pmc $cfg $c failed"
  done
  cd $R && python tools/pmc_summary.py "gpurun_out/r03c4/pmc_${cfg}_*/**/*counter_collection.csv" > $OUT/${cfg}_pmc_summary.txt 2>&1; cd /tmp
done
grep -h "fwd \|f+b" $OUT/kt_*.log
