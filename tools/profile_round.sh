#!/bin/bash
# Round-end profile set on the GPU box: bench line, rocprofv3 kernel-trace stats of the same command, and the
# FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, as the guide prescribes).  Outputs under gpurun_out/final/.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py --steps 10 --warmup 3 > $OUT/trace.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python3 $R/bench.py --steps 4 --warmup 2 > $OUT/pmc_$c.log 2>&1 || exit 1
done
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
cd $R && python tools/pmc_summary.py "gpurun_out/final/pmc_*/**/*counter_collection.csv" > $OUT/pmc_summary.txt 2>&1
cat $OUT/bench.json; head -8 $OUT/kernel_stats.csv; cat $OUT/pmc_summary.txt
