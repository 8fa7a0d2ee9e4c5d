#!/bin/bash
# Round-3 profile set on the GPU box (outputs under gpurun_out/r03/, summaries copied to profiles/ by the builder):
#   1. contract bench line (python bench.py)
#   2. rocprofv3 --kernel-trace --stats of the same command (per-kernel average durations)
#   3. FETCH_SIZE / WRITE_SIZE PMC passes of the bench (separate runs, as the guide prescribes)
#   4. SQ counter groups of the three C3 kernels
#   5. kernel-trace stats + FETCH/WRITE_SIZE of the other BASELINE configs: C2 (fwd), C4 (fwd+bwd), C5 (decode)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/trace.log 2>&1 || exit 1
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c3_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1 || exit 1
done
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/sq$i -o p -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/sq$i.log 2>&1 || echo "group $i ($grp) failed"
done
cd $R && python tools/pmc_summary.py "gpurun_out/r03/pmc_*/**/*counter_collection.csv" "gpurun_out/r03/sq*/**/*counter_collection.csv" > $OUT/c3_pmc_summary.txt 2>&1
cd /tmp
# other configs: one process per config and pass
for cfg in "fwd --cfg C2" "fwd bwd --cfg C4" "decode --cfg none"; do
  tag=$(echo $cfg | tr ' -' '__')
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$tag -o t -- python3 $R/tools/kbench.py $cfg --iters 10 > $OUT/kt_$tag.log 2>&1 || echo "trace $cfg failed"
  find $OUT/kt_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${tag}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${tag}_$c -o p -- python3 $R/tools/kbench.py $cfg --iters 4 > $OUT/pmc_${tag}_$c.log 2>&1 || echo "pmc $cfg $c failed"
  done
  cd $R && python tools/pmc_summary.py "gpurun_out/r03/pmc_${tag}_*/**/*counter_collection.csv" > $OUT/${tag}_pmc_summary.txt 2>&1; cd /tmp
done
cat $OUT/bench.json; head -8 $OUT/c3_kernel_stats.csv; cat $OUT/c3_pmc_summary.txt
