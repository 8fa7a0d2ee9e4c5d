#!/bin/bash
# What the chip does under the metric's kernels: rocm-smi power / clocks sampled while bench.py loops in the background
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
echo "== idle"; rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -i "power\|sclk\|mclk\|fclk" | head -12
python $R/bench.py --steps 1500 --warmup 5 --no-cpu-baseline --sustain-seconds 0 > /tmp/bench_loop.json 2>/dev/null &
BP=$!
sleep 25
for i in 1 2 3 4; do echo "== under load, sample $i"; rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk" | head -6; sleep 1.5; done
wait $BP
cut -c1-200 /tmp/bench_loop.json
