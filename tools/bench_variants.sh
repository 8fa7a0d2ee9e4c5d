#!/bin/bash
# usage: tools/bench_variants.sh "ENV1=a ENV2=b" "ENV1=c" ...   -> one bench.py line per variant (stage times)
for v in "$@"; do
  out=$(env $v timeout -k 10 300 python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1)
  python - "$v" "$out" <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2]); r = j["roofline"]
    print(f"{sys.argv[1]:60s} {j['value']:7.1f} TF  {j['ms_per_step']:.3f} ms  {r['stage_ms']}  {r['kernel']}")
except Exception as e:
    print(sys.argv[1], "ERR", e, sys.argv[2][:200])
PY
done
