#!/usr/bin/env python3
"""profiles/rNN_traffic_pmc.json from a tools/pmc_summary.py text (FETCH_SIZE / WRITE_SIZE of the C3 bench kernels):
usage: python tools/traffic_json.py gpurun_out/r03/c3_pmc_summary.txt > profiles/r03_traffic_pmc.json"""
import json
import re
import sys

NAMES = {"fwd_asm_pk_kernel": "fwd", "bwd_preprocess_vec_kernel": "bwd_preprocess", "bwd_dkdv_asm_kernel": "bwd_dkdv",
         "bwd_dq_asm_pk_kernel": "bwd_dq"}
out, cur = {}, None
for line in open(sys.argv[1]):
    m = re.match(r"(\w+)<", line)
    if m:
        cur = NAMES.get(m.group(1))
        continue
    m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([\d.]+)", line)
    if m and cur:
        out.setdefault(cur, {})[m.group(1)] = float(m.group(2))
res = {}
for k, v in out.items():
    f, w = v.get("FETCH_SIZE", 0.0) * 1024 * 2, v.get("WRITE_SIZE", 0.0) * 1024
    res[k] = {"fetch_bytes_corrected_x2": f, "write_bytes": w, "hbm_bytes_per_launch": f + w}
res["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_r03.sh) on `python bench.py --steps 4 "
                "--warmup 2`, mean per dispatch; FETCH_SIZE*1024*2 (gfx950 reports half of wide coalesced reads, "
                "MI355X_MICROARCH.md section HBM) + WRITE_SIZE*1024")
print(json.dumps(res, indent=1))
