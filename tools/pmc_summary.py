#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean of each counter over its dispatches.
usage: python tools/pmc_summary.py gpurun_out/pmc/p*/**/*_counter_collection.csv"""
import collections
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"(\w+_kernel)<([^>]*)>", name)
    return (m.group(1) + "<" + m.group(2).replace("__hip_bfloat16", "bf16") + ">") if m else name[:40]


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for pat in sys.argv[1:]:
    for path in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(path)):
            if "sfa" not in r["Kernel_Name"]:
                continue
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} {sum(v) / len(v):16.1f}   (n={len(v)})")
