#!/usr/bin/env python3
"""Instruction histogram of the MFMA-carrying basic blocks of one kernel in a hipcc -S dump.
usage: python tools/asm_hist.py <file.hip> <kernel-name-regex>"""
import collections
import re
import subprocess
import sys

src, pat = sys.argv[1], sys.argv[2]
out = "/tmp/asm_hist.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc",
                "-I/root/repo/include", "-I/root/repo/sink-flash-attention-kernel_amd/csrc", "-S",
                "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
names = [n for n in re.findall(r"^(_Z\S+):", s, re.M) if re.search(pat, n)]
for name in names[:1]:
    body = s[s.index(name + ":"):]
    body = body[:body.index(".Lfunc_end")]
    print(name)
    tot = collections.Counter()
    for b in re.split(r"\n(?=\.LBB\d+_\d+:)", body):
        lines = [l.strip() for l in b.split("\n") if l.strip() and not l.strip().startswith(";")]
        n_mfma = sum("v_mfma" in l for l in lines)
        if n_mfma >= 4:
            c = collections.Counter(l.split()[0] for l in lines[1:])
            tot.update(c)
            print(" ", lines[0].split()[0], len(lines), "instrs, mfma", n_mfma)
            print("     ", c.most_common(14))
    print("  TOTAL over mfma blocks:", tot.most_common(20))
