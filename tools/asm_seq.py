#!/usr/bin/env python3
"""Print the instruction-class sequence of the MFMA-heavy basic blocks of a kernel (hipcc -S).
M mfma, r ds_read, t ds_read_tr, w ds_write, G buffer_load, E v_exp, a v_accvgpr_*, W s_waitcnt, n s_nop,
B s_barrier, X scratch, v other VALU, s other SALU.   usage: asm_seq.py <file.hip> <kernel regex> [min_mfma]"""
import re
import subprocess
import sys

src, pat = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 16
out = "/tmp/asm_seq.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc",
                "-I/root/repo/include", "-I/root/repo/sink-flash-attention-kernel_amd/csrc", "-S",
                "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
name = [n for n in re.findall(r"^(_Z\S+):", s, re.M) if re.search(pat, n)][0]
body = s[s.index(name + ":"):]
body = body[:body.index(".Lfunc_end")]


def cls(l):
    t = l.split()[0]
    for pre, c in (("v_mfma", "M"), ("ds_read_b64_tr", "t"), ("ds_read", "r"), ("ds_write", "w"), ("buffer_load", "G"),
                   ("buffer_store", "S"), ("v_exp", "E"), ("v_accvgpr", "a"), ("s_waitcnt", "W"), ("s_nop", "n"),
                   ("s_barrier", "B"), ("scratch", "X"), ("v_", "v"), ("s_", "s")):
        if t.startswith(pre):
            return c
    return "?"


print(name)
for b in re.split(r"\n(?=\.LBB\d+_\d+:)", body):
    first = b.split("\n")[0]
    lines = [l.strip() for l in b.split("\n")[1:] if l.strip() and not l.strip().startswith(";")]
    nm = sum("v_mfma" in l for l in lines)
    if nm >= min_mfma:
        seq = "".join(cls(l) for l in lines)
        import collections
        print(first.split()[0], len(lines), "instrs", dict(collections.Counter(seq)))
        print(seq)
