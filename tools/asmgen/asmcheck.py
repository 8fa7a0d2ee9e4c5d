"""Assembles a generated instruction list with the ROCm assembler (syntax, operand and constant-bus rules); no GPU."""
from __future__ import annotations

import os
import re
import subprocess
import tempfile

from .core import emit_text

CLANG = "/opt/rocm/lib/llvm/bin/clang"


def assemble(items, mcpu="gfx950"):
    """returns (ok, stderr); inline-asm operands %[name] are bound to scratch SGPRs / a VGPR, %= to a constant"""
    text = emit_text(items, comments=False)
    names = sorted(set(re.findall(r"%\[(\w+)\]", text)))
    smap = {n: "s%d" % (2 + i) for i, n in enumerate(x for x in names if x != "tid")}
    smap["tid"] = "v0"
    text = re.sub(r"%\[(\w+)\]", lambda m: smap[m.group(1)], text).replace("%=", "0")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "k.s")
        open(src, "w").write(text + "\ns_endpgm\n")
        r = subprocess.run([CLANG, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=" + mcpu, "-c", src,
                            "-o", os.path.join(d, "k.o")], capture_output=True, text=True)
        return r.returncode == 0, r.stderr
