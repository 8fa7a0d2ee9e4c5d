#!/usr/bin/env python3
"""Diagnostic: run one C3 backward with the stamped build (tools/bin/libsfa_stamps.so, -DSFA_STAMPS) and print where
the plain dK/dV loop of one mid-grid workgroup spends its cycles (per wave, per phase).  Not part of the product."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sink-flash-attention-kernel_amd"), ROOT]
os.environ.setdefault("SFA_DKDV", "3")
import torch

from sink_attention import _native

_native.LIB_PATH = os.path.join(ROOT, "tools", "bin", "libsfa_stamps.so")
from sink_attention import sink_flash_attention

B, Hq, Hkv, N, D, ns, W = 4, 32, 8, 8192, 128, 4, 4096
torch.manual_seed(0)
q = torch.randn(B, Hq, N, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
k = torch.randn(B, Hkv, N, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
v = torch.randn(B, Hkv, N, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
for _ in range(3):
    o = sink_flash_attention(q, k, v, ns, W)
    o.backward(torch.randn_like(o))
torch.cuda.synchronize()
lib = _native.lib()
buf = (ctypes.c_ulonglong * 64)()
lib.sfa_debug_read_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
rc = lib.sfa_debug_read_stamps(buf)
names = ["dma issue", "lds reads landed", "S chain + exp (score)", "dP chain+pack (score) / MFMAs (acc)", "lds write (score)", "waitcnt+barrier", "(n_it)", "loop-top gap"]
print("rc", rc, _native.last_path())
for w in range(8):
    vals = [buf[w * 8 + i] for i in range(8)]
    n_it = vals[6]
    vals[7] = 0 if vals[7] > (1 << 40) else vals[7]
    tot = sum(vals[i] for i in (0, 1, 2, 3, 4, 5, 7))
    print(f"wave {w}: n_it={n_it} total={tot} cycles/iter={tot / max(n_it, 1):.0f}")
    for i in (7, 0, 1, 2, 3, 4, 5):
        print(f"    {names[i]:36s} {vals[i]:12d}  {vals[i] / max(n_it, 1):8.0f}/iter  {100.0 * vals[i] / max(tot, 1):5.1f}%")
