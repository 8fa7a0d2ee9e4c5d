#!/usr/bin/env python3
"""Phase anatomy of a dK/dV trip (library built by `tools/build_ab.sh '{}' phases`: fenced s_memtime stamps between
the four MFMA phases).  Cycles per trip in [A] S chains, [B] dP chains (+ exp / pack), [C] dV, [D] dK (+ next-trip
fetches).  The fences forbid the overlap the real kernel has: read shares, not lengths."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sink-flash-attention-kernel_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch

from bench import HipEvents
from kbench import CFG
from sink_attention import _native, sink_flash_attention

B, Hq, Hkv, N, D, ns, W, aux = CFG["C3"]
dev = "cuda"
torch.manual_seed(1)
q = torch.randn(B, Hq, N, D, device=dev, dtype=torch.bfloat16).requires_grad_(True)
k = torch.randn(B, Hkv, N, D, device=dev, dtype=torch.bfloat16).requires_grad_(True)
v = torch.randn(B, Hkv, N, D, device=dev, dtype=torch.bfloat16).requires_grad_(True)
do = torch.randn_like(q)
lib = _native.lib()
nblk = B * Hkv * ((N + 255) // 256)
dbg = torch.zeros(nblk * 4 * 4, dtype=torch.int32, device=dev)
lib.sfa_debug_set_ptr(dbg.data_ptr())
for var in (0, 1, 0, 1, 1):
    lib.sfa_debug_set_variant(0, var)
    ev = HipEvents(4)
    lib.sfa_debug_set_stage_events(ev.ev, 4)
    sink_flash_attention(q, k, v, num_sink=ns, window_size=W).backward(do)
    lib.sfa_debug_set_stage_events(None, 0)
    torch.cuda.synchronize()
    print(f"variant {var}: dkdv {ev.elapsed(1, 2):.4f} ms")
    q.grad = k.grad = v.grad = None
d = dbg.view(nblk, 4, 4).cpu().double()          # [bid][wave][A, B, C, D]
trips = 433152.0
print("cycles per trip (all workgroups, wave 0): A %.0f  B %.0f  C %.0f  D %.0f   sum %.0f" % (
    *(d[:, 0, i].sum().item() / trips for i in range(4)), d[:, 0, :].sum().item() / trips))
