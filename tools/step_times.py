#!/usr/bin/env python3
"""Diagnostic: per-step wall time of the C3 fwd+bwd step from a cold start (clock / allocator ramp)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sink-flash-attention-kernel_amd"), ROOT]
import torch
from sink_attention import sink_flash_attention
B, Hq, Hkv, N, D, ns, W = 4, 32, 8, 8192, 128, 4, 4096
q = torch.randn(B, Hq, N, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
k = torch.randn(B, Hkv, N, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
v = torch.randn(B, Hkv, N, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
do = torch.randn(B, Hq, N, D, device="cuda", dtype=torch.bfloat16)
torch.cuda.synchronize()
ts = []
for i in range(40):
    t0 = time.perf_counter()
    o = sink_flash_attention(q, k, v, ns, W)
    o.backward(do)
    q.grad = k.grad = v.grad = None
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.2f}" for t in ts))
