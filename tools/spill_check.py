#!/usr/bin/env python3
"""Diagnostic: dS-spill backward (SFA_FLAG_BWD_SPILL_DS) vs the recompute backward: gradient agreement on odd
shapes and stage times at C3."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sink-flash-attention-kernel_amd"), ROOT]
import torch

from sink_attention import _native, sink_flash_attention


def grads(q, k, v, do, ns, W, sa, gb):
    os.environ["SINK_ATTENTION_DS_SPILL_GB"] = gb
    qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
    ss = sa.clone().requires_grad_(True) if sa is not None else None
    o = sink_flash_attention(qq, kk, vv, ns, W, ss)
    o.backward(do)
    torch.cuda.synchronize()
    return qq.grad, kk.grad, vv.grad, (ss.grad if ss is not None else None), _native.last_path()


torch.manual_seed(0)
bad = 0
for (B, Hq, Hkv, N, D, ns, W, aux, dt) in [
    (1, 4, 2, 128, 128, 4, 32, True, torch.bfloat16), (2, 8, 2, 517, 128, 3, 100, True, torch.bfloat16),
    (1, 4, 4, 1000, 64, 0, 1000, False, torch.float16), (1, 8, 1, 777, 80, 130, 64, True, torch.bfloat16),
    (1, 2, 2, 2048, 96, 4, 2048, False, torch.float16), (1, 4, 1, 33, 128, 4, 8, True, torch.bfloat16),
    (1, 4, 2, 4096, 128, 4, 512, True, torch.bfloat16), (1, 2, 2, 300, 128, 0, 0, False, torch.bfloat16),
    (1, 2, 1, 1025, 128, 200, 1, True, torch.bfloat16),
]:
    q = torch.randn(B, Hq, N, D, device="cuda", dtype=dt)
    k = torch.randn(B, Hkv, N, D, device="cuda", dtype=dt)
    v = torch.randn(B, Hkv, N, D, device="cuda", dtype=dt)
    do = torch.randn(B, Hq, N, D, device="cuda", dtype=dt)
    sa = torch.randn(Hq, device="cuda") * 0.5 if aux else None
    a = grads(q, k, v, do, ns, W, sa, "0")
    b = grads(q, k, v, do, ns, W, sa, "64")
    for nm in (0, 1, 2):
        os.environ["SFA_DQ_GEMM_NW"] = "4"
    dq = (a[0].float() - b[0].float()).abs().max().item()
    dk = (a[1].float() - b[1].float()).abs().max().item()
    dv = (a[2].float() - b[2].float()).abs().max().item()
    ref = a[0].float().abs().max().item()
    ok = dk == 0 and dv == 0 and dq <= 2e-2 * max(ref, 1.0) and not torch.isnan(b[0]).any()
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} B{B} Hq{Hq} Hkv{Hkv} N{N} D{D} ns{ns} W{W} {str(dt)[6:]}: |ddq| {dq:.3e} (max|dq| {ref:.2f}) "
          f"|ddk| {dk:.1e} |ddv| {dv:.1e}  {b[4]}")
print("mismatches:", bad)

if len(sys.argv) > 1 and sys.argv[1] == "time":
    B, Hq, Hkv, N, D, ns, W = 4, 32, 8, 8192, 128, 4, 4096
    q = torch.randn(B, Hq, N, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(B, Hkv, N, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(B, Hkv, N, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    do = torch.randn(B, Hq, N, D, device="cuda", dtype=torch.bfloat16)
    for gb in ("0", "64"):
        os.environ["SINK_ATTENTION_DS_SPILL_GB"] = gb
        for it in range(3):
            o = sink_flash_attention(q, k, v, ns, W)
            o.backward(do)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        o = sink_flash_attention(q, k, v, ns, W)
        e0.record()
        for it in range(10):
            o.backward(do, retain_graph=True)
        e1.record()
        torch.cuda.synchronize()
        print(f"spill cap {gb} GB: backward {e0.elapsed_time(e1) / 10:.3f} ms  {_native.last_path()}")
