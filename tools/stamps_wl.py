#!/usr/bin/env python3
"""Per-workgroup life times of a work-list (persistent) kernel: library built with EXTRA=-DSFA_WL_STAMPS, every workgroup
records s_memrealtime (100 MHz) at entry and exit, its XCC and its item count.  Shows how well the static lists balance.
usage: python tools/stamps_wl.py [--cfg C3] [--kernel dq|fwd]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sink-flash-attention-kernel_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch

from kbench import CFG
from sink_attention import _native, sink_flash_attention


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="C3")
    ap.add_argument("--kernel", default="dq")
    ap.add_argument("--phases", action="store_true", help="library built with ASMGEN_DQPK_STAMPS=1: cycles of an item transition")
    ap.add_argument("--set", default="", help="which:value pairs for sfa_debug_set_variant, '+'-joined")
    args = ap.parse_args()
    B, Hq, Hkv, N, D, ns, W, aux = CFG[args.cfg]
    dev = "cuda"
    torch.manual_seed(1)
    q = torch.randn(B, Hq, N, D, device=dev, dtype=torch.bfloat16).requires_grad_(True)
    k = torch.randn(B, Hkv, N, D, device=dev, dtype=torch.bfloat16).requires_grad_(True)
    v = torch.randn(B, Hkv, N, D, device=dev, dtype=torch.bfloat16).requires_grad_(True)
    do = torch.randn_like(q)
    lib = _native.lib()
    for kv in filter(None, args.set.split("+")):
        kn, val = (int(x) for x in kv.split(":"))
        lib.sfa_debug_set_variant(kn, val)
    G = 8192
    dbg = torch.zeros(G * 4, dtype=torch.int32, device=dev)
    for it in range(4):
        dbg.zero_()
        out = sink_flash_attention(q, k, v, num_sink=ns, window_size=W)
        if args.kernel == "fwd":
            lib.sfa_debug_set_ptr(dbg.data_ptr())
            out = sink_flash_attention(q, k, v, num_sink=ns, window_size=W)
            lib.sfa_debug_set_ptr(None)
        else:
            lib.sfa_debug_set_ptr(dbg.data_ptr())
            out.backward(do)
            lib.sfa_debug_set_ptr(None)
        torch.cuda.synchronize()
        q.grad = k.grad = v.grad = None
    print(_native.last_path())
    if args.phases:
        d = dbg.view(G, 4).cpu().long() & 0xFFFFFFFF
        d = d[(d[:, 3] & 1023) > 1]
        nt = ((d[:, 3] & 1023) - 1).float()
        a0, a1, a2 = (d[:, i].float() / nt for i in range(3))
        print("per item transition (cycles, mean over %d waves): loop end -> next item's requests issued %.0f | -> finished item's "
              "stores issued %.0f | -> next item's data landed, first K fragments requested %.0f" % (len(d), a0.mean(), a1.mean(), a2.mean()))
        return
    d = dbg.view(G, 4).cpu().long() & 0xFFFFFFFF
    live = (d[:, 3] & 1023) > 0
    d = d[live]
    t0, t1, xcc, n = d[:, 0], d[:, 1], d[:, 2], d[:, 3] & 1023
    cyc = ((d[:, 3] >> 10) << 4).float()         # shader cycles (s_memtime) of the workgroup's life
    start, end = t0.min().item(), t1.max().item()
    span = (end - start) / 100.0
    life = ((t1 - t0) & 0xFFFFFFFF).float() / 100.0
    idle = (end - t1).float() / 100.0
    late = (t0 - start).float() / 100.0
    print("workgroups %d  kernel span %.1f us | life: mean %.1f min %.1f max %.1f us | idle at the end: mean %.1f max %.1f us (%.1f %% of the span) | "
          "start skew: mean %.1f max %.1f us" % (len(d), span, life.mean(), life.min(), life.max(), idle.mean(), idle.max(),
                                               100 * idle.mean() / span, late.mean(), late.max()))
    print("shader clock held during the kernel (cycles / life, mean over the workgroups): %.0f MHz" % (cyc / life).mean().item())
    for x in range(8):
        m = xcc == x
        if m.any():
            print("  XCC %d: %3d workgroups, items %d..%d, life mean %.1f min %.1f max %.1f us, sum %.0f us, last end %.1f us, clock %.0f MHz" % (
                x, int(m.sum()), n[m].min(), n[m].max(), life[m].mean(), life[m].min(), life[m].max(), life[m].sum(),
                (t1[m].max().item() - start) / 100.0, (cyc[m] / life[m]).mean().item()))


if __name__ == "__main__":
    main()
