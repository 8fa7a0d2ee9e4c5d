#!/usr/bin/env python3
"""Cycle anatomy of the hand-placed dK/dV kernel (library built by `tools/build_ab.sh '{}' stamps`: the ALT body
carries s_memtime stamps around the loop head).  Prints, per workgroup class, cycles per trip spent in the trip body and
in the loop head (scalar code + vmcnt wait + barrier + lgkmcnt wait), and the kernel time of both bodies.
usage: python tools/stamps_dkdv.py [--cfg C3]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sink-flash-attention-kernel_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch

from bench import HipEvents
from kbench import CFG
from sink_attention import _native, sink_flash_attention


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="C3")
    args = ap.parse_args()
    B, Hq, Hkv, N, D, ns, W, aux = CFG[args.cfg]
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
    d = dbg.view(nblk, 4, 4).cpu().long()          # [bid][wave][top, body, trips, -]
    top, body, trips = d[:, :, 0], d[:, :, 1], d[:, :, 2]
    live = trips[:, 0] > 0
    print("workgroups with trips:", int(live.sum()), "of", nblk, " total trips:", int(trips[:, 0].sum()))
    tt = trips[live].float()
    print("cycles per trip, all waves: body %.0f  head %.0f" % ((body[live].float().sum() / tt.sum()).item(),
                                                                  (top[live].float().sum() / tt.sum()).item()))
    for w in range(4):
        print("  wave %d: body %.0f head %.0f" % (w, (body[live][:, w].float().sum() / tt[:, w].sum()).item(),
                                                   (top[live][:, w].float().sum() / tt[:, w].sum()).item()))
    # by workgroup length
    for lo, hi in ((1, 200), (200, 520), (520, 600), (600, 2000)):
        m = live & (trips[:, 0] >= lo) & (trips[:, 0] < hi)
        if m.any():
            t = trips[m].float().sum()
            print("  workgroups with %4d..%4d trips (%3d): body %.0f head %.0f per trip" % (
                lo, hi, int(m.sum()), (body[m].float().sum() / t).item(), (top[m].float().sum() / t).item()))


if __name__ == "__main__":
    main()
