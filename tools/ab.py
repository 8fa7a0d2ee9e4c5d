#!/usr/bin/env python3
"""A/B timing of two co-compiled kernel bodies in ONE process on ONE device (library built with -DSFA_AB):
alternates sfa_debug_set_variant(which, 0 / 1) round by round and reports the per-kernel HIP-event times.
usage: python tools/ab.py [--which 0] [--rounds 12] [--cfg C3]        which: 0 dK/dV, 1 forward, 2 dQ"""
import argparse
import ctypes
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
    ap.add_argument("--which", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--cfg", default="C3")
    ap.add_argument("--values", default="0,1", help="the two values of the knob to compare")
    ap.add_argument("--combos", default="", help="instead of --which/--values: comma-separated settings, each a '+'-joined "
                    "list of which:value pairs, e.g. 5:1,5:0+6:5,5:0+6:9 (knobs not named in a setting are reset to 0)")
    args = ap.parse_args()
    B, Hq, Hkv, N, D, ns, W, aux = CFG[args.cfg]
    dev = "cuda"
    torch.manual_seed(1)
    q = torch.randn(B, Hq, N, D, device=dev, dtype=torch.bfloat16).requires_grad_(True)
    k = torch.randn(B, Hkv, N, D, device=dev, dtype=torch.bfloat16).requires_grad_(True)
    v = torch.randn(B, Hkv, N, D, device=dev, dtype=torch.bfloat16).requires_grad_(True)
    do = torch.randn_like(q)
    lib = _native.lib()
    vals = [int(x) for x in args.values.split(",")]
    combos = None
    if args.combos:
        combos = [[tuple(int(y) for y in kv.split(":")) for kv in c.split("+")] for c in args.combos.split(",")]
        vals = list(range(len(combos)))
        knobs = sorted({kn for c in combos for kn, _ in c})
    res = {v: [] for v in vals}
    grads = {}
    for r in range(args.rounds + 2):
        for var in vals:
            if combos:
                for kn in knobs:
                    lib.sfa_debug_set_variant(kn, 0)
                for kn, val in combos[var]:
                    lib.sfa_debug_set_variant(kn, val)
            else:
                lib.sfa_debug_set_variant(args.which, var)
            ev = HipEvents(4)
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            lib.sfa_debug_set_stage_events(ev.ev, 4)
            f0.record()
            out = sink_flash_attention(q, k, v, num_sink=ns, window_size=W)
            f1.record()
            out.backward(do)
            lib.sfa_debug_set_stage_events(None, 0)
            torch.cuda.synchronize()
            if r >= 2:
                res[var].append((f0.elapsed_time(f1), ev.elapsed(1, 2), ev.elapsed(2, 3)))
            grads[var] = (out.detach().clone(), q.grad.clone(), k.grad.clone(), v.grad.clone())
            q.grad = k.grad = v.grad = None
    lib.sfa_debug_set_variant(args.which, 0)
    if combos:
        for kn in knobs:
            lib.sfa_debug_set_variant(kn, 0)
    same = all(torch.equal(a, b) for v2 in vals[1:] for a, b in zip(grads[vals[0]], grads[v2]))
    for var in vals:
        cols = list(zip(*res[var]))
        med = [sorted(c)[len(c) // 2] for c in cols]
        mn = [min(c) for c in cols]
        if combos:
            print("setting", "+".join("%d:%d" % kv for kv in combos[var]), end=" -> ")
        print(f"variant {var}: fwd med {med[0]:.4f} min {mn[0]:.4f} | dkdv med {med[1]:.4f} min {mn[1]:.4f} | dq med {med[2]:.4f} "
              f"min {mn[2]:.4f} ms  (path {_native.last_path()})")
    print("results bitwise equal between variants:", same)


if __name__ == "__main__":
    main()
