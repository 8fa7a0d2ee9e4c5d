#!/usr/bin/env python3
"""Randomised cross-check of the decode paths on the GPU: plain decode vs a float64 torch reference, and ring /
fused-step / device-state / one-pass variants against the linearised cache.  usage: python tools/fuzz_decode.py [n] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sink-flash-attention-kernel_amd"), ROOT]
import torch

from sink_attention import sink_decode_attention
from sink_attention.cache import SinkCacheLayer

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.manual_seed(rng.randrange(1 << 30))
bad = 0


def ref_decode(q, k, v, sa):
    B, Hq, _, D = q.shape
    g = Hq // k.shape[1]
    kk, vv = k.double().repeat_interleave(g, 1), v.double().repeat_interleave(g, 1)
    s = (q.double() @ kk.transpose(-1, -2)) / D ** 0.5
    if sa is not None:
        s = torch.cat([s, sa.double().view(1, Hq, 1, 1).expand(B, Hq, 1, 1)], -1)
    p = torch.softmax(s, -1)
    return (p[..., :kk.shape[2]] @ vv)


for case in range(n_cases):
    D = rng.choice([32, 64, 80, 128, 256])
    Hkv = rng.choice([1, 2, 8])
    g = rng.choice([1, 2, 4, 8, 16])
    B = rng.choice([1, 2, 5])
    dt = rng.choice([torch.bfloat16, torch.float16, torch.float32])
    ns, W = rng.choice([0, 1, 4]), rng.choice([1, 7, 64, 300])
    pre = rng.choice([1, 3, 50, 400])
    aux = rng.random() < 0.5
    Hq = Hkv * g
    sa = torch.randn(Hq, device="cuda") * 0.5 if aux else None
    tol = 1e-4 if dt == torch.float32 else (2e-2 if dt == torch.bfloat16 else 4e-3)
    desc = f"B{B} Hq{Hq} Hkv{Hkv} D{D} ns{ns} W{W} pre{pre} {str(dt)[6:]} aux{int(aux)}"
    try:
        a, b, c = SinkCacheLayer(ns, W), SinkCacheLayer(ns, W), SinkCacheLayer(ns, W)
        c.one_pass = True
        kp, vp = torch.randn(B, Hkv, pre, D, device="cuda", dtype=dt), torch.randn(B, Hkv, pre, D, device="cuda", dtype=dt)
        for l in (a, b, c):
            l.update(kp, vp)
        b.enable_device_state()
        ok = True
        for step in range(rng.choice([1, 3, W + 3])):
            q = torch.randn(B, Hq, 1, D, device="cuda", dtype=dt)
            kn, vn = torch.randn(B, Hkv, 1, D, device="cuda", dtype=dt), torch.randn(B, Hkv, 1, D, device="cuda", dtype=dt)
            o1 = a.decode_step(q, kn, vn, s_aux=sa)
            o2 = b.decode_step_dyn(q, kn, vn, s_aux=sa)
            o3 = c.decode_step(q, kn, vn, s_aux=sa)
            kc, vc = a.get_kv()
            o4 = sink_decode_attention(q, kc, vc, s_aux=sa)
            r = ref_decode(q, kc, vc, sa)
            e = [(o.double() - r).abs().max().item() for o in (o1, o2, o3, o4)]
            # (the device-state step sizes its launch for the FULL cache, so while the ring is still filling its split
            # plan, hence its summation order, may differ from the host-state step: tolerance, not bitwise)
            if max(e) > tol or any(torch.isnan(o).any() for o in (o1, o2, o3, o4)):
                ok = False
                desc += f" step{step} errs{e}"
                break
        b.pull_state()
        if ok and (a.write_pos, a.window_len) != (b.write_pos, b.window_len):
            ok, desc = False, desc + " state mismatch"
    except Exception as ex:      # noqa: BLE001
        ok, desc = False, desc + " " + repr(ex)[:200]
    if not ok:
        bad += 1
        print("BAD", desc)
print(f"fuzz_decode: {n_cases} cases, {bad} bad")
