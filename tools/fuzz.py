#!/usr/bin/env python3
"""Randomised cross-check on the GPU: MFMA kernels against the exact-f32 kernels of the same library (and the
N_q < N_kv / packed paths against their padded / per-sequence formulations) on random shapes.
usage: python tools/fuzz.py [n_cases] [seed] [skew]      (skew: only shapes of the short-window dK/dV kernel - no sink keys,
head dims 64 / 80 / 96, windows up to 512, N_q = N_kv or packed)"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sink-flash-attention-kernel_amd"), ROOT]
import torch

from sink_attention.sink_flash_attention import _sink_flash_attention_ex
from sink_attention.varlen import sink_flash_attention_varlen

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.manual_seed(rng.randrange(1 << 30))
focus_skew = len(sys.argv) > 3 and sys.argv[3] == "skew"
bad = 0


def run(q, k, v, do, ns, W, sa, generic):
    qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
    ss = sa.clone().requires_grad_(True) if sa is not None else None
    o = _sink_flash_attention_ex(qq, kk, vv, ns, W, s_aux=ss, force_generic=generic)
    o.backward(do)
    return [o.detach().float(), qq.grad.float(), kk.grad.float(), vv.grad.float()] + ([ss.grad.float()] if ss is not None else [])


for case in range(n_cases):
    D = rng.choice([32, 64, 80, 96, 128, 128, 256])
    Hkv = rng.choice([1, 2, 4])
    g = rng.choice([1, 2, 4, 8])
    Hq = Hkv * g
    B = rng.choice([1, 2])
    N = rng.choice([1, 7, 31, 33, 64, 65, 100, 127, 128, 129, 200, 257, 500, 777, 1024, 1500, 2500, 4100])
    ns = rng.choice([0, 1, 4, 63, 64, 65, 130])
    W = rng.choice([0, 1, 5, 31, 64, 100, 128, 300, 1000, 4096])
    dt = rng.choice([torch.bfloat16, torch.float16])
    aux = rng.random() < 0.6
    mode = rng.choice(["plain", "plain", "offset", "varlen"])
    if focus_skew:
        D, ns, W = rng.choice([64, 80, 96]), 0, rng.choice([1, 5, 31, 64, 100, 128, 300, 512])
        mode = rng.choice(["plain", "plain", "varlen"])
    sa = (torch.randn(Hq, device="cuda") * 0.5) if aux else None
    tol_o, tol_g = (2e-2, 2e-1) if dt == torch.bfloat16 else (5e-3, 6e-2)
    desc = f"{mode} B{B} Hq{Hq} Hkv{Hkv} N{N} D{D} ns{ns} W{W} {str(dt)[6:]} aux{int(aux)}"
    try:
        if mode == "plain":
            q = torch.randn(B, Hq, N, D, device="cuda", dtype=dt)
            k = torch.randn(B, Hkv, N, D, device="cuda", dtype=dt)
            v = torch.randn(B, Hkv, N, D, device="cuda", dtype=dt)
            do = torch.randn(B, Hq, N, D, device="cuda", dtype=dt)
            a, b = run(q, k, v, do, ns, W, sa, False), run(q, k, v, do, ns, W, sa, True)
        elif mode == "offset":
            Nk = N + rng.choice([1, 17, 64, 100, 300])
            q = torch.randn(B, Hq, N, D, device="cuda", dtype=dt)
            k = torch.randn(B, Hkv, Nk, D, device="cuda", dtype=dt)
            v = torch.randn(B, Hkv, Nk, D, device="cuda", dtype=dt)
            do = torch.randn(B, Hq, N, D, device="cuda", dtype=dt)
            a = run(q, k, v, do, ns, W, sa, False)
            qp = torch.cat([torch.zeros(B, Hq, Nk - N, D, device="cuda", dtype=dt), q], 2)
            dop = torch.cat([torch.zeros(B, Hq, Nk - N, D, device="cuda", dtype=dt), do], 2)
            b = run(qp, k, v, dop, ns, W, sa, True)
            b[0], b[1] = b[0][:, :, Nk - N:], b[1][:, :, Nk - N:]
            desc += f" Nk{Nk}"
        else:
            nseq = rng.choice([1, 2, 3, 5])
            lens = [rng.choice([0, 1, 30, 64, 65, 200, 500]) for _ in range(nseq)]
            if sum(lens) == 0:
                lens[0] = 10
            cu = [0]
            for L in lens:
                cu.append(cu[-1] + L)
            T = cu[-1]
            q = torch.randn(1, Hq, T, D, device="cuda", dtype=dt)
            k = torch.randn(1, Hkv, T, D, device="cuda", dtype=dt)
            v = torch.randn(1, Hkv, T, D, device="cuda", dtype=dt)
            do = torch.randn(1, Hq, T, D, device="cuda", dtype=dt)
            qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
            ss = sa.clone().requires_grad_(True) if aux else None
            o = sink_flash_attention_varlen(qq, kk, vv, cu, ns, W, ss)
            o.backward(do)
            a = [o.detach().float(), qq.grad.float(), kk.grad.float(), vv.grad.float()] + ([ss.grad.float()] if aux else [])
            outs = [torch.zeros_like(x) for x in a]
            for s0, s1 in zip(cu[:-1], cu[1:]):
                if s1 > s0:
                    r = run(q[:, :, s0:s1], k[:, :, s0:s1], v[:, :, s0:s1], do[:, :, s0:s1], ns, W, sa, True)
                    for j in range(4):
                        outs[j][:, :, s0:s1] = r[j]
                    if aux:
                        outs[4] += r[4]
            b = outs
            desc += f" lens{lens}"
        errs = [(x - y).abs().max().item() if x.numel() else 0.0 for x, y in zip(a, b)]
        nan = any(torch.isnan(x).any().item() for x in a)
        lim = [tol_o, tol_g, tol_g, tol_g, tol_g * 20]
        ok = not nan and all(e <= l * max(1.0, y.abs().max().item() if y.numel() else 1.0) for e, l, y in zip(errs, lim, b))
    except Exception as e:      # noqa: BLE001 - report and continue
        ok, errs = False, [repr(e)[:200]]
    if not ok:
        bad += 1
        print("BAD", desc, errs)
    elif case % 100 == 0:
        print("ok ", desc, ["%.2e" % e for e in errs])
print(f"fuzz: {n_cases} cases, {bad} bad")
