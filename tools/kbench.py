#!/usr/bin/env python3
"""Kernel micro-bench (exploration tool, not the contract bench): times sink_flash_attention fwd / fwd+bwd and
sink_decode_attention at the BASELINE.json configs with HIP events on torch's current stream.
usage: python tools/kbench.py [fwd] [bwd] [decode] [--cfg C2,C3,C4] [--iters 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sink-flash-attention-kernel_amd"), ROOT]
import torch

from oracle.sink_oracle import pair_count_closed
from sink_attention import _native, sink_decode_attention, sink_flash_attention

CFG = {  # name: B, Hq, Hkv, N, D, ns, W, s_aux
    "C2": (4, 32, 32, 4096, 128, 4, 1024, False),
    "C3": (4, 32, 8, 8192, 128, 4, 4096, False),
    "C3mha": (4, 32, 32, 8192, 128, 4, 4096, False),
    "C4": (1, 64, 8, 8192, 80, 0, 128, True),
    "C4b4": (4, 64, 8, 8192, 80, 0, 128, True),
    "D64": (4, 32, 8, 8192, 64, 4, 4096, False),
    "causal": (4, 32, 8, 8192, 128, 0, 8192, False),
    "D256": (4, 16, 4, 8192, 256, 4, 4096, False),
    "C2gqa": (4, 32, 8, 4096, 128, 4, 1024, False),
    "W128d128": (4, 32, 8, 8192, 128, 4, 128, False),
    "refB1N32k": (1, 32, 8, 32768, 128, 4, 4096, False),
    "refB1N8k": (1, 32, 8, 8192, 128, 4, 4096, False),
    "refB1N16k": (1, 32, 8, 16384, 128, 4, 4096, False),
    "tp8N8k": (1, 4, 1, 8192, 128, 4, 4096, False),
    "tp8N32k": (1, 4, 1, 32768, 128, 4, 4096, False),
    "tp4N8k": (1, 8, 2, 8192, 128, 4, 4096, False),
    "oss_tp8_swa": (1, 8, 1, 8192, 64, 0, 128, True),
    "oss_tp8_full": (1, 8, 1, 8192, 64, 0, 8192, True),
    "oss_swa": (1, 64, 8, 8192, 64, 0, 128, True),
    "oss_full": (1, 64, 8, 8192, 64, 0, 8192, True),
    "refB2N8k": (2, 32, 8, 8192, 128, 4, 4096, False),
    "refB1N512": (1, 32, 8, 512, 128, 4, 4096, False),
    "refB1N1k": (1, 32, 8, 1024, 128, 4, 4096, False),
    "refB1N2k": (1, 32, 8, 2048, 128, 4, 4096, False),
    "refB1N4k": (1, 32, 8, 4096, 128, 4, 4096, False),
    "slmB1": (1, 32, 8, 16384, 128, 4, 1024, False),
    "slmB1d64": (1, 32, 8, 16384, 64, 4, 256, False),
    "D32": (4, 32, 8, 8192, 32, 4, 4096, False),
    "defaults": (4, 32, 8, 8192, 128, 4, 512, False),          # the reference's default arguments (num_sink=4, window_size=512)
    "defaults_mha": (4, 32, 32, 8192, 128, 4, 512, False),
}


def timeit(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["fwd"])
    ap.add_argument("--cfg", default="C2,C3,C4")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype]
    dev = "cuda"
    for name in args.cfg.split(","):
        if not name or name not in CFG:
            continue
        B, Hq, Hkv, N, D, ns, W, aux = CFG[name]
        torch.manual_seed(42)
        q = torch.randn(B, Hq, N, D, device=dev, dtype=dt)
        k = torch.randn(B, Hkv, N, D, device=dev, dtype=dt)
        v = torch.randn(B, Hkv, N, D, device=dev, dtype=dt)
        sa = (torch.randn(Hq, device=dev) * 0.5) if aux else None
        pairs = pair_count_closed(N, ns, W)
        f_fwd = 4 * D * pairs * B * Hq
        if "fwd" in args.what:
            med, mn = timeit(lambda: sink_flash_attention(q, k, v, ns, W, sa), args.iters)
            io = (2 * q.numel() + 2 * k.numel()) * 2
            print(f"{name} fwd  {_native.last_path():34s} med {med:8.3f} ms  min {mn:8.3f} ms  "
                  f"{f_fwd / med / 1e9:8.1f} TFLOP/s ({f_fwd / med / 1e9 / 2516.6 * 100:5.1f}% peak)  "
                  f"min-IO {io / med / 1e6:7.1f} GB/s", flush=True)
        if "bwd" in args.what:
            qg, kg, vg = (t.clone().requires_grad_(True) for t in (q, k, v))
            sag = sa.clone().requires_grad_(True) if aux else None
            do = torch.randn_like(q)

            def step():
                o = sink_flash_attention(qg, kg, vg, ns, W, sag)
                o.backward(do)
                qg.grad = kg.grad = vg.grad = None
            med, mn = timeit(step, args.iters)
            f = 3.5 * f_fwd
            print(f"{name} f+b  {_native.last_path():34s} med {med:8.3f} ms  min {mn:8.3f} ms  "
                  f"{f / med / 1e9:8.1f} TFLOP/s ({f / med / 1e9 / 2516.6 * 100:5.1f}% peak)", flush=True)
    if "decode" in args.what:
        for (B, Hq, Hkv, Nkv, D) in [(32, 32, 32, 131072, 128), (32, 32, 32, 4100, 128), (32, 32, 8, 131072, 128),
                                     (1, 32, 8, 4100, 128), (1, 64, 8, 4100, 128), (4, 64, 8, 4100, 128), (16, 64, 8, 4100, 128),
                                     (1, 64, 8, 4100, 64), (1, 32, 8, 132, 128)]:
            q = torch.randn(B, Hq, 1, D, device=dev, dtype=dt)
            k = torch.randn(B, Hkv, Nkv, D, device=dev, dtype=dt)
            v = torch.randn(B, Hkv, Nkv, D, device=dev, dtype=dt)
            med, mn = timeit(lambda: sink_decode_attention(q, k, v), args.iters)
            byts = 2 * k.numel() * 2 + 2 * q.numel() * 2
            print(f"decode B{B} Hq{Hq} Hkv{Hkv} N{Nkv} {_native.last_path():28s} med {med:8.4f} ms min {mn:8.4f} ms "
                  f"{byts / med / 1e6:8.1f} GB/s ({byts / med / 1e6 / 8000 * 100:5.1f}% of 8 TB/s)", flush=True)
            # the raw C call with a caller-owned workspace: two launches, and SFA_FLAG_DECODE_ONE_PASS (one launch)
            lib = _native.lib()
            out = torch.empty_like(q)
            nb = int(lib.sfa_decode_workspace_bytes(B, Hq, Hkv, Nkv, D, _native.SFA_DTYPE[q.dtype]))
            ws = torch.zeros(max(nb, 256), device=dev, dtype=torch.uint8)
            for flag, nm in ((0, "raw 2 launches"), (_native.FLAG_DECODE_ONE_PASS, "raw one pass")):
                def call():
                    st = lib.sfa_decode(_native.desc(q), _native.desc(k), _native.desc(v), _native.desc(out), None, ws.data_ptr(),
                                        ws.numel(), 1.0 / D ** 0.5, flag, _native.stream_ptr(q.device))
                    assert st == 0
                ref = sink_decode_attention(q, k, v)
                call()
                err = (out.float() - ref.float()).abs().max().item()
                med, mn = timeit(call, args.iters)
                print(f"   {nm:16s} {_native.last_path():34s} med {med:8.4f} ms min {mn:8.4f} ms  max|diff| {err:.2e}", flush=True)
            del q, k, v, ws


if __name__ == "__main__":
    main()
