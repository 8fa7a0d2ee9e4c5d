#!/usr/bin/env python3
"""Reproduce the reference's PUBLISHED tables (its README, collected in BASELINE.md; NVIDIA H200 there) on MI355X with
the reference's own methodology: B=1, H_q=32, H_kv=8, D=128, num_sink=4, W=4096, fp16;
  * fwd / fwd+bwd latency: median over 20 calls, perf_counter around a synchronised call (tests/benchmark.py:126-143)
  * decode latency: 100 back-to-back calls / 100 (tests/run_inference_benchmarks.py:204-249)
  * cache update + decode: 200 back-to-back calls each (:287-339), plus this library's copy-free append + ring decode
Not the contract bench (bench.py); an apples-to-apples table for DESIGN.md / profiles."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sink-flash-attention-kernel_amd"), ROOT]
import torch

from oracle.sink_oracle import pair_count_closed
from sink_attention import SinkCacheLayer, sink_decode_attention, sink_flash_attention

REF_FWD = {512: 0.07, 1024: 0.12, 2048: 0.27, 4096: 0.76, 8192: 1.88, 16384: 4.18, 32768: 8.77}
REF_FB = {512: 0.41, 1024: 0.47, 2048: 0.97, 4096: 2.81, 8192: 7.28, 16384: 16.39}
REF_DEC = {(32, 132): 0.056, (32, 516): 0.057, (32, 1028): 0.056, (32, 2052): 0.069, (32, 4100): 0.119,
           (64, 1028): 0.068, (64, 4100): 0.209}


def median_sync(fn, warmup=5, repeat=20):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(repeat):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def back_to_back(fn, warmup, repeats):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(repeats):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / repeats * 1e3


def main():
    torch.manual_seed(0)
    B, Hq, Hkv, D, ns, W = 1, 32, 8, 128, 4, 4096
    dt = torch.float16
    print(f"# MI355X, B={B} H_q={Hq} H_kv={Hkv} D={D} num_sink={ns} W={W} fp16; reference column = H200 (its README)")
    print("## forward / forward+backward latency (ms, median of 20 synchronised calls)")
    print(f"{'N':>6} {'fwd':>8} {'TF/s':>7} {'ref fwd':>8} {'x':>5} | {'fwd+bwd':>8} {'TF/s':>7} {'ref f+b':>8} {'x':>5}")
    for N in (512, 1024, 2048, 4096, 8192, 16384, 32768):
        q = torch.randn(B, Hq, N, D, device="cuda", dtype=dt, requires_grad=True)
        k = torch.randn(B, Hkv, N, D, device="cuda", dtype=dt, requires_grad=True)
        v = torch.randn(B, Hkv, N, D, device="cuda", dtype=dt, requires_grad=True)
        do = torch.randn(B, Hq, N, D, device="cuda", dtype=dt)
        pairs = pair_count_closed(N, ns, W)
        with torch.no_grad():
            tf = median_sync(lambda: sink_flash_attention(q, k, v, ns, W))

        def fb():
            o = sink_flash_attention(q, k, v, ns, W)
            o.backward(do)
            q.grad = k.grad = v.grad = None
        tb = median_sync(fb)
        ff, fbw = 4 * D * pairs * B * Hq, 14 * D * pairs * B * Hq
        rf, rb = REF_FWD.get(N), REF_FB.get(N)
        print(f"{N:>6} {tf:>8.3f} {ff / tf / 1e9:>7.1f} {rf if rf else float('nan'):>8.2f} {rf / tf if rf else float('nan'):>5.1f} | "
              f"{tb:>8.3f} {fbw / tb / 1e9:>7.1f} {rb if rb else float('nan'):>8.2f} {rb / tb if rb else float('nan'):>5.1f}")

    print("## decode latency (ms, 100 back-to-back calls)")
    print(f"{'H_q':>4} {'N_kv':>6} {'decode':>8} {'ref':>7} {'x':>5}")
    for (hq, nkv), ref in REF_DEC.items():
        q = torch.randn(1, hq, 1, D, device="cuda", dtype=dt)
        k = torch.randn(1, Hkv, nkv, D, device="cuda", dtype=dt)
        v = torch.randn(1, Hkv, nkv, D, device="cuda", dtype=dt)
        t = back_to_back(lambda: sink_decode_attention(q, k, v), 10, 100)
        print(f"{hq:>4} {nkv:>6} {t:>8.4f} {ref:>7.3f} {ref / t:>5.1f}")

    print("## cache update + decode attention (ms, 200 back-to-back calls each)")
    for hq, win, ref in ((32, 1024, None), (32, 4096, (0.081, 0.120)), (64, 4096, None)):
        layer = SinkCacheLayer(ns, win)
        kp = torch.randn(1, Hkv, ns + win, D, device="cuda", dtype=dt)
        vp = torch.randn(1, Hkv, ns + win, D, device="cuda", dtype=dt)
        layer.update(kp, vp)
        q = torch.randn(1, hq, 1, D, device="cuda", dtype=dt)
        kn = torch.randn(1, Hkv, 1, D, device="cuda", dtype=dt)
        vn = torch.randn(1, Hkv, 1, D, device="cuda", dtype=dt)
        upd = back_to_back(lambda: layer.update(kn, vn), 10, 200)           # reference-style: linearised copy
        kc, vc = layer.get_kv()
        att = back_to_back(lambda: sink_decode_attention(q, kc, vc), 10, 200)

        def fused():
            layer.append(kn, vn)                                           # one ring slot, no torch.cat
            layer.decode_attention(q)                                      # sfa_decode_ring reads the ring in place
        fu = back_to_back(fused, 10, 200)
        st = back_to_back(lambda: layer.decode_step(q, kn, vn), 10, 200)   # sfa_decode_ring_step: one pass
        r = f"ref {ref[0]:.3f} + {ref[1]:.3f} = {sum(ref):.3f}" if ref else "ref n/a"
        print(f"GQA({hq}/8) win={win}: update {upd:.4f} + decode {att:.4f} = {upd + att:.4f} ms ({r}); "
              f"copy-free append+ring decode {fu:.4f} ms; fused decode_step {st:.4f} ms")


def graph_step():
    """A whole generation step (24 layers, B=1, GQA 32/8, W=4096): eager fused steps vs ONE replay of a captured hipGraph
    (device-resident cache state, SinkCacheLayer.decode_step_dyn)."""
    dt, L, Hq, Hkv, D, ns, win = torch.float16, 24, 32, 8, 128, 4, 4096
    eager = [SinkCacheLayer(ns, win) for _ in range(L)]
    dyn = [SinkCacheLayer(ns, win) for _ in range(L)]
    for a, b in zip(eager, dyn):
        kp = torch.randn(1, Hkv, ns + win, D, device="cuda", dtype=dt)
        vp = torch.randn(1, Hkv, ns + win, D, device="cuda", dtype=dt)
        a.update(kp, vp)
        b.update(kp, vp)
        b.enable_device_state()
    q = torch.randn(1, Hq, 1, D, device="cuda", dtype=dt)
    kn = torch.randn(1, Hkv, 1, D, device="cuda", dtype=dt)
    vn = torch.randn(1, Hkv, 1, D, device="cuda", dtype=dt)
    outs = [torch.empty_like(q) for _ in range(L)]
    te = back_to_back(lambda: [l.decode_step(q, kn, vn) for l in eager], 5, 50)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i, l in enumerate(dyn):
            l.decode_step_dyn(q, kn, vn, out=outs[i])
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i, l in enumerate(dyn):
            l.decode_step_dyn(q, kn, vn, out=outs[i])
    tg = back_to_back(g.replay, 5, 50)
    print(f"## one generation step, {L} layers (cache update + attention per layer): eager fused steps {te:.3f} ms "
          f"({te / L * 1e3:.1f} us/layer); captured hipGraph replay {tg:.3f} ms ({tg / L * 1e3:.1f} us/layer); "
          f"reference-style update + decode would be {L} x 0.201 = {L * 0.201:.2f} ms on its H200")


if __name__ == "__main__":
    main()
    graph_step()
