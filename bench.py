#!/usr/bin/env python3
"""Contract benchmark: sink flash attention fwd+bwd at BASELINE.json's metric shape on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload ("C3", BASELINE.json configs[2], the shape the metric is quoted on):
    bf16, B=4 per GPU, H_q=32, H_kv=8, N=8192, D=128, num_sink=4, window=4096, synthetic randn inputs.
A step = one forward + one backward of sink_flash_attention through the product op (C ABI -> HIP kernels) with
inputs resident in HBM.  (batch, KV head) units are independent, so ranks just own disjoint batches: no
data-path collective, weak scaling, value = all ranks' algorithmic FLOPs / max-over-ranks time.

Algorithmic FLOPs: fwd+bwd = 14 * D * pairs(N, ns, W) * B * H_q; masked-out work,
recomputation and padding do not count.

Extra objects on the JSON line:
  roofline      the dominant kernel of the step (the longest of: forward kernel, backward dK/dV kernel, backward dQ
                kernel), its algorithmic FLOPs / its average duration measured with HIP events on the launching
                stream inside the timed steps (sfa_debug_set_stage_events for the backward stages), against the
                gfx950 dense bf16 MFMA peak 2516.6 TFLOP/s.  per_kernel: ms / algorithmic TFLOP/s / fraction of peak of
                all three kernels (fwd_frac repeats the forward's, north_star's 60 % target).  traffic: HBM bytes per
                launch from the committed rocprofv3 PMC run (profiles/*_pmc.json) if present, else null;
                traffic_source names that file (it is NOT measured inside this run).
  cpu_baseline  the reference's eager fp32 algorithm (oracle/sink_oracle.py restatement, dense N x N, torch on all
                host cores) timed on a bounded slice of the same workload (1 batch element x 1 KV group = 4 q heads),
                rank 0 at N=1 only: one warm-up repetition, then the median of three.
  sustained     the same step run back to back for >= --sustain-seconds (default 2 s) AFTER the contract region (no stage
                events, nothing else between the steps): ms_per_step / value once the chip has settled into the clock it
                holds under this load (the 20-step contract region lasts ~0.12 s).  The contract fields above are
                untouched by it.
"""
import argparse
import ctypes
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "sink-flash-attention-kernel_amd"), ROOT]

import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2516.6   # 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz (MI355X_MICROARCH.md, matrix cores)
WORKLOAD = dict(name="C3", B=4, Hq=32, Hkv=8, N=8192, D=128, ns=4, W=4096)


class HipEvents:
    """Raw hipEvent_t handles (the library records them on its launch stream)."""

    def __init__(self, n):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        self.ev = (ctypes.c_void_p * n)()
        for i in range(n):
            e = ctypes.c_void_p()
            assert self.hip.hipEventCreate(ctypes.byref(e)) == 0
            self.ev[i] = e

    def elapsed(self, i, j):
        ms = ctypes.c_float()
        self.hip.hipEventSynchronize(self.ev[j])
        rc = self.hip.hipEventElapsedTime(ctypes.byref(ms), self.ev[i], self.ev[j])
        return ms.value if rc == 0 else float("nan")


def cpu_baseline(w, reps=3):
    """Dense eager fp32 attention fwd+bwd on host cores for 1 batch x 1 KV group of the workload."""
    from oracle import sink_oracle as O
    g = w["Hq"] // w["Hkv"]
    N, D = w["N"], w["D"]
    torch.set_num_threads(os.cpu_count())
    gen = torch.Generator().manual_seed(1)
    q = torch.randn(1, g, N, D, generator=gen).requires_grad_(True)
    k = torch.randn(1, 1, N, D, generator=gen).requires_grad_(True)
    v = torch.randn(1, 1, N, D, generator=gen).requires_grad_(True)
    do = torch.randn(1, g, N, D, generator=gen)
    times = []
    for rep in range(1 + reps):                 # the first repetition warms up (allocator, thread pool) and is dropped
        q.grad = k.grad = v.grad = None
        t0 = time.perf_counter()
        o, _ = O.sink_attention_dense(q, k, v, w["ns"], w["W"], dtype=torch.float32)
        o.backward(do)
        times.append(time.perf_counter() - t0)
        del o
    timed = sorted(times[1:])
    dt = timed[len(timed) // 2]
    flops = O.flops_fwd_bwd(1, g, N, D, w["ns"], w["W"])
    return {"value": round(flops / dt / 1e12, 5), "unit": "TFLOP/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"dense eager fp32 fwd+bwd of 1 batch x 1 KV group ({g} q heads) at N={N} D={D} ns={w['ns']} "
                      f"W={w['W']}: median {dt:.2f} s of {reps} repetitions ({', '.join('%.2f' % t for t in times[1:])}) "
                      f"after one warm-up ({times[0]:.2f} s); per-(batch, KV head) work is independent, so the full "
                      f"workload is {w['B'] * w['Hkv']}x this"}


def pmc_traffic(kernel_key):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary (a separate
    builder run, see tools/pmc_round.sh), and where it came from.  (None, None) when there is none."""
    best = (None, None)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            d = json.load(open(path))
            if kernel_key in d and d[kernel_key].get("hbm_bytes_per_launch") is not None:
                best = (d[kernel_key]["hbm_bytes_per_launch"],
                        f"profiles/{os.path.basename(path)} (builder rocprofv3 --pmc run of this kernel, not this run)")
        except Exception:
            pass
    return best


def reduce_max_seconds(dt, device, world):
    """Step time of the job = the slowest rank's (the contract: MAX over ranks)."""
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    return dt


def job_value(flops_per_rank, world, seconds_per_step):
    """Whole-job throughput: every rank processed its own batch (weak scaling), TFLOP/s."""
    return flops_per_rank * world / seconds_per_step / 1e12


def dry_run(args):
    """No GPU: each rank 'works' for (rank+1) x 10 ms per step, then the same barrier / max-over-ranks / JSON
    logic as the real run.  Used by tests/test_multirank_gloo.py with world_size 2 on CPU."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    from oracle.sink_oracle import flops_fwd_bwd
    w = WORKLOAD
    f_fb = flops_fwd_bwd(w["B"], w["Hq"], w["N"], w["D"], w["ns"], w["W"])
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (rank + 1))
    if world > 1:
        dist.barrier()
    dt = reduce_max_seconds(time.perf_counter() - t0, torch.device("cpu"), world)
    if rank == 0:
        print(json.dumps({"metric": "dry-run (no kernels)", "dry_run": True, "n_gpus": world, "steps": args.steps,
                          "ms_per_step": round(dt / args.steps * 1e3, 3), "scaling": "weak",
                          "value": round(job_value(f_fb, world, dt / args.steps), 3), "unit": "TFLOP/s",
                          "global_batch": w["B"] * world}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="after the contract region: back-to-back steps for at least this long (0: skip), reported as `sustained`")
    ap.add_argument("--dump-slice", default=None, metavar="PATH",
                    help="after the timed region run ONE more step and save the tensors of the last (batch, KV head) "
                         "unit (q, k, v, dO, O, dQ, dK, dV) to PATH: tests/test_gpu_bench_contract.py checks them "
                         "against the oracle, so the bench cannot go fast on a wrong answer")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the multi-rank control path (gloo, no kernels, value is meaningless)")
    args = ap.parse_args()
    if args.dry_run:
        return dry_run(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    from oracle.sink_oracle import flops_fwd, flops_fwd_bwd
    from sink_attention import _native, sink_flash_attention

    w = WORKLOAD
    B, Hq, Hkv, N, D, ns, W = (w[x] for x in ("B", "Hq", "Hkv", "N", "D", "ns", "W"))
    torch.manual_seed(42 + rank)
    q = torch.randn(B, Hq, N, D, device=dev, dtype=torch.bfloat16).requires_grad_(True)
    k = torch.randn(B, Hkv, N, D, device=dev, dtype=torch.bfloat16).requires_grad_(True)
    v = torch.randn(B, Hkv, N, D, device=dev, dtype=torch.bfloat16).requires_grad_(True)
    do = torch.randn(B, Hq, N, D, device=dev, dtype=torch.bfloat16)

    lib = _native.lib()
    stages = [HipEvents(4) for _ in range(args.steps)]     # one set per timed step: per-kernel AVERAGES over the region
    fwd_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    paths = {}

    def step(i=None):
        if i is not None:
            fwd_ev[i][0].record()
        out = sink_flash_attention(q, k, v, num_sink=ns, window_size=W)
        if i is not None:
            fwd_ev[i][1].record()
        paths["fwd"] = _native.last_path()
        out.backward(do)
        q.grad = k.grad = v.grad = None

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    paths["bwd"] = _native.last_path()
    t0 = time.perf_counter()
    for i in range(args.steps):
        # the library records 4 events of set i on its launch stream during this step's backward (a few us of host work)
        lib.sfa_debug_set_stage_events(stages[i].ev, 4)
        step(i)
    lib.sfa_debug_set_stage_events(None, 0)
    barrier()
    dt = reduce_max_seconds(time.perf_counter() - t0, dev, world)

    ms_per_step = dt / args.steps * 1e3
    f_fb = flops_fwd_bwd(B, Hq, N, D, ns, W)
    f_f = flops_fwd(B, Hq, N, D, ns, W)
    value = job_value(f_fb, world, dt / args.steps)

    if rank == 0 and args.dump_slice:
        out = sink_flash_attention(q, k, v, num_sink=ns, window_size=W)
        out.backward(do)
        torch.cuda.synchronize()
        g_ = Hq // Hkv
        b_, hk_ = B - 1, Hkv - 1
        hs = slice(hk_ * g_, (hk_ + 1) * g_)
        cut = lambda t, h: t.detach()[b_:b_ + 1, h].cpu().clone()
        torch.save({"q": cut(q, hs), "k": cut(k, slice(hk_, hk_ + 1)), "v": cut(v, slice(hk_, hk_ + 1)),
                    "do": cut(do, hs), "o": cut(out, hs), "dq": cut(q.grad, hs), "dk": cut(k.grad, slice(hk_, hk_ + 1)),
                    "dv": cut(v.grad, slice(hk_, hk_ + 1)), "ns": ns, "W": W}, args.dump_slice)
        q.grad = k.grad = v.grad = None

    # ---- sustained: the same step back to back for >= --sustain-seconds (DVFS steady state), same barrier / max-over-ranks
    sustained = None
    if args.sustain_seconds > 0:
        n_sus, dts = 0, 0.0
        per = max(dt / args.steps, 1e-6)
        while dts < args.sustain_seconds:                    # (regions are added up until the requested time is reached)
            n = max(args.steps, int((args.sustain_seconds - dts) / per * 1.05) + 1)
            barrier()
            t1 = time.perf_counter()
            for _ in range(n):
                step()
            barrier()
            d1 = reduce_max_seconds(time.perf_counter() - t1, dev, world)
            n_sus, dts = n_sus + n, dts + d1
            per = max(dts / n_sus, 1e-6)
        sustained = {"steps": n_sus, "seconds": round(dts, 3), "ms_per_step": round(dts / n_sus * 1e3, 4),
                     "value": round(job_value(f_fb, world, dts / n_sus), 2),
                     "pct_mfma_peak": round(job_value(f_fb, world, dts / n_sus) / world / PEAK_BF16_TFLOPS * 100, 2)}

    if rank == 0:
        fwd_ms = sorted(s.elapsed_time(e) for s, e in fwd_ev)
        fwd_avg = sum(fwd_ms) / len(fwd_ms)
        avg = lambda a, b: sum(st.elapsed(a, b) for st in stages) / len(stages)
        pre, dkdv, dq = avg(0, 1), avg(1, 2), avg(2, 3)
        # algorithmic FLOPs per kernel: fwd 4, dK/dV kernel 4 (dV, dK) + its share ... -> use the 5-product split:
        # S and dP recomputations are not algorithmic work; dK/dV kernel owns dV+dK (4*D*pairs) plus S,dP (4) = 8,
        # dQ kernel owns dQ (2*D*pairs).  The two backward kernels together carry the 10*D*pairs of the backward.
        kernels = {
            "fwd": (fwd_avg, f_f, paths.get("fwd", "")),
            "bwd_dkdv": (dkdv, f_f * 2.0, paths.get("bwd", "")),
            "bwd_dq": (dq, f_f * 0.5, paths.get("bwd", "")),
        }
        dom = max(kernels, key=lambda n: kernels[n][0] if kernels[n][0] == kernels[n][0] else -1)
        dur, fl, path = kernels[dom]
        achieved = fl / (dur * 1e-3) / 1e12 if dur and dur == dur and dur > 0 else None
        per_kernel = {}
        for name, (ms_, fl_, _p) in kernels.items():
            tf = fl_ / (ms_ * 1e-3) / 1e12 if ms_ == ms_ and ms_ > 0 else None
            per_kernel[name] = {"ms": round(ms_, 4), "algorithmic_tflops": round(tf, 1) if tf else None,
                                "frac": round(tf / PEAK_BF16_TFLOPS, 4) if tf else None}
        traffic, traffic_src = pmc_traffic(dom)
        roofline = {"bound": "mfma", "kernel": f"{dom} ({path})",
                    "achieved": round(achieved, 2) if achieved else None, "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4) if achieved else None,
                    "avg_launch_ms": round(dur, 4), "traffic": traffic,
                    # NOT measured in this run: PMC counters need their own rocprofv3 pass (tools/pmc_round.sh)
                    "traffic_source": traffic_src,
                    "per_kernel": per_kernel,
                    "fwd_frac": per_kernel["fwd"]["frac"],
                    "stage_ms": {"fwd": round(fwd_avg, 4), "bwd_preprocess": round(pre, 4),
                                 "bwd_dkdv": round(dkdv, 4), "bwd_dq": round(dq, 4)}}
        line = {
            "metric": "attn fwd+bwd TFLOP/s (% MFMA peak) at B=4 H=32 N=8192 D=128 bf16",
            "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "pct_mfma_peak": round(value / world / PEAK_BF16_TFLOPS * 100, 2),
            "config": {"workload": "C3: sink_flash_attention fwd+bwd, GQA bf16, B=4 per GPU, H_q=32, H_kv=8, "
                                   "N=8192, D=128, num_sink=4, window=4096", "batch_per_gpu": B,
                       "global_batch": B * world, "seq_len": N, "parallelism": f"independent batches x{world}",
                       "algorithmic_gflop_per_step_per_gpu": round(f_fb / 1e9, 1)},
            "roofline": roofline,
        }
        if sustained is not None:
            line["sustained"] = sustained
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
