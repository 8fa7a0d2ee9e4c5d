#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE in the build container.

    cd /tmp && TRITON_INTERPRET=1 PYTHONDONTWRITEBYTECODE=1 \
        python3 /root/repo/tests/golden/make_golden.py

Imports the reference from /root/reference (read-only, never copied): its
Triton kernels run on CPU tensors through Triton's interpreter, its eager
oracles are plain torch.  Every fixture stores the INPUT tensors explicitly
(CPU generator, seed in the file) and the reference's outputs, so the GPU box
(which has no /root/reference) only ever needs the .npz files.

Keys per case:  q,k,v[,s_aux][,do]  inputs (fp32 arrays; "*_rounded" cases hold
values already rounded to bf16/fp16 so they are exact in that dtype),
  o_eager          the reference's eager oracle output
  o_kernel         the reference's Triton kernel output (interpreter) when it can run
  dq/dk/dv/ds_aux_eager   autograd of the eager oracle
  dq/dk/dv/ds_aux_kernel  the reference kernel's backward
"""
import os
import sys

os.environ.setdefault("TRITON_INTERPRET", "1")
sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)

import numpy as np
import torch

from sink_attention import sink_flash_attention, sink_decode_attention          # reference kernels
from sink_attention.verl_patch import _sink_flash_attention_forward, _is_packed  # reference boundary
from tests.test_sink_attention import naive_sink_attention                      # reference eager oracles
from tests.test_s_aux import reference_attention_with_s_aux
from tests.test_decode_kernel import reference_decode_attention

OUT = os.path.dirname(os.path.abspath(__file__))


def rnd(shape, gen, scale=1.0):
    return torch.randn(*shape, generator=gen, dtype=torch.float32) * scale


def npz(name, **arrs):
    """Arrays whose values are exact in fp16 are stored as float16; values exact in
    bf16 are stored as the high 16 bits of the fp32 word under key ``<name>__bf16``
    (tests/golden_util.py undoes both).  Everything else stays float32."""
    out = {}
    for k_, v_ in arrs.items():
        if v_ is None:
            continue
        if isinstance(v_, torch.Tensor):
            v_ = v_.detach().float().numpy()
        v_ = np.asarray(v_)
        if v_.dtype == np.float32 and v_.size > 64:
            with np.errstate(over="ignore"):
                h = v_.astype(np.float16)
            bits = v_.view(np.uint32)
            if np.array_equal(h.astype(np.float32), v_, equal_nan=True):
                v_ = h
            elif not (bits & 0xFFFF).any():
                k_, v_ = k_ + "__bf16", (bits >> 16).astype(np.uint16)
        out[k_] = v_
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, {k_: v_.shape for k_, v_ in out.items() if hasattr(v_, "shape")})


def eager(q, k, v, ns, W, s_aux):
    """The reference oracle appropriate for the case: with s_aux the gpt-oss
    style oracle (tests/test_s_aux.py), else naive_sink_attention."""
    if s_aux is None:
        return naive_sink_attention(q, k, v, ns, W)
    return reference_attention_with_s_aux(q, k, v, s_aux=s_aux, window_size=W, num_sink=ns)


def fwd_bwd_case(name, B, Hq, Hkv, N, D, ns, W, seed, use_s_aux, kernel=True, kernel_dtype=torch.float32,
                 eager_ok=True):
    g = torch.Generator().manual_seed(seed)
    q, k, v = rnd((B, Hq, N, D), g), rnd((B, Hkv, N, D), g), rnd((B, Hkv, N, D), g)
    s_aux = rnd((Hq,), g, 0.5) if use_s_aux else None
    do = rnd((B, Hq, N, D), g)
    if kernel_dtype != torch.float32:      # make inputs exact in the low-precision dtype
        q, k, v, do = (t.to(kernel_dtype).float() for t in (q, k, v, do))
    res = dict(q=q, k=k, v=v, s_aux=s_aux, do=do,
               meta=np.array([B, Hq, Hkv, N, D, ns, W, seed], dtype=np.int64))
    if eager_ok:
        leaves = [t.clone().requires_grad_(True) for t in (q, k, v)]
        sa = s_aux.clone().requires_grad_(True) if use_s_aux else None
        o = eager(leaves[0], leaves[1], leaves[2], ns, W, sa)
        grads = torch.autograd.grad(o, leaves + ([sa] if use_s_aux else []), do)
        res.update(o_eager=o, dq_eager=grads[0], dk_eager=grads[1], dv_eager=grads[2])
        if use_s_aux:
            res["ds_aux_eager"] = grads[3]
    if kernel:
        leaves = [t.to(kernel_dtype).clone().requires_grad_(True) for t in (q, k, v)]
        sa = s_aux.clone().requires_grad_(True) if use_s_aux else None
        o = sink_flash_attention(leaves[0], leaves[1], leaves[2], num_sink=ns, window_size=W, s_aux=sa)
        o.backward(do.to(kernel_dtype))
        res.update(o_kernel=o, dq_kernel=leaves[0].grad, dk_kernel=leaves[1].grad, dv_kernel=leaves[2].grad)
        if use_s_aux:
            res["ds_aux_kernel"] = sa.grad
    npz(name, **res)


def main():
    # F1: BASELINE config 1 exactly (fp32 eager on CPU): B=1 H=2 N=128 D=64 ns=4 W=32
    fwd_bwd_case("f1_c1_fp32", 1, 2, 2, 128, 64, 4, 32, 42, False)
    # F2: s_aux + GQA 4->1, ns=0, full-causal and sliding window
    fwd_bwd_case("f2_saux_gqa_full", 1, 4, 1, 128, 64, 0, 128, 42, True)
    fwd_bwd_case("f2_saux_gqa_win32", 1, 4, 1, 128, 64, 0, 32, 43, True)
    # F3: positional sinks AND s_aux, ragged N, D=128, tiny window
    fwd_bwd_case("f3_mixed_ragged", 1, 2, 1, 77, 128, 4, 7, 44, True)
    # fp16 kernel path (interpreter supports fp16, not bf16)
    fwd_bwd_case("f3_fp16_gqa", 1, 4, 2, 200, 64, 4, 64, 45, False, kernel_dtype=torch.float16)
    fwd_bwd_case("f3_fp16_d128", 2, 2, 1, 96, 128, 3, 50, 46, True, kernel_dtype=torch.float16)
    # F4: D=80 (gpt-oss-20b head dim), bf16-rounded inputs, eager oracle only
    # (the reference kernel cannot run D=80: tl.arange needs a power of two)
    g = torch.Generator().manual_seed(47)
    B, Hq, Hkv, N, D = 1, 4, 1, 64, 80
    q, k, v = (rnd(s, g).bfloat16().float() for s in ((B, Hq, N, D), (B, Hkv, N, D), (B, Hkv, N, D)))
    s_aux = rnd((Hq,), g, 0.5)
    do = rnd((B, Hq, N, D), g).bfloat16().float()
    leaves = [t.clone().requires_grad_(True) for t in (q, k, v)]
    sa = s_aux.clone().requires_grad_(True)
    o = reference_attention_with_s_aux(leaves[0], leaves[1], leaves[2], s_aux=sa, window_size=N)
    gr = torch.autograd.grad(o, leaves + [sa], do)
    npz("f4_d80_bf16_rounded", q=q, k=k, v=v, s_aux=s_aux, do=do, o_eager=o, dq_eager=gr[0], dk_eager=gr[1],
        dv_eager=gr[2], ds_aux_eager=gr[3], meta=np.array([B, Hq, Hkv, N, D, 0, N, 47], dtype=np.int64))
    # D=80 with a short window as in BASELINE config 4 (W=128 > N here -> use W=16)
    leaves = [t.clone().requires_grad_(True) for t in (q, k, v)]
    sa = s_aux.clone().requires_grad_(True)
    o = reference_attention_with_s_aux(leaves[0], leaves[1], leaves[2], s_aux=sa, window_size=16)
    gr = torch.autograd.grad(o, leaves + [sa], do)
    npz("f4_d80_win16", q=q, k=k, v=v, s_aux=s_aux, do=do, o_eager=o, dq_eager=gr[0], dk_eager=gr[1],
        dv_eager=gr[2], ds_aux_eager=gr[3], meta=np.array([B, Hq, Hkv, N, D, 0, 16, 47], dtype=np.int64))

    # F5: edge cases (kernel + eager, fp32).  The all-masked rows of W=0/ns=0 without
    # s_aux differ between the two reference oracles (uniform vs zero); only the kernel
    # output (zeros) is stored for that case.
    fwd_bwd_case("f5_w_ge_n", 1, 2, 2, 48, 32, 4, 64, 50, False)
    fwd_bwd_case("f5_ns_ge_n", 1, 2, 1, 40, 32, 64, 5, 51, False)
    fwd_bwd_case("f5_w0_ns4", 1, 2, 2, 40, 32, 4, 0, 52, False)
    fwd_bwd_case("f5_w1_ns4", 1, 2, 2, 64, 64, 4, 1, 53, False)
    fwd_bwd_case("f5_w0_ns0_saux", 1, 2, 2, 40, 32, 0, 0, 54, True, eager_ok=False)
    fwd_bwd_case("f5_w0_ns0", 1, 2, 2, 40, 32, 0, 0, 55, False, eager_ok=False)
    fwd_bwd_case("f5_d16_fd", 1, 2, 2, 32, 16, 0, 32, 56, True)      # the shape of test_ds_aux_gradient_numerical
    fwd_bwd_case("f5_big_ns", 1, 2, 2, 160, 32, 40, 24, 57, True)    # sink range spans >1 kernel block

    # F6: decode
    for name, Hq, Hkv, Nkv, D, dt, with_aux, seed in [
        ("f6_dec_n5_fp32", 4, 4, 5, 64, torch.float32, False, 60),
        ("f6_dec_n12_gqa_fp32", 8, 2, 12, 32, torch.float32, False, 61),
        ("f6_dec_n64_bf16", 8, 2, 64, 128, torch.bfloat16, True, 62),
        ("f6_dec_n300_bf16", 8, 2, 300, 128, torch.bfloat16, True, 63),
        ("f6_dec_n300_fp16_noaux", 4, 4, 300, 128, torch.float16, False, 64),
        ("f6_dec_n700_d64_fp32", 4, 1, 700, 64, torch.float32, True, 65),
        ("f6_dec_n520_d256_bf16", 2, 1, 520, 256, torch.bfloat16, True, 66),
    ]:
        g = torch.Generator().manual_seed(seed)
        B = 2 if "n300_bf16" in name else 1
        q, k, v = rnd((B, Hq, 1, D), g), rnd((B, Hkv, Nkv, D), g), rnd((B, Hkv, Nkv, D), g)
        q, k, v = (t.to(dt) for t in (q, k, v))
        s_aux = (rnd((Hq,), g, 2.0).to(torch.bfloat16 if dt != torch.float32 else dt)) if with_aux else None
        o_k = sink_decode_attention(q, k, v, s_aux=s_aux)
        o_e = reference_decode_attention(q, k, v, s_aux=s_aux)
        npz(name, q=q, k=k, v=v, s_aux=s_aux, o_kernel=o_k, o_eager=o_e,
            meta=np.array([B, Hq, Hkv, Nkv, D, seed, {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[dt]],
                          dtype=np.int64))
    # s_aux = 100 swallows all mass (tests/test_decode_kernel.py:168-183)
    g = torch.Generator().manual_seed(67)
    q, k, v = (rnd(s, g).bfloat16() for s in ((1, 4, 1, 128), (1, 2, 256, 128), (1, 2, 256, 128)))
    s_aux = torch.full((4,), 100.0).bfloat16()
    npz("f6_dec_saux100", q=q, k=k, v=v, s_aux=s_aux, o_kernel=sink_decode_attention(q, k, v, s_aux=s_aux),
        o_eager=reference_decode_attention(q, k, v, s_aux=s_aux),
        meta=np.array([1, 4, 2, 256, 128, 67, 2], dtype=np.int64))

    # F7: the HF/verl boundary: [B,N,H,D] in and out through the reference's replacement function
    g = torch.Generator().manual_seed(70)
    B, N, Hq, Hkv, D = 2, 48, 4, 2, 64
    qs, ks, vs = rnd((B, N, Hq, D), g), rnd((B, N, Hkv, D), g), rnd((B, N, Hkv, D), g)
    s_aux = rnd((Hq,), g, 0.5)
    o_none = _sink_flash_attention_forward(qs, ks, vs, None, N, is_causal=True, sliding_window=None, s_aux=s_aux)
    o_w16 = _sink_flash_attention_forward(qs, ks, vs, None, N, is_causal=True, sliding_window=16, s_aux=s_aux)
    o_noaux = _sink_flash_attention_forward(qs, ks, vs, None, N, is_causal=True, sliding_window=16)
    # Ulysses-style: s_aux has 2x the local heads; rank 0 slice is used when torch.distributed is not initialised
    s_aux_big = torch.cat([s_aux, rnd((Hq,), g, 0.5)])
    o_sp = _sink_flash_attention_forward(qs, ks, vs, None, N, is_causal=True, sliding_window=16, s_aux=s_aux_big)
    qd = rnd((1, 1, 4, 64), g)
    kd, vd = rnd((1, 64, 2, 64), g), rnd((1, 64, 2, 64), g)
    o_dec = _sink_flash_attention_forward(qd, kd, vd, None, 1, is_causal=True, sliding_window=16, s_aux=s_aux)
    pid_plain = torch.arange(10).view(1, 10)
    pid_packed = torch.tensor([[0, 1, 2, 3, 0, 1, 2, 0, 1, 2]])
    pid_1d = torch.arange(10)
    pid_len1 = torch.zeros(2, 1, dtype=torch.long)
    npz("f7_boundary", qs=qs, ks=ks, vs=vs, s_aux=s_aux, s_aux_big=s_aux_big, o_none=o_none, o_w16=o_w16,
        o_noaux=o_noaux, o_sp=o_sp, qd=qd, kd=kd, vd=vd, o_dec=o_dec,
        packed_truth=np.array([_is_packed(pid_plain), _is_packed(pid_packed), _is_packed(pid_1d),
                               _is_packed(pid_len1), _is_packed(None)], dtype=np.int64))


def cache_cases():
    """F8: the reference's SinkCacheLayer driven through prefill + decode steps:
    bookkeeping after every step and the linearised K/V it hands to the decode kernel."""
    from sink_attention.cache import SinkCacheLayer
    # the reference class predates transformers 5.x's abstract CacheLayerMixin.get_max_length and cannot be
    # instantiated against the installed transformers; clearing the abstract set does not touch its logic
    SinkCacheLayer.__abstractmethods__ = frozenset()
    for name, ns, W, prefill, steps, seed in [("f8_cache_evict", 2, 3, 4, 10, 80), ("f8_cache_short_prefill", 4, 8, 2, 6, 81),
                                              ("f8_cache_overflow", 4, 8, 16, 5, 82), ("f8_cache_exact", 3, 5, 8, 7, 83)]:
        g = torch.Generator().manual_seed(seed)
        Ntot = prefill + steps
        k_all, v_all = rnd((1, 2, Ntot, 8), g), rnd((1, 2, Ntot, 8), g)
        layer = SinkCacheLayer(ns, W)
        res = dict(k_all=k_all, v_all=v_all, meta=np.array([ns, W, prefill, steps, seed], dtype=np.int64))
        state = []
        ko, vo = layer.update(k_all[:, :, :prefill], v_all[:, :, :prefill])
        state.append([layer.sink_len, layer.window_len, layer.write_pos, layer.seen_tokens, ko.shape[2]])
        lk, lv = layer.get_kv()
        res["k_lin_0"], res["v_lin_0"] = lk.clone(), lv.clone()
        for i in range(steps):
            pos = prefill + i
            ko, vo = layer.update(k_all[:, :, pos:pos + 1], v_all[:, :, pos:pos + 1])
            state.append([layer.sink_len, layer.window_len, layer.write_pos, layer.seen_tokens, ko.shape[2]])
            res[f"k_lin_{i + 1}"], res[f"v_lin_{i + 1}"] = ko.clone(), vo.clone()
        res["state"] = np.array(state, dtype=np.int64)
        npz(name, **res)


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "cache":
        cache_cases()
    else:
        main()
        cache_cases()
