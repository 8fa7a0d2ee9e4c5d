"""sink_flash_attention: flash attention with sink tokens + sliding window + s_aux.

MI355X-native host side of the op: same signature, defaults, shape contract and
autograd behaviour as the reference's ``sink_attention/sink_flash_attention.py``
(``SinkFlashAttentionFunc`` :491-667, ``sink_flash_attention`` :670-689), but every
device kernel is hand-written HIP behind the C ABI of ``include/sfa.h`` (no Triton).

Attention pattern for query i:  keys j <= i with (j < num_sink or j >= i - W + 1).
``s_aux`` [H_q] is an extra per-head logit that enters the softmax denominator only
(the gpt-oss "sinks" parameter).

Differences from the reference (all inside its tolerances / a superset of its support):
  * any head dim (the reference needs a power of two), fp32 inputs use exact-f32 kernels;
  * q/k/v may be strided views (e.g. a transposed [B,N,H,D] activation) - no ``.contiguous()``
    copy unless the head dim itself is strided;
  * dK/dV are accumulated over the GQA group inside the kernel (the reference materialises
    per-Q-head dK/dV and sums in PyTorch, :585-586,:648-651);
  * Delta and ds_aux come from one HIP preprocess pass (the reference: eager ops :582,:658-665).
"""
import math
import os

import torch

from . import _native as N


class SinkFlashAttentionFunc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, num_sink, window_size, s_aux=None, out_bnhd=False, flags=0):
        N.require_gpu(q, k, v, s_aux)
        B, H_q, Nq, D = q.shape
        H_kv = k.shape[1]
        # shape contract of the reference (sink_flash_attention.py:494-498), relaxed in one way: k / v may hold MORE
        # rows than q; the queries are then the last N_q positions of the key sequence (MFMA kernels only)
        Nk = k.shape[2]
        assert k.shape == (B, H_kv, Nk, D) and Nk >= Nq, f"k {tuple(k.shape)} does not fit q {tuple(q.shape)}"
        assert v.shape == (B, H_kv, Nk, D)
        assert H_q % H_kv == 0
        if q.dtype not in N.SFA_DTYPE or k.dtype != q.dtype or v.dtype != q.dtype:
            raise TypeError(f"q/k/v must share one dtype in (float32, float16, bfloat16); got "
                            f"{q.dtype}, {k.dtype}, {v.dtype}")
        use_s_aux = s_aux is not None
        s_aux_f = None
        if use_s_aux:
            assert s_aux.shape == (H_q,), f"s_aux shape must be [H_q={H_q}], got {s_aux.shape}"
            s_aux_f = s_aux.detach().contiguous().float()
        num_sink, window_size = int(num_sink), int(window_size)
        scale = 1.0 / math.sqrt(D) if D > 0 else 1.0

        q, k, v = N.unit_inner(q), N.unit_inner(k), N.unit_inner(v)
        if out_bnhd:   # memory layout [B, N, H, D], returned as a [B, H, N, D] view
            o = torch.empty((B, Nq, H_q, D), device=q.device, dtype=q.dtype).transpose(1, 2)
        else:
            o = torch.empty((B, H_q, Nq, D), device=q.device, dtype=q.dtype)
        lse = torch.empty((B, H_q, Nq), device=q.device, dtype=torch.float32)
        lib = N.lib()
        with torch.cuda.device(q.device):
            st = lib.sfa_fwd(N.desc(q), N.desc(k), N.desc(v), N.desc(o), lse.data_ptr(),
                             s_aux_f.data_ptr() if use_s_aux else None, num_sink, window_size, scale, flags,
                             N.stream_ptr(q.device))
        N.check(st, "sfa_fwd")

        ctx.save_for_backward(q, k, v, o, lse, s_aux_f if use_s_aux else torch.empty(0, device=q.device))
        ctx.num_sink, ctx.window_size, ctx.scale = num_sink, window_size, scale
        ctx.use_s_aux, ctx.flags = use_s_aux, flags
        ctx.s_aux_dtype = s_aux.dtype if use_s_aux else None
        return o

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, do):
        q, k, v, o, lse, s_aux_f = ctx.saved_tensors
        B, H_q, Nq, D = q.shape
        H_kv = k.shape[1]
        do = N.unit_inner(do)          # also materialises expanded (stride-0) grads from .sum().backward()
        if do.dtype != q.dtype:
            do = do.to(q.dtype)
        dq = torch.empty((B, H_q, Nq, D), device=q.device, dtype=q.dtype)
        # no query rows (possible with N_q < N_kv): nothing is launched, the key gradients are exactly zero
        mk = torch.zeros if (Nq == 0 or B == 0 or H_q == 0) else torch.empty
        dk = mk((B, H_kv, k.shape[2], D), device=q.device, dtype=q.dtype)
        dv = mk((B, H_kv, k.shape[2], D), device=q.device, dtype=q.dtype)
        ds_aux = mk((H_q,), device=q.device, dtype=torch.float32) if ctx.use_s_aux else None
        lib = N.lib()
        flags = N.bwd_flags(ctx.flags)
        ws_bytes = lib.sfa_bwd_workspace_bytes(B, H_q, H_kv, Nq, D, N.SFA_DTYPE[q.dtype], ctx.num_sink,
                                               ctx.window_size, flags)
        ws = torch.empty((max(int(ws_bytes), 256),), device=q.device, dtype=torch.uint8)
        with torch.cuda.device(q.device):
            st = lib.sfa_bwd(N.desc(q), N.desc(k), N.desc(v), N.desc(o), N.desc(do), lse.data_ptr(),
                             s_aux_f.data_ptr() if ctx.use_s_aux else None, N.desc(dq), N.desc(dk), N.desc(dv),
                             ds_aux.data_ptr() if ctx.use_s_aux else None, ws.data_ptr(), ws.numel(),
                             ctx.num_sink, ctx.window_size, ctx.scale, flags, N.stream_ptr(q.device))
        N.check(st, "sfa_bwd")
        if ctx.use_s_aux and ds_aux.dtype != ctx.s_aux_dtype:
            ds_aux = ds_aux.to(ctx.s_aux_dtype)
        return dq, dk, dv, None, None, ds_aux, None, None


def sink_flash_attention(q, k, v, num_sink=4, window_size=512, s_aux=None):
    """
    Flash Attention with Attention Sink support (MI355X / HIP).

    Args:
        q: [B, H_q, N, D]    k, v: [B, H_kv, N, D]   (float32 / float16 / bfloat16)
        num_sink: leading tokens every query may attend to (default 4)
        window_size: causal sliding-window length (default 512)
        s_aux: optional learnable per-Q-head extra logit [H_q]; adds exp(s_aux) to the
               softmax denominator without a value row (gpt-oss attention sink).
    Returns:
        [B, H_q, N, D] in q's dtype; differentiable w.r.t. q, k, v and s_aux.
    """
    return SinkFlashAttentionFunc.apply(q, k, v, num_sink, window_size, s_aux)


def _sink_flash_attention_ex(q, k, v, num_sink, window_size, s_aux=None, out_bnhd=False, force_generic=False):
    """Internal entry used by the HF/verl boundary and the tests: ``out_bnhd`` makes the output
    live in [B,N,H,D] memory (returned as a [B,H,N,D] view) so the boundary needs no transpose copy;
    ``force_generic`` routes to the exact-f32 kernels."""
    return SinkFlashAttentionFunc.apply(q, k, v, num_sink, window_size, s_aux, out_bnhd,
                                        N.FLAG_FORCE_GENERIC if force_generic else 0)
