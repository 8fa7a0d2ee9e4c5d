"""Packed (variable-length) sequences for sink flash attention.

The reference cannot handle packed batches: ``verl_patch.py:73-93`` hands them back to stock flash attention,
which silently drops ``s_aux`` (the very thing the reference exists to fix, its README "Why").  Here the HIP kernels
take ``cu_seqlens`` (``sfa_fwd_varlen`` / ``sfa_bwd_varlen`` in ``include/sfa.h``): the grid covers (sequence, KV head,
tile), every workgroup reads its sequence's first row and length from ``cu_seqlens`` and builds its buffer
descriptors over exactly those rows, so the mask and the ``s_aux`` logit restart at every sequence boundary and the
whole pack is ONE launch per kernel.  Shapes the packed kernels do not cover (fp32, head dims outside 64/80/96/128,
unaligned views) run sequence by sequence through the ordinary op on strided views (no input copies).
"""
import math
import os
from typing import List, Sequence, Union

import torch

from . import _native as N
from .sink_flash_attention import _sink_flash_attention_ex


_CHECK_CU = os.environ.get("SINK_ATTENTION_CHECK_CU", "0") == "1"


def seq_bounds_from_position_ids(position_ids: torch.Tensor) -> List[int]:
    """cu_seqlens (host list) of a single packed row: a new sequence starts wherever position_ids is 0."""
    pid = position_ids.reshape(-1)
    starts = (pid == 0).nonzero(as_tuple=False).flatten().tolist()
    if not starts or starts[0] != 0:
        starts = [0] + starts
    return starts + [pid.numel()]


class SinkFlashAttentionVarlenFunc(torch.autograd.Function):
    """One launch per kernel for the whole pack; same saved state as SinkFlashAttentionFunc plus cu_seqlens."""

    @staticmethod
    def forward(ctx, q, k, v, cu_dev, max_seqlen, num_sink, window_size, s_aux, checked=True):
        _, H_q, T, D = q.shape
        # A device cu_seqlens handed in with max_seqlen is NOT validated (that would be a host synchronisation): rows it
        # does not cover (padding behind cu[-1], an understated max_seqlen) are then never written by the kernels, so
        # outputs and gradients start from zeros instead of uninitialised memory on that path.
        alloc = torch.empty if checked else torch.zeros
        use_s_aux = s_aux is not None
        s_aux_f = s_aux.detach().contiguous().float() if use_s_aux else None
        scale = 1.0 / math.sqrt(D)
        q, k, v = N.unit_inner(q), N.unit_inner(k), N.unit_inner(v)
        o = alloc((1, H_q, T, D), device=q.device, dtype=q.dtype)
        lse = alloc((H_q, T), device=q.device, dtype=torch.float32)
        lib = N.lib()
        with torch.cuda.device(q.device):
            st = lib.sfa_fwd_varlen(N.desc(q), N.desc(k), N.desc(v), N.desc(o), lse.data_ptr(),
                                    s_aux_f.data_ptr() if use_s_aux else None, cu_dev.data_ptr(), cu_dev.numel() - 1,
                                    max_seqlen, num_sink, window_size, scale, 0, N.stream_ptr(q.device))
        N.check(st, "sfa_fwd_varlen")
        ctx.save_for_backward(q, k, v, o, lse, cu_dev, s_aux_f if use_s_aux else torch.empty(0, device=q.device))
        ctx.cfg = (max_seqlen, num_sink, window_size, scale, use_s_aux, s_aux.dtype if use_s_aux else None, checked)
        return o

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, do):
        q, k, v, o, lse, cu_dev, s_aux_f = ctx.saved_tensors
        max_seqlen, num_sink, window_size, scale, use_s_aux, s_aux_dtype, checked = ctx.cfg
        alloc = torch.empty if checked else torch.zeros
        _, H_q, T, D = q.shape
        H_kv = k.shape[1]
        do = N.unit_inner(do)
        if do.dtype != q.dtype:
            do = do.to(q.dtype)
        dq = alloc((1, H_q, T, D), device=q.device, dtype=q.dtype)
        dk = alloc((1, H_kv, T, D), device=q.device, dtype=q.dtype)
        dv = alloc((1, H_kv, T, D), device=q.device, dtype=q.dtype)
        ds_aux = torch.empty((H_q,), device=q.device, dtype=torch.float32) if use_s_aux else None
        lib = N.lib()
        flags = N.bwd_flags()
        ws_bytes = lib.sfa_bwd_workspace_bytes(1, H_q, H_kv, T, D, N.SFA_DTYPE[q.dtype], num_sink, window_size, flags)
        ws = torch.empty((max(int(ws_bytes), 256),), device=q.device, dtype=torch.uint8)
        with torch.cuda.device(q.device):
            st = lib.sfa_bwd_varlen(N.desc(q), N.desc(k), N.desc(v), N.desc(o), N.desc(do), lse.data_ptr(),
                                    s_aux_f.data_ptr() if use_s_aux else None, N.desc(dq), N.desc(dk), N.desc(dv),
                                    ds_aux.data_ptr() if use_s_aux else None, cu_dev.data_ptr(), cu_dev.numel() - 1,
                                    max_seqlen, ws.data_ptr(), ws.numel(), num_sink, window_size, scale, flags,
                                    N.stream_ptr(q.device))
        N.check(st, "sfa_bwd_varlen")
        if use_s_aux and ds_aux.dtype != s_aux_dtype:
            ds_aux = ds_aux.to(s_aux_dtype)
        return dq, dk, dv, None, None, None, None, ds_aux, None


def _aligned(t: torch.Tensor) -> bool:
    es = t.element_size()
    return (t.data_ptr() % 16 == 0 and all((t.stride(i) * es) % 16 == 0 for i in range(3))
            and (t.shape[2] + 64) * t.stride(2) * es + 512 < (1 << 32) - 65536)


def sink_flash_attention_varlen(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                                cu_seqlens: Union[torch.Tensor, Sequence[int]], num_sink: int = 4,
                                window_size: int = 512, s_aux: torch.Tensor = None,
                                max_seqlen: int = None) -> torch.Tensor:
    """
    q [1, H_q, T, D], k/v [1, H_kv, T, D]: ``len(cu_seqlens) - 1`` sequences packed along T
    (sequence i = rows cu_seqlens[i] : cu_seqlens[i+1]).  Every sequence is attended independently with
    valid(i, j) = (j <= i) and (j < num_sink or j >= i - window_size + 1) in ITS OWN positions, and its own s_aux
    term.  Returns [1, H_q, T, D]; differentiable w.r.t. q, k, v and s_aux.

    ``cu_seqlens`` may be a host sequence or a tensor (a device int32 tensor together with ``max_seqlen`` avoids
    any host synchronisation, which is what the HF / verl boundary provides).
    """
    N.require_gpu(q, k, v, s_aux)
    assert q.shape[0] == 1 and k.shape[0] == 1 and v.shape[0] == 1, "packed layout: batch dim must be 1"
    T = q.shape[2]
    native = (q.dtype in (torch.float16, torch.bfloat16) and k.dtype == q.dtype and v.dtype == q.dtype
              and N.lib().sfa_varlen_supported(N.SFA_DTYPE[q.dtype], q.shape[3]) == 1)
    if native:
        qq, kk, vv = N.unit_inner(q), N.unit_inner(k), N.unit_inner(v)
        native = _aligned(qq) and _aligned(kk) and _aligned(vv)
    if native:
        checked = True
        if isinstance(cu_seqlens, torch.Tensor) and cu_seqlens.is_cuda and max_seqlen is not None:
            # trusted as given (contract, include/sfa.h sfa_fwd_varlen): cu[0] == 0, non-decreasing, cu[-1] <= T,
            # max_seqlen >= the longest sequence.  SINK_ATTENTION_CHECK_CU=1 validates it (one host synchronisation).
            cu_dev = cu_seqlens.to(torch.int32).contiguous()
            checked = False
            if _CHECK_CU:
                cu = cu_dev.tolist()
                assert len(cu) >= 2 and cu[0] == 0 and cu[-1] <= T, f"bad cu_seqlens {cu} for T={T}"
                assert all(b >= a for a, b in zip(cu[:-1], cu[1:])), f"cu_seqlens must be non-decreasing: {cu}"
                assert max_seqlen >= max(b - a for a, b in zip(cu[:-1], cu[1:])), "max_seqlen understates the longest sequence"
                checked = cu[-1] == T
        else:
            cu = cu_seqlens.tolist() if isinstance(cu_seqlens, torch.Tensor) else list(cu_seqlens)
            assert len(cu) >= 2 and cu[0] == 0 and cu[-1] == T, f"bad cu_seqlens {cu} for T={T}"
            assert all(b >= a for a, b in zip(cu[:-1], cu[1:])), f"cu_seqlens must be non-decreasing: {cu}"
            max_seqlen = max(b - a for a, b in zip(cu[:-1], cu[1:]))
            cu_dev = torch.tensor(cu, dtype=torch.int32, device=q.device)
        if T == 0 or max_seqlen == 0:
            return torch.empty_like(q)
        return SinkFlashAttentionVarlenFunc.apply(q, k, v, cu_dev, int(min(max_seqlen, T)), int(num_sink),
                                                  int(window_size), s_aux, checked)

    # sequence by sequence on strided views (fp32, head dims without a packed kernel, unaligned views)
    cu = cu_seqlens.tolist() if isinstance(cu_seqlens, torch.Tensor) else list(cu_seqlens)
    assert len(cu) >= 2 and cu[0] == 0 and cu[-1] == T, f"bad cu_seqlens {cu} for T={T}"
    outs = []
    for a, b in zip(cu[:-1], cu[1:]):
        assert b >= a
        if b == a:
            continue
        outs.append(_sink_flash_attention_ex(q[:, :, a:b], k[:, :, a:b], v[:, :, a:b], num_sink, window_size,
                                             s_aux=s_aux))
    return torch.cat(outs, dim=2) if len(outs) != 1 else outs[0]
