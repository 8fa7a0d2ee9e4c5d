"""Sequence-parallel helpers for sink flash attention.

The reference's ``sink_attention/sp_utils.py`` shards the sequence over an SP group, broadcasts the sink K/V from rank 0
(``prepare_sink_kv_for_sp`` :26-78), all-reduces their gradients (``reduce_sink_kv_grads`` :81-127) and wraps both in
``SinkAttentionSPWrapper`` (:146-180).  The first two are kept with the same signature and results.  The wrapper of
the reference cannot work on ranks > 0: it calls ``sink_flash_attention`` with N_q != N_kv (asserted against at
``sink_flash_attention.py:494-498``) and a rank never sees the window keys that live on the previous rank.  Here the
wrapper is REPAIRED: every rank gets the sink keys plus the ``window_size - 1`` keys preceding its chunk (the halo)
with an autograd-aware all-gather and runs the kernels with its N_q local queries against the N_kv > N_q keys
``[sinks | halo | local]`` (query row i sits at key position i + N_kv - N_q);
gradients of the gathered K/V flow back to their owners through the collective's backward.  Collectives are plain
``torch.distributed`` calls (RCCL over xGMI on MI355X, gloo on CPU): this file has no device code.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def prepare_sink_kv_for_sp(k: torch.Tensor, v: torch.Tensor, num_sink: int, sp_group, rank: Optional[int] = None
                           ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Broadcast the sink K/V ``[B, H_kv, num_sink, D]`` from SP rank 0; ranks > 0 get them prepended to their chunk,
    rank 0 returns its tensors unchanged (reference :26-78)."""
    if num_sink == 0:
        return k, v
    if rank is None:
        rank = dist.get_rank(sp_group)
    B, H_kv, _n, D = k.shape
    if rank == 0:
        sink_k, sink_v = k[:, :, :num_sink].contiguous(), v[:, :, :num_sink].contiguous()
    else:
        sink_k = torch.empty(B, H_kv, num_sink, D, device=k.device, dtype=k.dtype)
        sink_v = torch.empty(B, H_kv, num_sink, D, device=v.device, dtype=v.dtype)
    src = dist.get_global_rank(sp_group, 0) if sp_group is not None else 0
    dist.broadcast(sink_k, src=src, group=sp_group)
    dist.broadcast(sink_v, src=src, group=sp_group)
    if rank == 0:
        return k, v
    return torch.cat([sink_k, k], dim=2), torch.cat([sink_v, v], dim=2)


def reduce_sink_kv_grads(dk: torch.Tensor, dv: torch.Tensor, num_sink: int, sp_group, rank: Optional[int] = None
                         ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Sum the sink-row gradients over the SP group: rank 0 gets the total written into (a copy of) its dK/dV, the
    other ranks get their tensors with the prepended sink rows stripped (reference :81-127)."""
    if num_sink == 0:
        return dk, dv
    if rank is None:
        rank = dist.get_rank(sp_group)
    sink_dk, sink_dv = dk[:, :, :num_sink].contiguous(), dv[:, :, :num_sink].contiguous()
    dist.all_reduce(sink_dk, op=dist.ReduceOp.SUM, group=sp_group)
    dist.all_reduce(sink_dv, op=dist.ReduceOp.SUM, group=sp_group)
    if rank == 0:
        dk, dv = dk.clone(), dv.clone()
        dk[:, :, :num_sink] = sink_dk
        dv[:, :, :num_sink] = sink_dv
        return dk, dv
    return dk[:, :, num_sink:], dv[:, :, num_sink:]


def get_local_position_offset(rank: int, n_local: int, num_sink: int) -> int:
    """Global position of the first token of SP rank ``rank``'s chunk (reference :130-143)."""
    return rank * n_local


class _AllGatherSeq(torch.autograd.Function):
    """[B, H, n, D] on every rank -> [B, H, P*n, D] everywhere; backward sums each rank's slice of the gradient."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        P = dist.get_world_size(group)
        parts = [torch.empty_like(x) for _ in range(P)]
        dist.all_gather(parts, x.contiguous(), group=group)
        return torch.cat(parts, dim=2)

    @staticmethod
    def backward(ctx, g):
        P, r = dist.get_world_size(ctx.group), dist.get_rank(ctx.group)
        g = g.contiguous()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)   # gloo has no reduce_scatter; volume is K/V-sized
        n = g.shape[2] // P
        return g[:, :, r * n:(r + 1) * n], None


def sp_extended_kv(k_full: torch.Tensor, v_full: torch.Tensor, rank: int, n_local: int, num_sink: int,
                   window_size: int) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """Keys SP rank ``rank`` needs, as ``[sinks | halo | local]`` slices of the full-sequence K/V, and the number of
    leading rows (sinks + halo) that have no local query.  Positions inside the result are consistent with the
    mask ``j < num_sink or j >= i - window_size + 1`` of the global problem."""
    lo = rank * n_local
    ns = min(num_sink, lo)                       # rank 0 (lo == 0) already owns the sinks
    hs = max(ns, lo - max(window_size - 1, 0))   # first halo key
    sl = lambda t: torch.cat([t[:, :, :ns], t[:, :, hs:lo + n_local]], dim=2) if (ns or hs < lo) else t[:, :, lo:lo + n_local]
    return sl(k_full), sl(v_full), ns + (lo - hs)


def sp_local_attention(q_local, k_ext, v_ext, lead: int, num_sink: int, window_size: int, s_aux=None):
    """Attention of the local queries against ``[sinks | halo | local]`` keys.  The MFMA kernels take N_q < N_kv
    directly (the queries are the last N_q key positions); shapes they do not cover (fp32, other head dims) get
    zero queries for the ``lead`` key rows, whose outputs are dropped."""
    from . import _native as N
    from .sink_flash_attention import _sink_flash_attention_ex
    if lead and q_local.dtype in (torch.float16, torch.bfloat16) and \
            N.lib().sfa_varlen_supported(N.SFA_DTYPE[q_local.dtype], q_local.shape[3]) == 1:
        return _sink_flash_attention_ex(q_local, k_ext, v_ext, num_sink, window_size, s_aux=s_aux)
    if lead:
        q_ext = torch.cat([q_local.new_zeros(q_local.shape[0], q_local.shape[1], lead, q_local.shape[3]), q_local], dim=2)
    else:
        q_ext = q_local
    out = _sink_flash_attention_ex(q_ext, k_ext, v_ext, num_sink, window_size, s_aux=s_aux)
    return out[:, :, lead:]


class SinkAttentionSPWrapper(torch.nn.Module):
    """``wrapper(q_local, k_local, v_local)`` -> this rank's rows of the full-sequence sink attention
    (equal-length contiguous chunks, rank r owns positions [r*n, (r+1)*n))."""

    def __init__(self, num_sink: int = 4, window_size: int = 4096, sp_group=None):
        super().__init__()
        self.num_sink, self.window_size, self.sp_group = num_sink, window_size, sp_group

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, s_aux=None) -> torch.Tensor:
        from .sink_flash_attention import _sink_flash_attention_ex
        if self.sp_group is None or dist.get_world_size(self.sp_group) == 1:
            return _sink_flash_attention_ex(q, k, v, self.num_sink, self.window_size, s_aux=s_aux)
        rank = dist.get_rank(self.sp_group)
        k_full, v_full = _AllGatherSeq.apply(k, self.sp_group), _AllGatherSeq.apply(v, self.sp_group)
        k_ext, v_ext, lead = sp_extended_kv(k_full, v_full, rank, q.shape[2], self.num_sink, self.window_size)
        return sp_local_attention(q, k_ext, v_ext, lead, self.num_sink, self.window_size, s_aux=s_aux)
