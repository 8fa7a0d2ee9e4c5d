"""CPU oracle for sink flash attention -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a CPU restatement (plain torch, fp64 by default) of the algorithm
of RulinShao/sink-flash-attention-kernel's hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the shipped op (``sink_attention/*``) never does and fails loudly when the
HIP library is missing.

Parity pinning: every function below is checked in ``tests/test_oracle.py``
against golden vectors produced by running the reference itself in the build
container (``tests/golden/make_golden.py``: the reference's eager oracles and
its Triton kernels under ``TRITON_INTERPRET=1``), committed as
``tests/golden/*.npz``.

What each function restates (paths relative to the reference repo):

* ``valid_mask``              sink_attention/sink_flash_attention.py:11,30-39
* ``sink_attention_dense``    tests/test_sink_attention.py:15-50 (naive_sink_attention),
                              tests/test_s_aux.py:16-72 (reference_attention_with_s_aux),
                              kernel semantics sink_flash_attention.py:134-146,183-194
* ``sink_attention_bwd_dense``sink_flash_attention.py:568-667 (Delta, dQ, dK, dV, ds_aux)
* ``sink_attention_banded`` / ``sink_attention_bwd_banded``
                              the same maths evaluated row-block by row-block over
                              only the sink + window key ranges (two-range walk of
                              sink_flash_attention.py:151-180) so it scales to N=8192+
* ``decode_dense``            tests/test_decode_kernel.py:19-55 (reference_decode_attention),
                              sink_attention/decode_kernel.py:205-226
* ``pair_count``              SURVEY.md section 8(d) algorithmic FLOP formula
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

NEG_INF = float("-inf")


def valid_mask(n_q_rows: torch.Tensor, n_k_cols: torch.Tensor, num_sink: int, window: int) -> torch.Tensor:
    """valid(i, j) = (j <= i) and (j < num_sink or j >= i - W + 1).

    ``n_q_rows`` [R] and ``n_k_cols`` [C] are absolute positions.
    Follows sink_flash_attention.py:30-39 / tests/test_sink_attention.py:35-41.
    """
    i = n_q_rows.view(-1, 1)
    j = n_k_cols.view(1, -1)
    causal = j <= i
    sink = j < num_sink
    win = j >= (i - window + 1)
    return causal & (sink | win)


def pair_count(N: int, num_sink: int, window: int) -> int:
    """Number of valid (i, j) pairs per head."""
    total = 0
    for i in range(N):
        total += min(i + 1, max(window, 0)) + min(num_sink, max(0, i - max(window, 0) + 1))
    return total


def pair_count_closed(N: int, ns: int, W: int) -> int:
    Wp = min(max(W, 0), N)
    T = N - Wp
    t = Wp * (Wp + 1) // 2 + T * Wp
    if T >= ns:
        t += ns * (ns + 1) // 2 + (T - ns) * ns
    else:
        t += T * (T + 1) // 2
    return t


def _expand_kv(x: torch.Tensor, groups: int) -> torch.Tensor:
    return x if groups == 1 else x.repeat_interleave(groups, dim=1)


def sink_attention_dense(
    q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
    num_sink: int, window: int, s_aux: Optional[torch.Tensor] = None,
    dtype: torch.dtype = torch.float64,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Dense masked softmax attention.  Returns (O [B,Hq,N,D], LSE [B,Hq,N]).

    k / v may hold MORE rows than q (N_kv >= N_q): the queries are then the LAST N_q positions of the key sequence
    (row i sits at position i + N_kv - N_q), the convention of chunked prefill / a sequence-parallel rank that got
    its halo keys prepended.  Not a reference feature (it asserts N_q == N_kv, sink_flash_attention.py:494-498).

    Mask by -inf (tests/test_sink_attention.py:43); a row with no valid key and
    no s_aux gives O = 0 (``nan_to_num`` at :45, kernel: l==0 -> 1 at
    sink_flash_attention.py:183) and LSE = -inf.  ``s_aux`` is an extra logit
    per Q head that enters the denominator only (tests/test_s_aux.py:58-68).
    """
    B, Hq, N, D = q.shape
    Hkv = k.shape[1]
    g = Hq // Hkv
    scale = 1.0 / math.sqrt(D)
    qf, kf, vf = q.to(dtype), _expand_kv(k.to(dtype), g), _expand_kv(v.to(dtype), g)
    s = torch.matmul(qf, kf.transpose(-2, -1)) * scale                  # [B,Hq,N,Nk]
    Nk = k.shape[2]
    m = valid_mask(torch.arange(N) + (Nk - N), torch.arange(Nk), num_sink, window)
    s = s.masked_fill(~m, NEG_INF)
    if s_aux is not None:
        col = s_aux.to(dtype).view(1, Hq, 1, 1).expand(B, Hq, N, 1)
        s_all = torch.cat([s, col], dim=-1)
    else:
        s_all = s
    lse = torch.logsumexp(s_all, dim=-1)                                # -inf for empty rows
    p = torch.exp(s - lse.unsqueeze(-1))
    p = torch.nan_to_num(p, nan=0.0)
    o = torch.matmul(p, vf)
    return o, lse


def sink_attention_bwd_dense(
    q, k, v, do, num_sink: int, window: int, s_aux=None, dtype=torch.float64,
):
    """Explicit backward (no autograd): returns dQ, dK, dV, ds_aux (or None).

    Formulas of sink_flash_attention.py:568-667:
      Delta = rowsum(dO * O); P = exp(S - LSE); dV = P^T dO; dP = dO V^T;
      dS = P * (dP - Delta); dQ = scale * dS K; dK = scale * dS^T Q;
      ds_aux[h] = -sum_{b,n} exp(s_aux[h] - LSE) * Delta.
    dK/dV are summed over the GQA group (:648-651).
    """
    B, Hq, N, D = q.shape
    Hkv = k.shape[1]
    g = Hq // Hkv
    scale = 1.0 / math.sqrt(D)
    o, lse = sink_attention_dense(q, k, v, num_sink, window, s_aux, dtype)
    qf, kf, vf = q.to(dtype), _expand_kv(k.to(dtype), g), _expand_kv(v.to(dtype), g)
    dof = do.to(dtype)
    s = torch.matmul(qf, kf.transpose(-2, -1)) * scale
    Nk = k.shape[2]
    m = valid_mask(torch.arange(N) + (Nk - N), torch.arange(Nk), num_sink, window)
    p = torch.exp(s.masked_fill(~m, NEG_INF) - lse.unsqueeze(-1))
    p = torch.nan_to_num(p, nan=0.0)
    delta = (dof * o).sum(-1)                                           # [B,Hq,N]
    dv = torch.matmul(p.transpose(-2, -1), dof)
    dp = torch.matmul(dof, vf.transpose(-2, -1))
    ds = p * (dp - delta.unsqueeze(-1))
    dq = torch.matmul(ds, kf) * scale
    dk = torch.matmul(ds.transpose(-2, -1), qf) * scale
    if g > 1:
        dk = dk.view(B, Hkv, g, Nk, D).sum(2)
        dv = dv.view(B, Hkv, g, Nk, D).sum(2)
    ds_aux = None
    if s_aux is not None:
        sink_prob = torch.exp(s_aux.to(dtype).view(1, Hq, 1) - lse)
        ds_aux = -(sink_prob * delta).sum(dim=(0, 2))
    return dq, dk, dv, ds_aux


def _row_block_keys(r0: int, r1: int, N: int, num_sink: int, window: int) -> torch.Tensor:
    """Key positions any row of [r0, r1) can see: sink range then window range
    (the two-range walk of sink_flash_attention.py:151-180)."""
    ns = min(num_sink, r1)
    w0 = max(r0 - window + 1, ns, 0)
    sink = torch.arange(0, ns)
    win = torch.arange(w0, r1) if r1 > w0 else torch.arange(0)
    return torch.cat([sink, win])


def sink_attention_banded(
    q, k, v, num_sink: int, window: int, s_aux=None, dtype=torch.float64, block: int = 256,
):
    """Same result as ``sink_attention_dense`` without materialising N x N."""
    B, Hq, N, D = q.shape
    Hkv = k.shape[1]
    g = Hq // Hkv
    scale = 1.0 / math.sqrt(D)
    o = torch.zeros(B, Hq, N, D, dtype=dtype)
    lse = torch.full((B, Hq, N), NEG_INF, dtype=dtype)
    for r0 in range(0, N, block):
        r1 = min(N, r0 + block)
        cols = _row_block_keys(r0, r1, N, num_sink, window)
        rows = torch.arange(r0, r1)
        qb = q[:, :, r0:r1].to(dtype)
        kb = _expand_kv(k[:, :, cols].to(dtype), g)
        vb = _expand_kv(v[:, :, cols].to(dtype), g)
        s = torch.matmul(qb, kb.transpose(-2, -1)) * scale
        m = valid_mask(rows, cols, num_sink, window)
        s = s.masked_fill(~m, NEG_INF)
        if s_aux is not None:
            col = s_aux.to(dtype).view(1, Hq, 1, 1).expand(B, Hq, r1 - r0, 1)
            s_all = torch.cat([s, col], dim=-1)
        else:
            s_all = s
        l = torch.logsumexp(s_all, dim=-1)
        p = torch.nan_to_num(torch.exp(s - l.unsqueeze(-1)), nan=0.0)
        o[:, :, r0:r1] = torch.matmul(p, vb)
        lse[:, :, r0:r1] = l
    return o, lse


def sink_attention_bwd_banded(
    q, k, v, do, num_sink: int, window: int, s_aux=None, dtype=torch.float64, block: int = 256,
):
    """Banded explicit backward; same outputs as ``sink_attention_bwd_dense``."""
    B, Hq, N, D = q.shape
    Hkv = k.shape[1]
    g = Hq // Hkv
    scale = 1.0 / math.sqrt(D)
    o, lse = sink_attention_banded(q, k, v, num_sink, window, s_aux, dtype, block)
    dq = torch.zeros(B, Hq, N, D, dtype=dtype)
    dk = torch.zeros(B, Hkv, N, D, dtype=dtype)
    dv = torch.zeros(B, Hkv, N, D, dtype=dtype)
    delta = (do.to(dtype) * o).sum(-1)
    for r0 in range(0, N, block):
        r1 = min(N, r0 + block)
        cols = _row_block_keys(r0, r1, N, num_sink, window)
        rows = torch.arange(r0, r1)
        qb = q[:, :, r0:r1].to(dtype)
        dob = do[:, :, r0:r1].to(dtype)
        kb = _expand_kv(k[:, :, cols].to(dtype), g)
        vb = _expand_kv(v[:, :, cols].to(dtype), g)
        s = torch.matmul(qb, kb.transpose(-2, -1)) * scale
        m = valid_mask(rows, cols, num_sink, window)
        p = torch.exp(s.masked_fill(~m, NEG_INF) - lse[:, :, r0:r1].unsqueeze(-1))
        p = torch.nan_to_num(p, nan=0.0)
        dp = torch.matmul(dob, vb.transpose(-2, -1))
        ds = p * (dp - delta[:, :, r0:r1].unsqueeze(-1))
        dq[:, :, r0:r1] = torch.matmul(ds, kb) * scale
        dkb = torch.matmul(ds.transpose(-2, -1), qb) * scale           # [B,Hq,C,D]
        dvb = torch.matmul(p.transpose(-2, -1), dob)
        C = cols.numel()
        dk.index_add_(2, cols, dkb.view(B, Hkv, g, C, D).sum(2))
        dv.index_add_(2, cols, dvb.view(B, Hkv, g, C, D).sum(2))
    ds_aux = None
    if s_aux is not None:
        sink_prob = torch.exp(s_aux.to(dtype).view(1, Hq, 1) - lse)
        ds_aux = -(sink_prob * delta).sum(dim=(0, 2))
    return dq, dk, dv, ds_aux


def decode_dense(q, k, v, s_aux=None, dtype=torch.float64):
    """Single-query attention over every key handed in (no mask), with the
    optional s_aux logit in the denominator.  q [B,Hq,1,D], k/v [B,Hkv,Nkv,D].
    Follows tests/test_decode_kernel.py:19-55; an empty / fully -inf cache gives
    zeros like decode_kernel.py:220-224 (alpha masked, L clamped)."""
    B, Hq, _, D = q.shape
    Hkv = k.shape[1]
    g = Hq // Hkv
    scale = 1.0 / math.sqrt(D)
    kf, vf = _expand_kv(k.to(dtype), g), _expand_kv(v.to(dtype), g)
    s = torch.matmul(q.to(dtype), kf.transpose(-2, -1)) * scale         # [B,Hq,1,Nkv]
    if s_aux is not None:
        col = s_aux.to(dtype).view(1, Hq, 1, 1).expand(B, Hq, 1, 1)
        s_all = torch.cat([col, s], dim=-1)
    else:
        s_all = s
    lse = torch.logsumexp(s_all, dim=-1, keepdim=True)
    p = torch.nan_to_num(torch.exp(s - lse), nan=0.0)
    return torch.matmul(p, vf)


def flops_fwd(B: int, Hq: int, N: int, D: int, ns: int, W: int) -> int:
    return 4 * D * pair_count_closed(N, ns, W) * B * Hq


def flops_fwd_bwd(B: int, Hq: int, N: int, D: int, ns: int, W: int) -> int:
    return 14 * D * pair_count_closed(N, ns, W) * B * Hq
