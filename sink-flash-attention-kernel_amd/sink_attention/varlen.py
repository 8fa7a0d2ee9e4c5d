"""Packed (variable-length) sequences for sink flash attention.

The reference cannot handle packed batches: ``verl_patch.py:73-93`` hands them back to stock flash attention,
which silently drops ``s_aux`` (the very thing the reference exists to fix, its README "Why").  Here every
sequence of the pack is run through the same HIP kernels on a strided VIEW of the packed tensors (the C ABI takes
arbitrary B/H/N strides), so the mask and the ``s_aux`` logit restart at every sequence boundary.  Inputs are not
copied; the per-sequence outputs are concatenated once.  Autograd flows through the per-sequence ops.
"""
from typing import List, Sequence, Union

import torch

from .sink_flash_attention import _sink_flash_attention_ex


def seq_bounds_from_position_ids(position_ids: torch.Tensor) -> List[int]:
    """cu_seqlens (host list) of a single packed row: a new sequence starts wherever position_ids is 0."""
    pid = position_ids.reshape(-1)
    starts = (pid == 0).nonzero(as_tuple=False).flatten().tolist()
    if not starts or starts[0] != 0:
        starts = [0] + starts
    return starts + [pid.numel()]


def sink_flash_attention_varlen(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                                cu_seqlens: Union[torch.Tensor, Sequence[int]], num_sink: int = 4,
                                window_size: int = 512, s_aux: torch.Tensor = None) -> torch.Tensor:
    """
    q [1, H_q, T, D], k/v [1, H_kv, T, D]: ``len(cu_seqlens) - 1`` sequences packed along T
    (sequence i = rows cu_seqlens[i] : cu_seqlens[i+1]).  Every sequence is attended independently with
    valid(i, j) = (j <= i) and (j < num_sink or j >= i - window_size + 1) in ITS OWN positions, and its own s_aux
    term.  Returns [1, H_q, T, D].
    """
    assert q.shape[0] == 1 and k.shape[0] == 1 and v.shape[0] == 1, "packed layout: batch dim must be 1"
    cu = cu_seqlens.tolist() if isinstance(cu_seqlens, torch.Tensor) else list(cu_seqlens)
    assert len(cu) >= 2 and cu[0] == 0 and cu[-1] == q.shape[2], f"bad cu_seqlens {cu} for T={q.shape[2]}"
    outs = []
    for a, b in zip(cu[:-1], cu[1:]):
        assert b >= a
        if b == a:
            continue
        outs.append(_sink_flash_attention_ex(q[:, :, a:b], k[:, :, a:b], v[:, :, a:b], num_sink, window_size,
                                             s_aux=s_aux))
    return torch.cat(outs, dim=2) if len(outs) != 1 else outs[0]
