"""Argument handling shared by the two ``_flash_attention_forward`` replacements (verl_patch, generate_patch).

transformers calls ``_flash_attention_forward(query_states, key_states, value_states, attention_mask, query_length,
is_causal=..., dropout=..., position_ids=..., ...)``; the parameter list has grown over releases.  The replacements
take the five leading tensors / ints by position and everything else as ``*args, **kwargs``: ``bind`` folds stray
positional extras into keywords (order of transformers' signature), so a fallback can forward the call VERBATIM to the
saved original whatever that release accepts, and the replacement only looks at the handful of options it acts on.
"""
from typing import Any, Dict, Tuple

# order of the optional parameters after ``query_length`` in transformers.modeling_flash_attention_utils
OPTIONAL_ORDER: Tuple[str, ...] = (
    "is_causal", "dropout", "position_ids", "softmax_scale", "sliding_window", "use_top_left_mask", "softcap",
    "deterministic", "cu_seq_lens_q", "cu_seq_lens_k", "max_length_q", "max_length_k", "target_dtype", "implementation")


def bind(args: tuple, kwargs: Dict[str, Any]) -> Dict[str, Any]:
    """Merge positional extras into the keyword dict (a keyword given twice is an error, as in a normal call)."""
    if len(args) > len(OPTIONAL_ORDER):
        raise TypeError(f"_flash_attention_forward: {len(args)} positional options, at most {len(OPTIONAL_ORDER)} known")
    kw = dict(kwargs)
    for name, value in zip(OPTIONAL_ORDER, args):
        if name in kw:
            raise TypeError(f"_flash_attention_forward: got multiple values for '{name}'")
        kw[name] = value
    return kw


def is_packed(position_ids) -> bool:
    """True when position_ids [B, N] restart inside a row (several sequences packed in one)."""
    if position_ids is None or position_ids.dim() < 2 or position_ids.size(1) <= 1:
        return False
    return bool((position_ids[:, 1:] < position_ids[:, :-1]).any().item())


def wants_varlen(kw: Dict[str, Any]) -> bool:
    return all(kw.get(n) is not None for n in ("cu_seq_lens_q", "cu_seq_lens_k", "max_length_q", "max_length_k"))
