"""HuggingFace / verl boundary: route ``_flash_attention_forward`` to the HIP sink attention.

Mirrors the behaviour of the reference's ``sink_attention/verl_patch.py``:
  * ``patch_verl_with_sink_attention()`` (no arguments, idempotent) replaces
    ``_flash_attention_forward`` at the same three attribute sites (:215-234);
  * the replacement pops gpt-oss's ``s_aux`` from kwargs (:66), falls back to the saved original
    for varlen / packed / non-causal / masked / softcapped calls (:73-93), routes N_q != N_kv to
    the decode kernel (:98-126), slices ``s_aux`` to this rank's heads under Ulysses SP
    (:134-154), uses ``sliding_window or N`` as the window with num_sink = 0 (:156-174);
  * ``softmax_scale``, ``dropout``, ``target_dtype`` ... are accepted and ignored as there.
Two deliberate differences: the [B,N,H,D] activations are handed to the kernel as strided
views and the output is produced directly in [B,N,H,D] memory (the reference makes four
transpose+contiguous copies per call, :119-126,:164-177); ``unpatch_verl`` resets the saved
original so a later re-patch works (the reference leaves it set, making re-patching a no-op).
"""
import os

import torch

from . import _hf_args as _hf
from .decode_kernel import sink_decode_attention
from .sink_flash_attention import _sink_flash_attention_ex

_original_flash_attention_forward = None

# SURVEY section 8 f-3.  False (default) = the reference's behaviour: packed / varlen calls go back to the original
# flash attention, which drops s_aux.  True (or env SINK_ATTENTION_VARLEN=1) = run every packed sequence through the
# sink attention kernels (sink_attention.varlen) so s_aux and the per-layer window are honoured there too.
ENABLE_VARLEN = os.environ.get("SINK_ATTENTION_VARLEN", "0") == "1"


def _local_s_aux(s_aux, H_q):
    """s_aux for the heads this rank holds.  [H_total] == H_q: as is; a multiple of H_q: the
    Ulysses slice of this SP rank; anything else: dropped (verl_patch.py:134-154)."""
    if s_aux is None:
        return None
    H_total = s_aux.shape[0]
    if H_total == H_q:
        return s_aux
    if H_total > H_q and H_total % H_q == 0:
        try:
            from verl.utils.ulysses import get_ulysses_sequence_parallel_rank
            sp_rank = get_ulysses_sequence_parallel_rank()
        except (ImportError, RuntimeError):
            sp_rank = 0
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                sp_rank = torch.distributed.get_rank() % (H_total // H_q)
        return s_aux[sp_rank * H_q:(sp_rank + 1) * H_q]
    return None


def _is_packed(position_ids) -> bool:
    return _hf.is_packed(position_ids)


def _sink_flash_attention_forward(query_states, key_states, value_states, attention_mask, query_length,
                                  *args, **kwargs):
    """Replacement for transformers' ``_flash_attention_forward``; tensors are [B, N, H, D]."""
    kw = _hf.bind(args, kwargs)
    s_aux = kw.pop("s_aux", None)
    is_causal, sliding_window = kw.get("is_causal", True), kw.get("sliding_window")
    position_ids = kw.get("position_ids")
    varlen = _hf.wants_varlen(kw)
    packed = position_ids is not None and query_states.size(0) > 0 and _hf.is_packed(position_ids)
    plain = is_causal and attention_mask is None and kw.get("softcap") is None

    if (ENABLE_VARLEN and (varlen or packed) and plain
            and query_states.shape[0] == 1 and query_states.shape[1] == key_states.shape[1]):
        from .varlen import seq_bounds_from_position_ids, sink_flash_attention_varlen
        if varlen:     # device offsets + the longest length: no host synchronisation
            cu, max_len = kw["cu_seq_lens_q"], int(kw["max_length_q"])
        else:
            cu, max_len = seq_bounds_from_position_ids(position_ids[0] if position_ids.dim() > 1 else position_ids), None
        T = query_states.shape[1]
        out = sink_flash_attention_varlen(query_states.transpose(1, 2), key_states.transpose(1, 2),
                                          value_states.transpose(1, 2), cu, num_sink=0,
                                          window_size=sliding_window if sliding_window is not None else T,
                                          s_aux=_local_s_aux(s_aux, query_states.shape[2]), max_seqlen=max_len)
        return out.transpose(1, 2).contiguous()
    if varlen or packed or not plain:
        if s_aux is not None:
            kw["s_aux"] = s_aux
        return _original_flash_attention_forward(query_states, key_states, value_states, attention_mask, query_length,
                                                 **kw)

    N_q, H_q = query_states.shape[1], query_states.shape[2]
    N_kv = key_states.shape[1]
    s_aux_local = _local_s_aux(s_aux, H_q)
    q = query_states.transpose(1, 2)        # [B, H, N, D] strided views, no copies
    k = key_states.transpose(1, 2)
    v = value_states.transpose(1, 2)

    if N_q == 1 and N_kv != 1:              # decode step against a KV cache
        out = sink_decode_attention(q, k, v, s_aux=s_aux_local)       # [B, H, 1, D]
        return out.transpose(1, 2).contiguous()
    # N_q < N_kv with N_q > 1 (chunked prefill against a cache; the reference would hand it to the decode kernel and
    # fail its N_q == 1 assert): the prefill kernels take the queries as the last N_q key positions

    # full-attention layer (sliding_window None): every row sees all the keys before it, i.e. the window is the KEY
    # count (N_q would cut a chunk's rows off from the cached keys)
    window_size = sliding_window if sliding_window is not None else N_kv
    out = _sink_flash_attention_ex(q, k, v, 0, window_size, s_aux=s_aux_local, out_bnhd=True)
    return out.transpose(1, 2)              # already contiguous [B, N, H, D]


def _patch_sites():
    import transformers.modeling_flash_attention_utils as fa_utils
    sites = [fa_utils]
    try:
        from transformers.integrations import flash_attention
        sites.append(flash_attention)
    except (ImportError, AttributeError):
        pass
    try:
        import verl.models.transformers.monkey_patch as verl_mp
        sites.append(verl_mp)
    except (ImportError, AttributeError):
        pass
    return sites


def patch_verl_with_sink_attention():
    """Install the sink attention kernel wherever ``_flash_attention_forward`` is looked up:
    transformers.modeling_flash_attention_utils, transformers.integrations.flash_attention and
    (if importable) verl.models.transformers.monkey_patch.  Calling it twice is a no-op."""
    global _original_flash_attention_forward
    if _original_flash_attention_forward is not None:
        return
    sites = _patch_sites()
    _original_flash_attention_forward = sites[0]._flash_attention_forward
    for mod in sites:
        mod._flash_attention_forward = _sink_flash_attention_forward
    print("[SinkAttention] Patched _flash_attention_forward with the MI355X HIP sink attention "
          "(s_aux from kwargs, per-layer sliding_window, Ulysses s_aux slicing)")


def unpatch_verl():
    """Put the original ``_flash_attention_forward`` back at every patched site."""
    global _original_flash_attention_forward
    if _original_flash_attention_forward is None:
        return
    for mod in _patch_sites():
        mod._flash_attention_forward = _original_flash_attention_forward
    _original_flash_attention_forward = None
    print("[SinkAttention] Restored original flash attention")
