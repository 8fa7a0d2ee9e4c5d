"""``model.generate()`` with the sink attention kernels and the sink + sliding-window KV cache.

Same surface as the reference's ``sink_attention/generate_patch.py``: ``patch_for_generation(model=None, num_sink=4,
window_size=4096)`` replaces transformers' ``_flash_attention_forward`` (``modeling_flash_attention_utils`` and
``integrations.flash_attention``, :142-168) and returns a fresh ``SinkAttentionCache`` to hand to ``generate()`` as
``past_key_values``; ``unpatch_generation()`` restores the original (:170-187).  The replacement routes prefill
(N_q > 1) to the prefill kernel with the configured ``num_sink`` / ``window_size`` and a single-token step to the decode
kernel over whatever K/V the cache handed back (:121-131); varlen / packed / non-causal calls go to the saved original
(:86-107).  Like the reference it ignores ``s_aux`` unless asked (``honor_s_aux=True``, our addition: gpt-oss passes its
sinks parameter in kwargs and dropping it changes the logits).

Differences: the [B,N,H,D] activations go to the kernels as strided views and the prefill output is produced in
[B,N,H,D] memory (no transpose+contiguous copies, :113-116,:134); patching twice keeps the TRUE original (the reference
would save its own replacement as "original" and recurse on fallback).
"""
from . import _hf_args as _hf
from .cache import SinkAttentionCache
from .decode_kernel import sink_decode_attention
from .sink_flash_attention import _sink_flash_attention_ex

_original_flash_attention_forward = None

_GENERATION_CONFIG = {"num_sink": 4, "window_size": 4096, "enabled": False, "honor_s_aux": False}


def _generation_flash_attention_forward(query_states, key_states, value_states, attention_mask, query_length,
                                        *args, **kwargs):
    """Replacement for transformers' ``_flash_attention_forward`` during generation; tensors are [B, N, H, D]."""
    kw = _hf.bind(args, kwargs)
    position_ids = kw.get("position_ids")
    packed = position_ids is not None and query_states.size(0) > 0 and _hf.is_packed(position_ids)
    if _hf.wants_varlen(kw) or packed or not kw.get("is_causal", True):
        return _original_flash_attention_forward(query_states, key_states, value_states, attention_mask, query_length,
                                                 **kw)

    s_aux = kw.get("s_aux") if _GENERATION_CONFIG["honor_s_aux"] else None
    if s_aux is not None and s_aux.shape[0] != query_states.shape[2]:
        s_aux = None
    q = query_states.transpose(1, 2)         # [B, H, N, D] strided views, no copies
    k = key_states.transpose(1, 2)
    v = value_states.transpose(1, 2)
    if q.shape[2] > 1:                        # prefill (N_q <= N_kv): the kernel masks (sink + window)
        out = _sink_flash_attention_ex(q, k, v, _GENERATION_CONFIG["num_sink"], _GENERATION_CONFIG["window_size"],
                                       s_aux=s_aux, out_bnhd=True)
        return out.transpose(1, 2)            # already contiguous [B, N, H, D]
    out = sink_decode_attention(q, k, v, s_aux=s_aux)     # decode: every key the cache returned is valid
    return out.transpose(1, 2).contiguous()


def _sites():
    import transformers.modeling_flash_attention_utils as fa_utils
    sites = [fa_utils]
    try:
        from transformers.integrations import flash_attention
        sites.append(flash_attention)
    except (ImportError, AttributeError):
        pass
    return sites


def patch_for_generation(model=None, num_sink: int = 4, window_size: int = 4096,
                         honor_s_aux: bool = False) -> SinkAttentionCache:
    """Patch transformers for generation with sink attention; returns the cache to pass as ``past_key_values``.
    ``model`` is unused (kept for the reference's call signature): the patch is global."""
    global _original_flash_attention_forward
    _GENERATION_CONFIG.update(num_sink=num_sink, window_size=window_size, enabled=True, honor_s_aux=honor_s_aux)
    sites = _sites()
    if _original_flash_attention_forward is None:
        _original_flash_attention_forward = sites[0]._flash_attention_forward
    for mod in sites:
        mod._flash_attention_forward = _generation_flash_attention_forward
    return SinkAttentionCache(num_sink=num_sink, window_size=window_size)


def unpatch_generation():
    """Restore the original flash attention forward."""
    global _original_flash_attention_forward
    if _original_flash_attention_forward is None:
        return
    for mod in _sites():
        mod._flash_attention_forward = _original_flash_attention_forward
    _GENERATION_CONFIG["enabled"] = False
    _original_flash_attention_forward = None
