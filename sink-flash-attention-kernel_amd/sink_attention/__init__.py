"""sink_attention (MI355X / gfx950): drop-in for the hot path of
RulinShao/sink-flash-attention-kernel -- same import name and public names for
``sink_flash_attention``, ``sink_decode_attention``, ``patch_verl_with_sink_attention``
and ``unpatch_verl`` (plus the sink + ring KV cache with a copy-free decode, the generation patch, the
sequence-parallel helpers and packed-sequence attention); kernels are hand-written HIP
behind libsfa.so (no Triton)."""
from .sink_flash_attention import sink_flash_attention, SinkFlashAttentionFunc
from .decode_kernel import sink_decode_attention, sink_decode_attention_ring
from .cache import SinkCacheLayer, SinkAttentionCache
from .verl_patch import patch_verl_with_sink_attention, unpatch_verl
from .generate_patch import patch_for_generation, unpatch_generation
from .sp_utils import prepare_sink_kv_for_sp, reduce_sink_kv_grads, SinkAttentionSPWrapper
from .varlen import sink_flash_attention_varlen
from ._native import set_backward_options

__version__ = "0.1.0"

__all__ = [
    "sink_flash_attention",
    "sink_decode_attention",
    "patch_verl_with_sink_attention",
    "unpatch_verl",
    "SinkFlashAttentionFunc",
    "sink_decode_attention_ring",
    "SinkCacheLayer",
    "SinkAttentionCache",
    "patch_for_generation",
    "unpatch_generation",
    "prepare_sink_kv_for_sp",
    "reduce_sink_kv_grads",
    "SinkAttentionSPWrapper",
    "sink_flash_attention_varlen",
    "set_backward_options",
]
