"""Sink + sliding-window KV cache for decode, with a copy-free attention path.

Same observable behaviour as the reference's ``sink_attention/cache.py`` (``SinkCacheLayer`` :29-238,
``SinkAttentionCache`` :241-330): a fixed sink buffer ``[B, H_kv, num_sink, D]`` plus a circular window buffer
``[B, H_kv, window_size, D]``; prefill stores the state and hands the full K/V back (the prefill kernel masks),
a decode step overwrites one ring slot.  ``update()`` / ``get_kv()`` still return the linearised ``[sink, window]``
tensors for callers that want them, but attention itself no longer needs them:

    layer.append(k_new, v_new)                       # one slot written, no torch.cat
    out = layer.decode_attention(q, s_aux=sinks)     # sfa_decode_ring reads sink buffer + ring in place

The reference rebuilds ``[B, H_kv, ns+W, D]`` with up to three ``torch.cat`` per step and measures that copy as
costly as the attention itself (its README, "cache update + decode").  Bookkeeping is host-side Python on torch
tensors and works on any device; ``decode_attention`` needs the HIP library.
"""
from abc import ABC
from typing import List, Optional, Tuple

import torch

try:  # the HF base classes are optional, exactly as in the reference (cache.py:20-26)
    from transformers.cache_utils import Cache as _HFCache, CacheLayerMixin as _HFLayer
    _HAS_HF = True
except ImportError:  # pragma: no cover
    _HAS_HF = False
    _HFCache = object
    _HFLayer = ABC


class SinkCacheLayer(_HFLayer if _HAS_HF else object):
    """One layer's cache: ``num_sink`` pinned leading tokens + a ring of the last ``window_size`` others."""

    def __init__(self, num_sink: int, window_size: int):
        if _HAS_HF:
            super().__init__()
        self.num_sink = int(num_sink)
        self.window_size = int(window_size)
        self.sink_k = self.sink_v = self.window_k = self.window_v = None
        self.sink_len = 0        # valid rows of the sink buffer
        self.window_len = 0      # valid slots of the ring
        self.write_pos = 0       # ring slot the next decoded token goes to
        self.prefilled = False
        self.seen_tokens = 0
        self.is_initialized = False

    # ------------------------------------------------------------------ state
    def lazy_initialization(self, key_states: torch.Tensor, *_, **__):
        B, H_kv, _n, D = key_states.shape
        mk = lambda n: torch.zeros(B, H_kv, n, D, dtype=key_states.dtype, device=key_states.device)
        self.sink_k, self.sink_v = mk(self.num_sink), mk(self.num_sink)
        self.window_k, self.window_v = mk(self.window_size), mk(self.window_size)
        self.is_initialized = True

    def _prefill(self, k, v):
        N = k.shape[2]
        self.seen_tokens = N
        ns = min(N, self.num_sink)
        self.sink_k[:, :, :ns] = k[:, :, :ns]
        self.sink_v[:, :, :ns] = v[:, :, :ns]
        self.sink_len = ns
        rest = N - ns
        if rest <= 0:
            self.window_len, self.write_pos = 0, 0
        elif rest <= self.window_size:
            self.window_k[:, :, :rest] = k[:, :, ns:]
            self.window_v[:, :, :rest] = v[:, :, ns:]
            self.window_len = rest
            self.write_pos = rest % self.window_size
        else:   # keep only the newest window_size tokens; the ring is full and wraps at slot 0
            self.window_k.copy_(k[:, :, N - self.window_size:])
            self.window_v.copy_(v[:, :, N - self.window_size:])
            self.window_len, self.write_pos = self.window_size, 0
        self.prefilled = True
        return k, v

    def append(self, k: torch.Tensor, v: torch.Tensor) -> None:
        """Write decoded token(s) ``[B, H_kv, n, D]`` into the ring (oldest evicted).  No linearisation."""
        if not self.is_initialized:
            self.lazy_initialization(k)
        if not self.prefilled:      # first tokens ever: same placement as a prefill (sinks first)
            self._prefill(k, v)
            return
        for i in range(k.shape[2]):
            self.seen_tokens += 1
            if self.window_size > 0:
                self.window_k[:, :, self.write_pos] = k[:, :, i]
                self.window_v[:, :, self.write_pos] = v[:, :, i]
                self.write_pos = (self.write_pos + 1) % self.window_size
                self.window_len = min(self.window_len + 1, self.window_size)
        self.prefilled = True

    def update(self, key_states, value_states, cache_kwargs: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Reference-compatible update: prefill returns the full input K/V, decode returns linearised [sink, window]."""
        if not self.is_initialized:
            self.lazy_initialization(key_states)
        if not self.prefilled:
            return self._prefill(key_states, value_states)
        self.append(key_states, value_states)
        return self.get_kv()

    def get_kv(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Chronological ``[B, H_kv, sink_len + window_len, D]`` copy (oldest first) - only for callers that need it."""
        ks, vs = [self.sink_k[:, :, :self.sink_len]], [self.sink_v[:, :, :self.sink_len]]
        if self.window_len > 0:
            if self.window_len < self.window_size or self.write_pos == 0:
                ks.append(self.window_k[:, :, :self.window_len])
                vs.append(self.window_v[:, :, :self.window_len])
            else:
                ks += [self.window_k[:, :, self.write_pos:], self.window_k[:, :, :self.write_pos]]
                vs += [self.window_v[:, :, self.write_pos:], self.window_v[:, :, :self.write_pos]]
        return torch.cat(ks, dim=2), torch.cat(vs, dim=2)

    # -------------------------------------------------------------- attention
    def decode_attention(self, q: torch.Tensor, s_aux: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Single-query attention of ``q [B, H_q, 1, D]`` over the cached keys, reading both buffers in place."""
        from .decode_kernel import sink_decode_attention_ring
        return sink_decode_attention_ring(q, self.sink_k, self.sink_v, self.sink_len, self.window_k, self.window_v,
                                          self.window_len, s_aux=s_aux)

    def decode_step(self, q: torch.Tensor, k_new: torch.Tensor, v_new: torch.Tensor,
                    s_aux: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One generation step = ``append(k_new, v_new)`` + ``decode_attention(q)``, as ONE kernel pass once the cache
        is in steady state (``sfa_decode_ring_step``: the kernel stores the token into its ring slot and attends over
        the updated cache; no ``torch.cat``, no separate slot-write launches)."""
        steady = (self.prefilled and self.window_size > 0 and k_new.shape[2] == 1 and self.window_k is not None
                  and self.window_k.is_cuda)
        if not steady:
            self.append(k_new, v_new)
            return self.decode_attention(q, s_aux=s_aux)
        pos = self.write_pos
        new_len = min(self.window_len + 1, self.window_size)
        out = self._ring_step(q, k_new, v_new, new_len, pos, s_aux)
        self.seen_tokens += 1
        self.write_pos = (pos + 1) % self.window_size
        self.window_len = new_len
        return out

    def _ring_step(self, q, k_new, v_new, new_len, pos, s_aux):
        """sfa_decode_ring_step with the per-layer constants (buffer descriptors, workspace) built once: at B=1 the
        step is host-bound, so the Python work per token is kept to the four per-call descriptors."""
        import math
        from . import _native as N
        st = getattr(self, "_step_state", None)
        key = (self.sink_k.data_ptr(), self.window_k.data_ptr(), q.shape, q.dtype)
        if st is None or st["key"] != key:
            B, H_q, _one, D = q.shape
            H_kv = self.sink_k.shape[1]
            if q.dtype != self.window_k.dtype or q.dtype not in N.SFA_DTYPE:
                raise TypeError("q and the cache buffers must share one dtype")
            assert _one == 1 and H_q % H_kv == 0 and (D * q.element_size()) % 16 == 0
            lib = N.lib()
            ws_bytes = lib.sfa_decode_workspace_bytes(B, H_q, H_kv, self.num_sink + self.window_size, D,
                                                      N.SFA_DTYPE[q.dtype])
            st = dict(key=key, lib=lib, scale=1.0 / math.sqrt(D),
                      descs=[N.desc(t) for t in (self.sink_k, self.sink_v, self.window_k, self.window_v)],
                      # owned across calls and zeroed once: the one-pass decode keeps its arrival counters there
                      ws=torch.zeros((max(int(ws_bytes), 256),), device=q.device, dtype=torch.uint8))
            self._step_state = st
        N.require_gpu(q, k_new, v_new, s_aux)
        if k_new.dtype != q.dtype or v_new.dtype != q.dtype or k_new.shape != v_new.shape:
            raise TypeError("k_new / v_new must be [B, H_kv, 1, D] tensors of q's dtype")
        q, k_new, v_new = N.unit_inner(q.detach()), N.unit_inner(k_new.detach()), N.unit_inner(v_new.detach())
        s_aux_f = s_aux.detach().contiguous().float() if s_aux is not None else None
        out = torch.empty(q.shape, device=q.device, dtype=q.dtype)
        sk, sv, wk, wv = st["descs"]
        with torch.cuda.device(q.device):
            rc = st["lib"].sfa_decode_ring_step(N.desc(q), sk, sv, self.sink_len, wk, wv, new_len, pos, N.desc(k_new),
                                                N.desc(v_new), N.desc(out),
                                                s_aux_f.data_ptr() if s_aux_f is not None else None,
                                                st["ws"].data_ptr(), st["ws"].numel(), st["scale"],
                                                self._decode_flags(N), N.stream_ptr(q.device))
        N.check(rc, "sfa_decode_ring_step")
        return out

    one_pass = False     # opt-in: SFA_FLAG_DECODE_ONE_PASS (last-arriver fold inside the split kernel, one launch)

    def _decode_flags(self, N) -> int:
        # Measured on MI355X (round 3, fence-free fold: write-through partial stores, sc1 loads; profiles/r03_kbench_decode_1pass.log):
        # B=1, 4100 keys 21.2 vs 19.2 us with two launches (the fold's loads are round trips to the memory side, dearer than
        # a back-to-back launch), 132 keys (one split) 15.8 vs 19.2 us - so two launches stay the default.
        return N.FLAG_DECODE_ONE_PASS if self.one_pass else 0

    # ------------------------------------------------ device-resident state (hipGraph capture)
    def enable_device_state(self) -> torch.Tensor:
        """Move the ring bookkeeping {sink_len, window_len, write_pos} to a device int32 tensor so that
        ``decode_step_dyn`` needs no host-side integers: a whole generation step can then be captured with
        ``torch.cuda.graph`` and replayed (the kernels read and advance the state themselves).  Call after prefill."""
        assert self.is_initialized and self.prefilled and self.window_size > 0, "prefill the cache first"
        self._dev_state = torch.tensor([self.sink_len, self.window_len, self.write_pos], dtype=torch.int32,
                                       device=self.window_k.device)
        return self._dev_state

    def pull_state(self) -> None:
        """Refresh the host-side counters from the device state (one small device-to-host copy)."""
        st = getattr(self, "_dev_state", None)
        if st is not None:
            sl, wl, wp = st.tolist()
            W = self.window_size
            if wl < W:                       # still filling: one slot per step
                steps = wl - self.window_len
            elif self.window_len < W:        # filled up since the last pull: the step that took the last slot wrapped
                steps = (W - self.window_len) + wp          # write_pos to 0, every later step advanced it by one
            else:                            # full before and after (steps counted modulo the ring size)
                steps = (wp - self.write_pos) % W
            self.seen_tokens += steps
            self.sink_len, self.window_len, self.write_pos = sl, wl, wp

    def decode_step_dyn(self, q: torch.Tensor, k_new: torch.Tensor, v_new: torch.Tensor,
                        s_aux: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``decode_step`` with the state on the device (``sfa_decode_ring_step_dyn``): capturable into a hipGraph.
        ``out`` (optional, [B,H_q,1,D]) lets the caller keep a static output buffer across replays."""
        import math
        from . import _native as N
        st = getattr(self, "_dev_state", None)
        assert st is not None, "call enable_device_state() first"
        ss = getattr(self, "_step_state", None)
        key = (self.sink_k.data_ptr(), self.window_k.data_ptr(), q.shape, q.dtype)
        if ss is None or ss["key"] != key:
            B, H_q, _one, D = q.shape
            lib = N.lib()
            ws_bytes = lib.sfa_decode_workspace_bytes(B, H_q, self.sink_k.shape[1], self.num_sink + self.window_size, D,
                                                      N.SFA_DTYPE[q.dtype])
            ss = dict(key=key, lib=lib, scale=1.0 / math.sqrt(D),
                      descs=[N.desc(t) for t in (self.sink_k, self.sink_v, self.window_k, self.window_v)],
                      # owned across calls and zeroed once: the one-pass decode keeps its arrival counters there
                      ws=torch.zeros((max(int(ws_bytes), 256),), device=q.device, dtype=torch.uint8))
            self._step_state = ss
        N.require_gpu(q, k_new, v_new, s_aux)
        if k_new.dtype != q.dtype or v_new.dtype != q.dtype or q.dtype != self.window_k.dtype:
            raise TypeError("q, k_new, v_new and the cache buffers must share one dtype")
        q, k_new, v_new = N.unit_inner(q.detach()), N.unit_inner(k_new.detach()), N.unit_inner(v_new.detach())
        s_aux_f = s_aux.detach().contiguous().float() if s_aux is not None else None
        if out is None:
            out = torch.empty(q.shape, device=q.device, dtype=q.dtype)
        sk, sv, wk, wv = ss["descs"]
        with torch.cuda.device(q.device):
            rc = ss["lib"].sfa_decode_ring_step_dyn(N.desc(q), sk, sv, wk, wv, N.desc(k_new), N.desc(v_new), N.desc(out),
                                                    s_aux_f.data_ptr() if s_aux_f is not None else None, st.data_ptr(),
                                                    ss["ws"].data_ptr(), ss["ws"].numel(), ss["scale"],
                                                    self._decode_flags(N), N.stream_ptr(q.device))
        N.check(rc, "sfa_decode_ring_step_dyn")
        return out

    # ------------------------------------------------------- HF layer surface
    def get_seq_length(self, *_, **__) -> int:
        return self.sink_len + self.window_len

    def get_mask_sizes(self, cache_position, *_, **__) -> Tuple[int, int]:
        return self.get_seq_length(), 0

    def get_max_cache_shape(self) -> int:
        return self.num_sink + self.window_size

    def get_max_length(self) -> int:   # abstract in transformers >= 5 (the reference predates it)
        return self.num_sink + self.window_size

    def reorder_cache(self, beam_idx: torch.LongTensor):
        if self.sink_k is None:
            return
        idx = beam_idx.to(self.sink_k.device)
        self.sink_k, self.sink_v = self.sink_k.index_select(0, idx), self.sink_v.index_select(0, idx)
        self.window_k, self.window_v = self.window_k.index_select(0, idx), self.window_v.index_select(0, idx)


class SinkAttentionCache(_HFCache if _HAS_HF else object):
    """Per-layer ``SinkCacheLayer``s behind the transformers ``Cache`` interface (layers created on first use)."""

    def __init__(self, num_sink: int = 4, window_size: int = 4096):
        self.num_sink, self.window_size = num_sink, window_size
        self._seen_tokens = 0
        if _HAS_HF:
            super().__init__(layer_class_to_replicate=None, layers=[])
        else:  # pragma: no cover
            self.layers: List[SinkCacheLayer] = []

    def __len__(self):
        return len(self.layers)

    def __getitem__(self, idx):
        return self.layers[idx]

    def __repr__(self):
        return (f"SinkAttentionCache(num_sink={self.num_sink}, window_size={self.window_size}, "
                f"layers={len(self.layers)}, seen_tokens={self._seen_tokens})")

    def _layer(self, layer_idx: int) -> SinkCacheLayer:
        while len(self.layers) <= layer_idx:
            self.layers.append(SinkCacheLayer(self.num_sink, self.window_size))
        return self.layers[layer_idx]

    def update(self, key_states, value_states, layer_idx: int, cache_kwargs: Optional[dict] = None):
        out = self._layer(layer_idx).update(key_states, value_states, cache_kwargs)
        if layer_idx == 0:
            self._seen_tokens = self.layers[0].seen_tokens
        return out

    def append(self, key_states, value_states, layer_idx: int) -> None:
        """Copy-free decode update of one layer (pair with ``self[layer_idx].decode_attention``)."""
        self._layer(layer_idx).append(key_states, value_states)
        if layer_idx == 0:
            self._seen_tokens = self.layers[0].seen_tokens

    def decode_step(self, q, key_states, value_states, layer_idx: int, s_aux=None):
        """Fused cache update + attention of one layer for one new token (``SinkCacheLayer.decode_step``)."""
        out = self._layer(layer_idx).decode_step(q, key_states, value_states, s_aux=s_aux)
        if layer_idx == 0:
            self._seen_tokens = self.layers[0].seen_tokens
        return out

    def get_seq_length(self, layer_idx: int = 0, *_, **__) -> int:
        return self.layers[layer_idx].get_seq_length() if layer_idx < len(self.layers) else 0

    def get_max_cache_length(self) -> int:
        return self.num_sink + self.window_size

    def reorder_cache(self, beam_idx):
        for layer in self.layers:
            layer.reorder_cache(beam_idx)

    @property
    def seen_tokens(self) -> int:
        return self._seen_tokens
