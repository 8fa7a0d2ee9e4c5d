"""sink_decode_attention: single-query (decode) attention with s_aux, MI355X / HIP.

Same public contract as the reference's ``sink_attention/decode_kernel.py:120-226``:
q [B,H_q,1,D] against every key handed in (no mask: windowing is the cache's job),
optional ``s_aux`` [H_q] folded in as a virtual KV split (m = s_aux, l = 1, o = 0).
Both phases (split-KV streaming and the reduction, which the reference does in
~10 PyTorch ops) are HIP kernels behind ``sfa_decode`` (include/sfa.h).
The head dim no longer has to be a power of two; a K/V row must be a multiple of 16 bytes.
"""
import math

import torch

from . import _native as N


def sink_decode_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                          s_aux: torch.Tensor = None) -> torch.Tensor:
    N.require_gpu(q, k, v, s_aux)
    B, H_q, N_q, D = q.shape
    H_kv = k.shape[1]
    N_kv = k.shape[2]
    assert N_q == 1, f"sink_decode_attention requires N_q=1, got {N_q}"
    assert H_q % H_kv == 0, f"H_q ({H_q}) must be divisible by H_kv ({H_kv})"
    assert k.shape == (B, H_kv, N_kv, D) and v.shape == k.shape, "k/v must be [B, H_kv, N_kv, D]"
    if q.dtype not in N.SFA_DTYPE or k.dtype != q.dtype or v.dtype != q.dtype:
        raise TypeError(f"q/k/v must share one dtype in (float32, float16, bfloat16); got "
                        f"{q.dtype}, {k.dtype}, {v.dtype}")
    row_bytes = D * q.element_size()
    assert row_bytes % 16 == 0 and row_bytes <= 1024, f"D={D}: a K/V row must be a multiple of 16 bytes, <= 1 KiB"
    scale = 1.0 / math.sqrt(D)

    def rows16(t):
        t = N.unit_inner(t)
        es = t.element_size()
        if t.data_ptr() % 16 or any((t.stride(i) * es) % 16 for i in range(3)):
            t = t.contiguous()
        return t

    q, k, v = rows16(q.detach()), rows16(k.detach()), rows16(v.detach())
    s_aux_f = None
    if s_aux is not None:
        assert s_aux.shape == (H_q,), f"s_aux shape must be [H_q={H_q}], got {s_aux.shape}"
        s_aux_f = s_aux.detach().contiguous().float()
    out = torch.empty((B, H_q, 1, D), device=q.device, dtype=q.dtype)
    lib = N.lib()
    ws_bytes = lib.sfa_decode_workspace_bytes(B, H_q, H_kv, N_kv, D, N.SFA_DTYPE[q.dtype])
    ws = torch.empty((max(int(ws_bytes), 256),), device=q.device, dtype=torch.uint8)
    with torch.cuda.device(q.device):
        st = lib.sfa_decode(N.desc(q), N.desc(k), N.desc(v), N.desc(out),
                            s_aux_f.data_ptr() if s_aux_f is not None else None, ws.data_ptr(), ws.numel(),
                            scale, 0, N.stream_ptr(q.device))
    N.check(st, "sfa_decode")
    return out


def sink_decode_attention_ring(q: torch.Tensor, sink_k: torch.Tensor, sink_v: torch.Tensor, sink_len: int,
                               window_k: torch.Tensor, window_v: torch.Tensor, window_len: int,
                               s_aux: torch.Tensor = None, k_new: torch.Tensor = None, v_new: torch.Tensor = None,
                               write_pos: int = -1) -> torch.Tensor:
    """Decode over a sink buffer + window ring WITHOUT linearising them (``sfa_decode_ring``).

    Equivalent to ``sink_decode_attention(q, cat(sink_k[:, :, :sink_len], ring in any order), ...)`` - softmax does
    not depend on key order - but reads both cache buffers in place; replaces the ``torch.cat`` copies of the
    reference's ``SinkCacheLayer.get_kv`` (cache.py:185-216).
        q [B,H_q,1,D]; sink_k/v [B,H_kv,num_sink,D] (first ``sink_len`` rows valid);
        window_k/v [B,H_kv,window_size,D] (first ``window_len`` slots valid; all of them once the ring is full).

    With ``k_new`` / ``v_new`` [B,H_kv,1,D] and ``write_pos`` the call is a whole generation step
    (``sfa_decode_ring_step``): the kernel stores the new token into ring slot ``write_pos`` (``window_k`` /
    ``window_v`` are modified IN PLACE) and attends over the cache as it is after that store; ``window_len`` is then the
    number of valid slots AFTER the append.
    """
    N.require_gpu(q, sink_k, sink_v, window_k, window_v, s_aux)
    B, H_q, N_q, D = q.shape
    H_kv = sink_k.shape[1]
    assert N_q == 1, f"sink_decode_attention requires N_q=1, got {N_q}"
    assert H_q % H_kv == 0, f"H_q ({H_q}) must be divisible by H_kv ({H_kv})"
    assert sink_v.shape == sink_k.shape and window_v.shape == window_k.shape
    assert window_k.shape[:2] == sink_k.shape[:2] and window_k.shape[3] == D and sink_k.shape[3] == D
    assert 0 <= sink_len <= sink_k.shape[2] and 0 <= window_len <= window_k.shape[2]
    for t in (sink_k, sink_v, window_k, window_v):
        if t.dtype != q.dtype:
            raise TypeError("q and the cache buffers must share one dtype")
    if q.dtype not in N.SFA_DTYPE:
        raise TypeError(f"unsupported dtype {q.dtype}")
    row_bytes = D * q.element_size()
    assert row_bytes % 16 == 0 and row_bytes <= 1024, f"D={D}: a K/V row must be a multiple of 16 bytes, <= 1 KiB"

    def rows16(t):
        t = N.unit_inner(t.detach())
        es = t.element_size()
        if t.numel() and (t.data_ptr() % 16 or any((t.stride(i) * es) % 16 for i in range(3))):
            t = t.contiguous()
        return t

    fused = k_new is not None
    if fused:
        assert v_new is not None and k_new.shape == (B, H_kv, 1, D) and v_new.shape == k_new.shape
        assert 0 <= write_pos < window_len, f"write_pos {write_pos} outside the {window_len} valid slots"
        for t in (window_k, window_v):   # the kernel writes the ring in place: no silent copies allowed here
            es = t.element_size()
            assert t.stride(-1) == 1 and t.data_ptr() % 16 == 0 and all((t.stride(i) * es) % 16 == 0 for i in range(3)), \
                "fused cache step: the ring buffers must have 16-byte aligned rows"
        if k_new.dtype != q.dtype or v_new.dtype != q.dtype:
            raise TypeError("k_new / v_new must have q's dtype")
        k_new, v_new = rows16(k_new), rows16(v_new)
    q, sink_k, sink_v = (rows16(t) for t in (q, sink_k, sink_v))
    if not fused:
        window_k, window_v = rows16(window_k), rows16(window_v)
    s_aux_f = None
    if s_aux is not None:
        assert s_aux.shape == (H_q,), f"s_aux shape must be [H_q={H_q}], got {s_aux.shape}"
        s_aux_f = s_aux.detach().contiguous().float()
    out = torch.empty((B, H_q, 1, D), device=q.device, dtype=q.dtype)
    lib = N.lib()
    ws_bytes = lib.sfa_decode_workspace_bytes(B, H_q, H_kv, int(sink_len) + int(window_len), D, N.SFA_DTYPE[q.dtype])
    ws = torch.empty((max(int(ws_bytes), 256),), device=q.device, dtype=torch.uint8)
    with torch.cuda.device(q.device):
        if fused:
            st = lib.sfa_decode_ring_step(N.desc(q), N.desc(sink_k), N.desc(sink_v), int(sink_len), N.desc(window_k),
                                          N.desc(window_v), int(window_len), int(write_pos), N.desc(k_new),
                                          N.desc(v_new), N.desc(out),
                                          s_aux_f.data_ptr() if s_aux_f is not None else None, ws.data_ptr(),
                                          ws.numel(), 1.0 / math.sqrt(D), 0, N.stream_ptr(q.device))
        else:
            st = lib.sfa_decode_ring(N.desc(q), N.desc(sink_k), N.desc(sink_v), int(sink_len), N.desc(window_k),
                                     N.desc(window_v), int(window_len), N.desc(out),
                                     s_aux_f.data_ptr() if s_aux_f is not None else None, ws.data_ptr(), ws.numel(),
                                     1.0 / math.sqrt(D), 0, N.stream_ptr(q.device))
    N.check(st, "sfa_decode_ring_step" if fused else "sfa_decode_ring")
    return out
