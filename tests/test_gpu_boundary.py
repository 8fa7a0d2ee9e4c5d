"""GPU: the HF/verl boundary (_sink_flash_attention_forward) against golden vectors produced by
the reference's own replacement function (tests/golden/f7_boundary.npz), and the C-ABI error paths."""
import ctypes

import pytest
import torch

import golden_util as G
from util import maxdiff

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_boundary_matches_reference_outputs():
    from sink_attention.verl_patch import _sink_flash_attention_forward as fwd
    g = G.load("f7_boundary")
    qs, ks, vs, sa = (g[x].to(DEV) for x in ("qs", "ks", "vs", "s_aux"))
    N = qs.shape[1]
    for key, kw in (("o_none", dict(sliding_window=None, s_aux=sa)), ("o_w16", dict(sliding_window=16, s_aux=sa)),
                    ("o_noaux", dict(sliding_window=16)),
                    ("o_sp", dict(sliding_window=16, s_aux=g["s_aux_big"].to(DEV)))):
        out = fwd(qs, ks, vs, None, N, is_causal=True, **kw)
        assert out.shape == qs.shape and out.is_contiguous()
        assert maxdiff(out, g[key]) < 2e-5, key
    out = fwd(g["qd"].to(DEV), g["kd"].to(DEV), g["vd"].to(DEV), None, 1, is_causal=True, sliding_window=16, s_aux=sa)
    assert out.shape == (1, 1, 4, 64) and maxdiff(out, g["o_dec"]) < 2e-5


def test_boundary_backward_through_bnhd_layout():
    from sink_attention.verl_patch import _sink_flash_attention_forward as fwd
    from oracle import sink_oracle as O
    g = torch.Generator().manual_seed(9)
    B, N, Hq, Hkv, D = 1, 96, 4, 2, 64
    qs = torch.randn(B, N, Hq, D, generator=g).to(DEV).requires_grad_(True)
    ks = torch.randn(B, N, Hkv, D, generator=g).to(DEV).requires_grad_(True)
    vs = torch.randn(B, N, Hkv, D, generator=g).to(DEV).requires_grad_(True)
    sa = (torch.randn(Hq, generator=g) * 0.5).to(DEV).requires_grad_(True)
    do = torch.randn(B, N, Hq, D, generator=g).to(DEV)
    out = fwd(qs, ks, vs, None, N, is_causal=True, sliding_window=24, s_aux=sa)
    out.backward(do)
    t = lambda x: x.detach().cpu().transpose(1, 2)
    dq, dk, dv, dsa = O.sink_attention_bwd_dense(t(qs), t(ks), t(vs), t(do), 0, 24, sa.detach().cpu())
    assert maxdiff(t(qs.grad), dq) < 1e-4 and maxdiff(t(ks.grad), dk) < 1e-4 and maxdiff(t(vs.grad), dv) < 1e-4
    assert maxdiff(sa.grad, dsa) < 1e-4


@pytest.mark.parametrize("sliding_window", [None, 40])
def test_chunked_prefill_against_a_cache(sliding_window):
    """N_q < N_kv with N_q > 1 (a prefill chunk against cached keys; the reference asserts N_q == N_kv): the chunk's
    rows are the LAST N_q key positions.  A full-attention layer (sliding_window=None) must see ALL earlier keys, not
    just the chunk (the window is the key count, not N_q)."""
    from sink_attention.verl_patch import _sink_flash_attention_forward as fwd
    from oracle import sink_oracle as O
    g = torch.Generator().manual_seed(31)
    B, Nq, Nkv, Hq, Hkv, D = 1, 16, 300, 4, 2, 64
    qs = torch.randn(B, Nq, Hq, D, generator=g).bfloat16().to(DEV)
    ks = torch.randn(B, Nkv, Hkv, D, generator=g).bfloat16().to(DEV)
    vs = torch.randn(B, Nkv, Hkv, D, generator=g).bfloat16().to(DEV)
    sa = (torch.randn(Hq, generator=g) * 0.5).to(DEV)
    out = fwd(qs, ks, vs, None, Nq, is_causal=True, sliding_window=sliding_window, s_aux=sa)
    t = lambda x: x.detach().cpu().transpose(1, 2)
    W = sliding_window if sliding_window is not None else Nkv
    ref, _ = O.sink_attention_dense(t(qs), t(ks), t(vs), 0, W, sa.cpu())
    assert out.shape == qs.shape and maxdiff(t(out), ref) < 2e-2


def test_fallback_and_patch_roundtrip_on_gpu():
    import transformers.modeling_flash_attention_utils as fa_utils
    import sink_attention.verl_patch as vp
    orig = fa_utils._flash_attention_forward
    vp.patch_verl_with_sink_attention()
    try:
        assert fa_utils._flash_attention_forward is vp._sink_flash_attention_forward
        g = G.load("f7_boundary")
        qs, ks, vs = (g[x].to(DEV) for x in ("qs", "ks", "vs"))
        out = fa_utils._flash_attention_forward(qs, ks, vs, None, qs.shape[1], is_causal=True, sliding_window=16,
                                                s_aux=g["s_aux"].to(DEV), attn_implementation="flash_attention_2",
                                                layer_idx=0)
        assert maxdiff(out, g["o_w16"]) < 2e-5
    finally:
        vp.unpatch_verl()
    assert fa_utils._flash_attention_forward is orig


def test_c_abi_error_codes():
    from sink_attention import _native as N
    lib = N.lib()
    q = torch.zeros(1, 4, 8, 64, device=DEV)
    k = torch.zeros(1, 3, 8, 64, device=DEV)        # 4 % 3 != 0
    o = torch.zeros_like(q)
    lse = torch.zeros(1, 4, 8, device=DEV)
    st = lib.sfa_fwd(N.desc(q), N.desc(k), N.desc(k), N.desc(o), lse.data_ptr(), None, 0, 8, 0.125, 0,
                     N.stream_ptr(q.device))
    assert st == -1 and b"divisible" in lib.sfa_last_error()
    k2 = torch.zeros(1, 2, 8, 64, device=DEV)
    ws = torch.zeros(256, dtype=torch.uint8, device=DEV)
    st = lib.sfa_bwd(N.desc(q), N.desc(k2), N.desc(k2), N.desc(o), N.desc(o), lse.data_ptr(), None, N.desc(q),
                     N.desc(k2), N.desc(k2), None, ws.data_ptr(), 16, 0, 8, 0.125, 0, N.stream_ptr(q.device))
    assert st == -3 and b"workspace" in lib.sfa_last_error()
    with pytest.raises(AssertionError):
        from sink_attention import sink_decode_attention
        sink_decode_attention(torch.zeros(1, 4, 2, 64, device=DEV), k2, k2)   # N_q must be 1


def test_ops_run_on_the_current_stream_and_capture_into_a_hip_graph():
    """The C ABI launches on the stream it is handed and never allocates or synchronises, so the ops work on a side
    stream and inside torch.cuda.graph capture (hipGraph), forward and backward."""
    from sink_attention import sink_decode_attention, sink_flash_attention
    from util import rand
    g = torch.Generator().manual_seed(61)
    B, Hq, Hkv, N, D, ns, W = 2, 8, 2, 640, 128, 4, 200
    q, k, v, do = (rand(s, g, torch.bfloat16).to(DEV) for s in ((B, Hq, N, D), (B, Hkv, N, D), (B, Hkv, N, D), (B, Hq, N, D)))
    sa = rand((Hq,), g, torch.float32, 0.5).to(DEV)
    qd, kd, vd, sad = (t.clone().requires_grad_(True) for t in (q, k, v, sa))
    ref = sink_flash_attention(qd, kd, vd, ns, W, sad)
    ref.backward(do)
    ref_g = [t.grad.clone() for t in (qd, kd, vd, sad)]
    q1 = rand((B, Hq, 1, D), g, torch.bfloat16).to(DEV)
    ref_dec = sink_decode_attention(q1, k, v, s_aux=sa)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        qs, ks, vs, ss = (t.clone().requires_grad_(True) for t in (q, k, v, sa))
        for _ in range(2):            # warm-up on the side stream (also what graph capture requires)
            for t in (qs, ks, vs, ss):
                t.grad = None
            out = sink_flash_attention(qs, ks, vs, ns, W, ss)
            out.backward(do)
            dec = sink_decode_attention(q1, k, v, s_aux=sa)
        assert torch.equal(out, ref) and torch.equal(dec, ref_dec)
        assert all(torch.equal(a.grad, b) for a, b in zip((qs, ks, vs, ss), ref_g))
        for t in (qs, ks, vs, ss):
            t.grad = None
    torch.cuda.current_stream().wait_stream(side)

    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_g = sink_flash_attention(qs, ks, vs, ns, W, ss)
        out_g.backward(do)
        dec_g = sink_decode_attention(q1, k, v, s_aux=sa)
    with torch.no_grad():
        qs.copy_(q * 0.5)             # new inputs in the captured buffers; the replay must see them
    graph.replay()
    torch.cuda.synchronize()
    q2 = (q * 0.5).clone().requires_grad_(True)
    k2, v2, s2 = (t.clone().requires_grad_(True) for t in (k, v, sa))
    exp = sink_flash_attention(q2, k2, v2, ns, W, s2)
    exp.backward(do)
    assert torch.equal(out_g, exp) and torch.equal(dec_g, ref_dec)
    assert torch.equal(qs.grad, q2.grad) and torch.equal(ks.grad, k2.grad) and torch.equal(vs.grad, v2.grad)
    assert torch.equal(ss.grad, s2.grad)


def test_raw_sfa_bwd_first_call_of_a_thread_under_graph_capture():
    """include/sfa.h, SFA_FLAG_BWD_OVERLAP: the side stream of the dQ / dK-dV overlap is created at a thread's first
    overlapped call on a device, and never while the caller's stream is capturing.  A fresh thread whose FIRST sfa_bwd
    (small grid, overlap requested) runs inside torch.cuda.graph capture must therefore capture a plain single-stream
    backward; after one uncaptured call of the same thread (which creates the stream) a captured call forks / joins the
    side stream inside the graph.  All three against the oracle."""
    import threading
    from sink_attention import _native as N
    from sink_attention import sink_flash_attention
    from util import oracle_bwd, rand, assert_close
    lib = N.lib()
    g = torch.Generator().manual_seed(71)
    B, Hq, Hkv, Nq, D, ns, W = 1, 4, 1, 1024, 128, 4, 300          # 16 dQ workgroups, 4 key blocks: a small grid
    q, k, v, do = (rand(s, g, torch.bfloat16) for s in ((B, Hq, Nq, D), (B, Hkv, Nq, D), (B, Hkv, Nq, D), (B, Hq, Nq, D)))
    dq_r, dk_r, dv_r, _ = oracle_bwd(q, k, v, do, ns, W)
    qd, kd, vd, dod = (t.to(DEV) for t in (q, k, v, do))
    from sink_attention.sink_flash_attention import SinkFlashAttentionFunc
    # forward through the C ABI to get o and lse
    o = torch.empty_like(qd)
    lse = torch.empty(B, Hq, Nq, device=DEV, dtype=torch.float32)
    scale = D ** -0.5
    N.check(lib.sfa_fwd(N.desc(qd), N.desc(kd), N.desc(vd), N.desc(o), lse.data_ptr(), None, ns, W, scale, 0,
                        N.stream_ptr(qd.device)), "sfa_fwd")
    flags = N.FLAG_BWD_OVERLAP
    ws_bytes = lib.sfa_bwd_workspace_bytes(B, Hq, Hkv, Nq, D, N.SFA_DTYPE[qd.dtype], ns, W, flags)
    ws = torch.empty(int(ws_bytes), dtype=torch.uint8, device=DEV)
    grads = [[torch.zeros_like(qd), torch.zeros_like(kd), torch.zeros_like(vd)] for _ in range(3)]
    torch.cuda.synchronize()
    paths, errors = [], []

    def bwd(dst):
        st = lib.sfa_bwd(N.desc(qd), N.desc(kd), N.desc(vd), N.desc(o), N.desc(dod), lse.data_ptr(), None, N.desc(dst[0]),
                         N.desc(dst[1]), N.desc(dst[2]), None, ws.data_ptr(), ws.numel(), ns, W, scale, flags,
                         N.stream_ptr(qd.device))
        N.check(st, "sfa_bwd")
        paths.append(N.last_path())

    def worker():
        try:
            torch.cuda.set_device(qd.device)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                g1 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g1, stream=s, capture_error_mode="thread_local"):
                    bwd(grads[0])                      # first sfa_bwd of this thread: under capture
                g1.replay()
                bwd(grads[1])                          # uncaptured: creates the side stream
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, stream=s, capture_error_mode="thread_local"):
                    bwd(grads[2])                      # captured WITH the fork / join
                g2.replay()
            s.synchronize()
        except Exception as e:      # noqa: BLE001 - reported by the main thread
            errors.append(e)

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    assert "overlap" not in paths[0] and paths[1].endswith("_overlap") and paths[2].endswith("_overlap"), paths
    for dst in grads:
        assert_close(dst[0], dq_r, 5e-2, 5e-2, "dq")
        assert_close(dst[1], dk_r, 5e-2 * max(1.0, dk_r.abs().max().item()), 5e-2, "dk")
        assert_close(dst[2], dv_r, 5e-2 * max(1.0, dv_r.abs().max().item()), 5e-2, "dv")
    assert all(torch.equal(a, b) for a, b in zip(grads[0], grads[1])) and all(torch.equal(a, b) for a, b in zip(grads[1], grads[2]))
