"""GPU parity: sink_decode_attention vs the CPU oracle.  Ports of the reference's
tests/test_decode_kernel.py (tolerances :76,:181,:223) and the decode rows of
tests/test_inference.py (fp32 at 1e-4, :86; the cache is out of scope, so the keys it
would hand over -- sink tokens + last ``window`` tokens -- are gathered directly)."""
import pytest
import torch

import golden_util as G
from oracle import sink_oracle as O
from util import assert_close, maxdiff, rand

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _op():
    from sink_attention import sink_decode_attention
    return sink_decode_attention


def _case(B, Hq, Hkv, Nkv, D, dtype, aux_scale=None, seed=42, aux_dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    q, k, v = rand((B, Hq, 1, D), g, dtype), rand((B, Hkv, Nkv, D), g, dtype), rand((B, Hkv, Nkv, D), g, dtype)
    sa = rand((Hq,), g, torch.float32, aux_scale).to(aux_dtype) if aux_scale is not None else None
    return q, k, v, sa


def _run(q, k, v, sa):
    out = _op()(q.to(DEV), k.to(DEV), v.to(DEV), s_aux=None if sa is None else sa.to(DEV))
    ref = O.decode_dense(q, k, v, sa)
    return out, ref


@pytest.mark.parametrize("Nkv", [64, 256, 1024, 4096])
def test_basic(Nkv):
    out, ref = _run(*_case(1, 8, 8, Nkv, 128, torch.bfloat16))
    assert out.shape == (1, 8, 1, 128) and out.dtype == torch.bfloat16
    assert_close(out, ref.bfloat16(), 1e-2, 1e-2)


def test_fp16():
    out, ref = _run(*_case(1, 8, 8, 512, 128, torch.float16))
    assert_close(out, ref.half(), 1e-2, 1e-2)


def test_batch_size_2():
    out, ref = _run(*_case(2, 8, 8, 512, 128, torch.bfloat16))
    assert_close(out, ref.bfloat16(), 1e-2, 1e-2)


@pytest.mark.parametrize("D", [64, 128, 256])
def test_head_dims(D):
    out, ref = _run(*_case(1, 4, 4, 512, D, torch.bfloat16))
    assert_close(out, ref.bfloat16(), 1e-2, 1e-2)


@pytest.mark.parametrize("Hq,Hkv", [(16, 4), (32, 8), (8, 1), (6, 2), (64, 8)])
def test_gqa(Hq, Hkv):
    out, ref = _run(*_case(1, Hq, Hkv, 512, 128, torch.bfloat16))
    assert_close(out, ref.bfloat16(), 1e-2, 1e-2)


@pytest.mark.parametrize("Nkv", [64, 256, 1024, 4096])
def test_s_aux_correctness(Nkv):
    out, ref = _run(*_case(1, 8, 8, Nkv, 128, torch.bfloat16, aux_scale=2.0))
    assert_close(out, ref.bfloat16(), 1e-2, 1e-2)


def test_s_aux_gqa():
    out, ref = _run(*_case(1, 32, 8, 512, 128, torch.bfloat16, aux_scale=3.0))
    assert_close(out, ref.bfloat16(), 1e-2, 1e-2)


def test_s_aux_absorbs_mass():
    q, k, v, _ = _case(1, 4, 4, 256, 128, torch.bfloat16)
    no_sink = _op()(q.to(DEV), k.to(DEV), v.to(DEV), s_aux=None)
    big = _op()(q.to(DEV), k.to(DEV), v.to(DEV), s_aux=torch.full((4,), 100.0, dtype=torch.bfloat16, device=DEV))
    assert big.abs().max().item() < 0.01
    assert (no_sink - big).abs().max().item() > 0.01


def test_s_aux_zero_is_noop_equivalent():
    q, k, v, _ = _case(1, 4, 4, 512, 128, torch.bfloat16)
    z = _op()(q.to(DEV), k.to(DEV), v.to(DEV), s_aux=torch.zeros(4, dtype=torch.bfloat16, device=DEV))
    n = _op()(q.to(DEV), k.to(DEV), v.to(DEV), s_aux=None)
    assert (z - n).abs().max().item() < 1.0


@pytest.mark.parametrize("Nkv", [8192, 16384])
def test_long_kv(Nkv):
    out, ref = _run(*_case(1, 32, 8, Nkv, 128, torch.bfloat16, aux_scale=2.0))
    assert_close(out, ref.bfloat16(), 2e-2, 2e-2)


def test_non_block_aligned():
    out, ref = _run(*_case(1, 8, 8, 300, 128, torch.bfloat16, aux_scale=1.0))
    assert_close(out, ref.bfloat16(), 1e-2, 1e-2)


@pytest.mark.parametrize("name", G.names("f6"))
def test_golden_decode(name):
    g = G.load(name)
    dt = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}[int(g["meta"][6])]
    sa = g.get("s_aux")
    out = _op()(g["q"].to(dt).to(DEV), g["k"].to(dt).to(DEV), g["v"].to(dt).to(DEV),
                s_aux=None if sa is None else sa.to(DEV))
    tol = {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dt]
    assert maxdiff(out, g["o_kernel"]) < tol and maxdiff(out, g["o_eager"]) < tol


# ---- tests/test_inference.py decode rows: last row of full sink attention == decode over the cached keys
@pytest.mark.parametrize("Hq,Hkv,D,ns,W,N,seed,dtype,tol", [
    (4, 4, 64, 4, 8, 20, 42, torch.float32, 1e-4),      # test_decode_correctness :54-86
    (8, 2, 64, 4, 8, 16, 123, torch.float32, 1e-4),     # test_decode_correctness_gqa
    (2, 2, 32, 2, 3, 14, 99, torch.float32, 1e-4),      # test_decode_with_eviction
    (4, 4, 64, 4, 8, 20, 42, torch.float16, 1e-2),      # test_decode_fp16 :228
])
def test_decode_equals_last_row_of_prefill(Hq, Hkv, D, ns, W, N, seed, dtype, tol):
    g = torch.Generator().manual_seed(seed)
    q, k, v = rand((1, Hq, N, D), g, dtype), rand((1, Hkv, N, D), g, dtype), rand((1, Hkv, N, D), g, dtype)
    for pos in range(max(ns, 1), N):
        ref, _ = O.sink_attention_dense(q[:, :, :pos + 1], k[:, :, :pos + 1], v[:, :, :pos + 1], ns, W)
        keep = sorted(set(range(min(ns, pos + 1))) | set(range(max(0, pos - W + 1), pos + 1)))
        idx = torch.tensor(keep)
        out = _op()(q[:, :, pos:pos + 1].to(DEV), k[:, :, idx].to(DEV), v[:, :, idx].to(DEV))
        assert_close(out, ref[:, :, pos:pos + 1].to(dtype), tol, tol, f"pos {pos}")


def test_odd_head_dims_and_strided_cache():
    for D, dt in ((80, torch.bfloat16), (16, torch.float16), (32, torch.float32), (40, torch.float32)):
        out, ref = _run(*_case(2, 8, 2, 129, D, dt, aux_scale=1.0, aux_dtype=torch.float32))
        assert maxdiff(out, ref) < (2e-5 if dt == torch.float32 else 1.6e-2), (D, dt)
    # K/V as a prefix view of a larger cache buffer (non-contiguous along batch/head)
    g = torch.Generator().manual_seed(5)
    q = rand((2, 8, 1, 128), g, torch.bfloat16)
    kbuf, vbuf = rand((2, 2, 1024, 128), g, torch.bfloat16), rand((2, 2, 1024, 128), g, torch.bfloat16)
    out = _op()(q.to(DEV), kbuf.to(DEV)[:, :, :517], vbuf.to(DEV)[:, :, :517])
    assert maxdiff(out, O.decode_dense(q, kbuf[:, :, :517], vbuf[:, :, :517])) < 1.6e-2


def test_baseline_c5_slice():
    """BASELINE config 5 shape (H=32 MHA, D=128, N_kv=131072) on a B=1 slice + size-independent property:
    splitting the cache in two halves and merging by hand equals one call."""
    B, H, D, Nkv = 1, 32, 128, 131072
    g = torch.Generator().manual_seed(11)
    q = rand((B, H, 1, D), g, torch.bfloat16).to(DEV)
    k = rand((B, H, Nkv, D), g, torch.bfloat16).to(DEV)
    v = rand((B, H, Nkv, D), g, torch.bfloat16).to(DEV)
    out = _op()(q, k, v)
    ref = O.decode_dense(q.cpu(), k.cpu(), v.cpu())
    assert_close(out, ref.bfloat16(), 2e-2, 2e-2, "C5 slice")


# ---- decode over the sink buffer + window ring in place (sfa_decode_ring)
@pytest.mark.parametrize("dtype,Hq,Hkv,D,ns,W,prefill,steps", [
    (torch.float32, 4, 4, 64, 4, 8, 11, 14),       # wraps the ring several times (tests/test_inference.py:157-199)
    (torch.bfloat16, 8, 2, 128, 4, 64, 100, 40),   # GQA, ring full from the prefill on
    (torch.float16, 8, 8, 128, 2, 300, 5, 30),     # ring never fills
    (torch.bfloat16, 16, 2, 80, 0, 33, 50, 10),    # no sink buffer rows at all, D = 80, s_aux
])
def test_ring_decode_equals_linearised_decode_and_oracle(dtype, Hq, Hkv, D, ns, W, prefill, steps):
    from sink_attention import SinkCacheLayer, sink_decode_attention
    g = torch.Generator().manual_seed(21)
    Ntot = prefill + steps
    q = rand((1, Hq, Ntot, D), g, dtype)
    k, v = rand((1, Hkv, Ntot, D), g, dtype), rand((1, Hkv, Ntot, D), g, dtype)
    sa = rand((Hq,), g, torch.float32, 0.7) if ns == 0 else None
    layer = SinkCacheLayer(ns, W)
    layer.append(k[:, :, :prefill].to(DEV), v[:, :, :prefill].to(DEV))
    tol = {torch.float32: 2e-5, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
    for pos in range(prefill, Ntot):
        layer.append(k[:, :, pos:pos + 1].to(DEV), v[:, :, pos:pos + 1].to(DEV))
        qd = q[:, :, pos:pos + 1].to(DEV)
        out = layer.decode_attention(qd, s_aux=None if sa is None else sa.to(DEV))
        from sink_attention import _native
        assert "ring" in _native.last_path()
        # (1) same as the reference flow: linearise with get_kv(), then the plain decode
        kl, vl = layer.get_kv()
        assert maxdiff(out, sink_decode_attention(qd, kl, vl, s_aux=None if sa is None else sa.to(DEV))) < tol
        # (2) the CPU oracle on the keys the cache policy keeps
        keep = torch.tensor(sorted(set(range(min(ns, pos + 1))) | set(range(max(ns, pos - W + 1), pos + 1))))
        ref = O.decode_dense(q[:, :, pos:pos + 1], k[:, :, keep], v[:, :, keep], sa)
        assert maxdiff(out, ref) < tol, pos


def test_ring_decode_c5_policy_shape():
    """BASELINE config 5 with its cache policy: B=32, H=32, D=128, num_sink=4, window=4096 -> 4100 keys per step."""
    from sink_attention import SinkCacheLayer
    B, H, D, ns, W = 32, 32, 128, 4, 4096
    g = torch.Generator().manual_seed(5)
    k, v = rand((B, H, ns + W + 7, D), g, torch.bfloat16), rand((B, H, ns + W + 7, D), g, torch.bfloat16)
    q = rand((B, H, 1, D), g, torch.bfloat16)
    layer = SinkCacheLayer(ns, W)
    layer.append(k[:, :, :ns + W].to(DEV), v[:, :, :ns + W].to(DEV))
    for i in range(7):
        layer.append(k[:, :, ns + W + i:ns + W + i + 1].to(DEV), v[:, :, ns + W + i:ns + W + i + 1].to(DEV))
    out = layer.decode_attention(q.to(DEV))
    keep = torch.tensor(list(range(ns)) + list(range(ns + 7, ns + W + 7)))
    ref = O.decode_dense(q[:4], k[:4][:, :, keep], v[:4][:, :, keep])
    assert_close(out[:4], ref.bfloat16(), 1e-2, 1e-2, "C5 policy")


@pytest.mark.parametrize("dtype,Hq,Hkv,D,ns,W,prefill", [
    (torch.float16, 32, 8, 128, 4, 64, 30),      # ring not full yet, then fills and wraps
    (torch.bfloat16, 8, 8, 64, 2, 16, 100),      # full ring from the start (prefill longer than the window)
    (torch.float32, 4, 1, 80, 0, 8, 3),          # no sinks, MQA, fp32, D=80
    (torch.float16, 8, 2, 128, 4, 32, 2)])       # prefill shorter than num_sink (sink buffer partly filled)
def test_fused_cache_step_matches_append_then_decode(dtype, Hq, Hkv, D, ns, W, prefill):
    """decode_step (sfa_decode_ring_step: slot store + attention in one pass) against append() + decode_attention()
    on a twin cache, and against the oracle on the chronological keys, across wrap-around."""
    from sink_attention import _native
    from sink_attention.cache import SinkCacheLayer
    g = torch.Generator().manual_seed(91)
    B = 2
    a, b = SinkCacheLayer(ns, W), SinkCacheLayer(ns, W)
    kp, vp = rand((B, Hkv, prefill, D), g, dtype).to(DEV), rand((B, Hkv, prefill, D), g, dtype).to(DEV)
    a.update(kp, vp)
    b.update(kp, vp)
    sa = rand((Hq,), g, torch.float32, 0.5).to(DEV)
    fused_seen = False
    for step in range(W + 5):
        q = rand((B, Hq, 1, D), g, dtype).to(DEV)
        kn, vn = rand((B, Hkv, 1, D), g, dtype).to(DEV), rand((B, Hkv, 1, D), g, dtype).to(DEV)
        out = a.decode_step(q, kn, vn, s_aux=sa)
        fused_seen |= "ringstep" in _native.last_path()
        b.append(kn, vn)
        ref = b.decode_attention(q, s_aux=sa)
        assert torch.equal(out, ref), step
        assert (a.write_pos, a.window_len, a.sink_len, a.seen_tokens) == (b.write_pos, b.window_len, b.sink_len, b.seen_tokens)
        assert torch.equal(a.window_k, b.window_k) and torch.equal(a.window_v, b.window_v)
    assert fused_seen
    kc, vc = a.get_kv()
    o_ref = O.decode_dense(q.cpu(), kc.cpu(), vc.cpu(), sa.cpu())
    assert maxdiff(out, o_ref) < (1e-4 if dtype == torch.float32 else 1e-2)


def test_decode_step_in_a_hip_graph_with_device_state():
    """One generation step of a 3-layer cache captured ONCE with torch.cuda.graph and replayed: the kernels take
    {sink_len, window_len, write_pos} from device memory and advance them, so no host integer changes between replays.
    Every replay must equal the eager fused step on a twin cache, across ring wrap-around."""
    from sink_attention.cache import SinkCacheLayer
    g = torch.Generator().manual_seed(97)
    B, Hq, Hkv, D, ns, W, L = 2, 8, 2, 128, 4, 24, 3
    dt = torch.float16
    graph_layers = [SinkCacheLayer(ns, W) for _ in range(L)]
    eager_layers = [SinkCacheLayer(ns, W) for _ in range(L)]
    for a, b in zip(graph_layers, eager_layers):
        kp, vp = rand((B, Hkv, 10, D), g, dt).to(DEV), rand((B, Hkv, 10, D), g, dt).to(DEV)
        a.update(kp, vp)
        b.update(kp, vp)
        a.enable_device_state()
    sa = rand((Hq,), g, torch.float32, 0.5).to(DEV)
    # static buffers of the captured step
    qs = [torch.zeros(B, Hq, 1, D, device=DEV, dtype=dt) for _ in range(L)]
    ks = [torch.zeros(B, Hkv, 1, D, device=DEV, dtype=dt) for _ in range(L)]
    vs = [torch.zeros(B, Hkv, 1, D, device=DEV, dtype=dt) for _ in range(L)]
    outs = [torch.zeros(B, Hq, 1, D, device=DEV, dtype=dt) for _ in range(L)]

    def step():
        for i, layer in enumerate(graph_layers):
            layer.decode_step_dyn(qs[i], ks[i], vs[i], s_aux=sa, out=outs[i])

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):           # warm-up outside the graph (advances the device state by one token)
        step()
    torch.cuda.current_stream().wait_stream(side)
    for i, b in enumerate(eager_layers):    # twin takes the same (all-zero) token
        b.decode_step(qs[i], ks[i], vs[i], s_aux=sa)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for i, b in enumerate(eager_layers):    # the capture itself does not execute: nothing to mirror; replay now
        pass
    for t in range(W + 6):
        for i in range(L):
            qs[i].copy_(rand((B, Hq, 1, D), g, dt))
            ks[i].copy_(rand((B, Hkv, 1, D), g, dt))
            vs[i].copy_(rand((B, Hkv, 1, D), g, dt))
        graph.replay()
        torch.cuda.synchronize()
        for i, b in enumerate(eager_layers):
            ref = b.decode_step(qs[i], ks[i], vs[i], s_aux=sa)
            assert torch.equal(outs[i], ref), (t, i)
    for a, b in zip(graph_layers, eager_layers):
        a.pull_state()
        assert (a.sink_len, a.window_len, a.write_pos) == (b.sink_len, b.window_len, b.write_pos)
        assert torch.equal(a.window_k, b.window_k) and torch.equal(a.window_v, b.window_v)


def test_pull_state_counts_tokens_across_the_fill_boundary():
    """Device-resident state stepped from a not-yet-full ring past the point where it fills: pull_state() must count
    every step (fill steps + the steps after the wrap), and the twin stepped on the host agrees in every counter."""
    from sink_attention.cache import SinkCacheLayer
    g = torch.Generator().manual_seed(5)
    B, Hq, Hkv, D, ns, W = 1, 4, 2, 64, 2, 12
    a, b = SinkCacheLayer(ns, W), SinkCacheLayer(ns, W)
    kp, vp = rand((B, Hkv, ns + W - 3, D), g, torch.float16).to(DEV), rand((B, Hkv, ns + W - 3, D), g, torch.float16).to(DEV)
    a.update(kp, vp)
    b.update(kp, vp)
    assert a.window_len == W - 3
    a.enable_device_state()
    for _ in range(8):     # 3 steps fill the ring, 5 more wrap
        q = rand((B, Hq, 1, D), g, torch.float16).to(DEV)
        kn, vn = rand((B, Hkv, 1, D), g, torch.float16).to(DEV), rand((B, Hkv, 1, D), g, torch.float16).to(DEV)
        o1 = a.decode_step_dyn(q, kn, vn)
        o2 = b.decode_step(q, kn, vn)
        assert torch.equal(o1, o2)
    a.pull_state()
    assert (a.sink_len, a.window_len, a.write_pos, a.seen_tokens) == (b.sink_len, b.window_len, b.write_pos, b.seen_tokens)


def test_ring_step_workspace_serves_every_fill_level():
    """B * H_kv = 64: the split count is capped by the workgroup target and is not monotonic in the key count; the
    workspace the cache sizes ONCE for the full ring must serve every fill level (it raised SFA_ERR_WORKSPACE
    mid-generation before the workspace query became monotonic)."""
    from sink_attention.cache import SinkCacheLayer
    g = torch.Generator().manual_seed(6)
    B, Hq, Hkv, D, ns, W = 8, 32, 8, 128, 4, 16384
    a = SinkCacheLayer(ns, W)
    n0 = 6600
    kp = rand((B, Hkv, n0, D), g, torch.bfloat16).to(DEV)
    a.update(kp, kp)
    q = rand((B, Hq, 1, D), g, torch.bfloat16).to(DEV)
    kn = rand((B, Hkv, 1, D), g, torch.bfloat16).to(DEV)
    for _ in range(80):          # crosses N_kv = 6657, where the planned split count jumps above the full ring's
        out = a.decode_step(q, kn, kn)
    kc, vc = a.get_kv()
    assert maxdiff(out[:1], O.decode_dense(q[:1].cpu(), kc[:1].cpu(), vc[:1].cpu())) < 2e-2


def test_one_pass_decode_matches_two_launches():
    """SFA_FLAG_DECODE_ONE_PASS (the last split to arrive folds the partials inside the split kernel) against the
    default two-launch decode, through the cache's fused step, over several steps (the arrival counters must be left
    zero by every call)."""
    from sink_attention import _native
    from sink_attention.cache import SinkCacheLayer
    g = torch.Generator().manual_seed(101)
    B, Hq, Hkv, D, ns, W = 2, 16, 2, 128, 4, 600
    a, b = SinkCacheLayer(ns, W), SinkCacheLayer(ns, W)
    a.one_pass = True
    kp, vp = rand((B, Hkv, 700, D), g, torch.bfloat16).to(DEV), rand((B, Hkv, 700, D), g, torch.bfloat16).to(DEV)
    a.update(kp, vp)
    b.update(kp, vp)
    sa = rand((Hq,), g, torch.float32, 0.5).to(DEV)
    for step in range(6):
        q = rand((B, Hq, 1, D), g, torch.bfloat16).to(DEV)
        kn, vn = rand((B, Hkv, 1, D), g, torch.bfloat16).to(DEV), rand((B, Hkv, 1, D), g, torch.bfloat16).to(DEV)
        o1 = a.decode_step(q, kn, vn, s_aux=sa)
        assert "1pass" in _native.last_path()
        o2 = b.decode_step(q, kn, vn, s_aux=sa)
        assert "1pass" not in _native.last_path()
        assert torch.equal(o1, o2), step
