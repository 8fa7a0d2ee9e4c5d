"""GPU: SURVEY section 8 f-4.  The repaired SP attention (per-rank [sinks | halo | local] problem, ranks simulated in
one process) against the full-sequence op and the oracle, and the generation patch's prefill / decode routing with
the sink + ring cache."""
import pytest
import torch

from oracle import sink_oracle as O
from util import dkdv_kernel_name, maxdiff, rand

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("P,n,ns,W", [(4, 64, 4, 100), (2, 96, 0, 50), (3, 40, 6, 200)])
def test_sp_rank_local_attention_matches_full_sequence(P, n, ns, W, dkdv):
    from sink_attention import sink_flash_attention
    from sink_attention.sp_utils import sp_extended_kv, sp_local_attention
    g = torch.Generator().manual_seed(41)
    B, Hq, Hkv, D = 1, 4, 2, 64
    N = P * n
    q, k, v = rand((B, Hq, N, D), g, torch.bfloat16), rand((B, Hkv, N, D), g, torch.bfloat16), rand(
        (B, Hkv, N, D), g, torch.bfloat16)
    sa = rand((Hq,), g, torch.float32, 0.5)
    do = rand((B, Hq, N, D), g, torch.bfloat16)
    ref, _ = O.sink_attention_dense(q, k, v, ns, W, sa)
    g_ref = O.sink_attention_bwd_dense(q, k, v, do, ns, W, sa)

    kd, vd = k.to(DEV).requires_grad_(True), v.to(DEV).requires_grad_(True)
    qd = q.to(DEV).requires_grad_(True)
    sad = sa.to(DEV).requires_grad_(True)
    outs = []
    for r in range(P):                  # what SinkAttentionSPWrapper.forward does on rank r after its all-gather
        k_ext, v_ext, lead = sp_extended_kv(kd, vd, r, n, ns, W)
        outs.append(sp_local_attention(qd[:, :, r * n:(r + 1) * n], k_ext, v_ext, lead, ns, W, s_aux=sad))
    out = torch.cat(outs, dim=2)
    assert maxdiff(out, ref) < 2e-2
    out.backward(do.to(DEV))
    assert maxdiff(qd.grad, g_ref[0]) < 1.5e-1 and maxdiff(kd.grad, g_ref[1]) < 1.5e-1
    assert maxdiff(vd.grad, g_ref[2]) < 1.5e-1 and maxdiff(sad.grad, g_ref[3]) < 1.5
    full = sink_flash_attention(q.to(DEV), k.to(DEV), v.to(DEV), ns, W, sa.to(DEV))
    assert maxdiff(out, full) < 2e-2


def test_generation_forward_prefill_then_decode_with_cache():
    import sink_attention.generate_patch as gp
    from sink_attention import patch_for_generation, unpatch_generation
    g = torch.Generator().manual_seed(42)
    B, Hq, Hkv, D, N, ns, W = 2, 4, 2, 64, 50, 4, 16
    cache = patch_for_generation(None, num_sink=ns, window_size=W)
    try:
        qs, ks, vs = rand((B, N, Hq, D), g, torch.float16), rand((B, N, Hkv, D), g, torch.float16), rand(
            (B, N, Hkv, D), g, torch.float16)
        t = lambda x: x.transpose(1, 2)
        # prefill through the patched entry point: the attention layer hands the FULL K/V the cache returned
        kc, vc = cache.update(t(ks.to(DEV)), t(vs.to(DEV)), 0)
        out = gp._generation_flash_attention_forward(qs.to(DEV), t(kc), t(vc), None, N, is_causal=True)
        ref, _ = O.sink_attention_dense(t(qs), t(ks), t(vs), ns, W, None)
        assert out.shape == (B, N, Hq, D) and out.is_contiguous() and maxdiff(t(out), ref) < 1e-2
        # decode steps: K/V from the cache = [sinks | last W tokens]; reference = the banded row of the full sequence
        k_all, v_all = ks, vs
        for step in range(3):
            q1, k1, v1 = rand((B, 1, Hq, D), g, torch.float16), rand((B, 1, Hkv, D), g, torch.float16), rand(
                (B, 1, Hkv, D), g, torch.float16)
            k_all, v_all = torch.cat([k_all, k1], 1), torch.cat([v_all, v1], 1)
            kc, vc = cache.update(t(k1.to(DEV)), t(v1.to(DEV)), 0)
            assert kc.shape[2] == ns + W
            o1 = gp._generation_flash_attention_forward(q1.to(DEV), t(kc), t(vc), None, 1, is_causal=True)
            n_tot = k_all.shape[1]
            q_pad = torch.cat([torch.zeros(B, n_tot - 1, Hq, D, dtype=torch.float16), q1], 1)
            r, _ = O.sink_attention_dense(t(q_pad), t(k_all), t(v_all), ns, W, None)
            assert maxdiff(t(o1), r[:, :, -1:]) < 1e-2
    finally:
        unpatch_generation()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Hq,Hkv,D,Nq,Nk,ns,W", [
    (8, 2, 128, 200, 517, 4, 100),       # window shorter than the lead
    (4, 4, 64, 64, 1000, 0, 300),        # no sinks, MHA
    (8, 1, 80, 333, 400, 130, 64),       # sinks longer than a key block, MQA, D=80
    (4, 2, 96, 1, 257, 2, 10000),        # a single query row, window longer than everything
    (4, 2, 128, 1024, 1100, 4, 512)])    # many query tiles
def test_fewer_queries_than_keys_fwd_bwd(dtype, Hq, Hkv, D, Nq, Nk, ns, W, dkdv):
    """N_q < N_kv on the MFMA kernels (queries = the last N_q key positions) against the oracle and against the same
    problem padded with zero queries."""
    from sink_attention import _native
    from sink_attention.sink_flash_attention import _sink_flash_attention_ex
    g = torch.Generator().manual_seed(51)
    B = 2
    q, do = rand((B, Hq, Nq, D), g, dtype), rand((B, Hq, Nq, D), g, dtype)
    k, v = rand((B, Hkv, Nk, D), g, dtype), rand((B, Hkv, Nk, D), g, dtype)
    sa = rand((Hq,), g, torch.float32, 0.5)
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    sad = sa.to(DEV).requires_grad_(True)
    out = _sink_flash_attention_ex(qd, kd, vd, ns, W, s_aux=sad)
    assert out.shape == (B, Hq, Nq, D) and "mfma" in _native.last_path()
    out.backward(do.to(DEV))
    want = dkdv_kernel_name(dkdv, B, Hkv, Nq, Nk, D, W, ns=ns)
    assert want in _native.last_path(), (want, _native.last_path())
    o_r, _ = O.sink_attention_dense(q, k, v, ns, W, sa)
    dq_r, dk_r, dv_r, dsa_r = O.sink_attention_bwd_dense(q, k, v, do, ns, W, sa)
    to, tg = (1e-2, 5e-2) if dtype == torch.float16 else (2e-2, 1.5e-1)
    assert maxdiff(out, o_r) < to
    assert kd.grad.shape == (B, Hkv, Nk, D)
    assert maxdiff(qd.grad, dq_r) < tg and maxdiff(kd.grad, dk_r) < tg and maxdiff(vd.grad, dv_r) < tg
    assert maxdiff(sad.grad, dsa_r) < tg * 10
    # the padded formulation (zero queries for the leading key rows) gives the same rows
    qp = torch.cat([torch.zeros(B, Hq, Nk - Nq, D, dtype=dtype), q], dim=2).to(DEV)
    padded = _sink_flash_attention_ex(qp, kd.detach(), vd.detach(), ns, W, s_aux=sad.detach())[:, :, Nk - Nq:]
    assert maxdiff(out, padded.double()) < 2e-3
