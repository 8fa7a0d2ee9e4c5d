"""GPU: packed (varlen) sequences, SURVEY section 8 f-3.  Oracle = the CPU oracle applied per sequence."""
import pytest
import torch

from oracle import sink_oracle as O
from util import dkdv_kernel_name, maxdiff, rand

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _per_seq_oracle(q, k, v, do, cu, ns, W, sa):
    o = torch.zeros(q.shape, dtype=torch.float64)
    dq, dk, dv = torch.zeros(q.shape, dtype=torch.float64), torch.zeros(k.shape, dtype=torch.float64), torch.zeros(
        v.shape, dtype=torch.float64)
    dsa = torch.zeros(q.shape[1], dtype=torch.float64)
    for a, b in zip(cu[:-1], cu[1:]):
        sl = (slice(None), slice(None), slice(a, b))
        o[sl], _ = O.sink_attention_dense(q[sl], k[sl], v[sl], ns, W, sa)
        g = O.sink_attention_bwd_dense(q[sl], k[sl], v[sl], do[sl], ns, W, sa)
        dq[sl], dk[sl], dv[sl] = g[0], g[1], g[2]
        if sa is not None:
            dsa += g[3]
    return o, dq, dk, dv, dsa


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_varlen_fwd_bwd_matches_per_sequence_oracle(dtype, dkdv):
    from sink_attention.varlen import sink_flash_attention_varlen
    g = torch.Generator().manual_seed(31)
    Hq, Hkv, D, ns, W = 8, 2, 128, 2, 40
    cu = [0, 70, 71, 300, 517]
    T = cu[-1]
    q, k, v = rand((1, Hq, T, D), g, dtype), rand((1, Hkv, T, D), g, dtype), rand((1, Hkv, T, D), g, dtype)
    do = rand((1, Hq, T, D), g, dtype)
    sa = rand((Hq,), g, torch.float32, 0.5)
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    sad = sa.to(DEV).requires_grad_(True)
    out = sink_flash_attention_varlen(qd, kd, vd, torch.tensor(cu), num_sink=ns, window_size=W, s_aux=sad)
    out.backward(do.to(DEV))
    o_r, dq_r, dk_r, dv_r, dsa_r = _per_seq_oracle(q, k, v, do, cu, ns, W, sa)
    to, tg = (2e-5, 2e-4) if dtype == torch.float32 else (2e-2, 1.5e-1)
    assert maxdiff(out, o_r) < to
    assert maxdiff(qd.grad, dq_r) < tg and maxdiff(kd.grad, dk_r) < tg and maxdiff(vd.grad, dv_r) < tg
    assert maxdiff(sad.grad, dsa_r) < tg * 10


@pytest.mark.parametrize("Hq,Hkv,D,ns,W,cu", [
    (8, 2, 128, 4, 300, [0, 1000, 1001, 1900, 4000, 4127]),      # long + 1-row + ragged sequences, GQA
    (4, 4, 64, 0, 64, [0, 63, 64, 200, 200, 455]),               # MHA, D=64, an EMPTY sequence in the pack
    (8, 1, 80, 130, 50, [0, 129, 700]),                          # MQA, D=80, sinks longer than a key block
    (4, 2, 96, 2, 100000, [0, 257, 640]),                        # window larger than every sequence
    (4, 1, 96, 0, 4096, [0, 200, 264, 265]),                     # pack longer than 256 rows, every sequence shorter: the
    (1, 1, 80, 130, 300, [0, 200, 400]),                         # row constants must follow the LAUNCH's kernel choice
    (8, 2, 128, 4, 300, [0, 1100, 1101, 2300, 4400]),            # D = 128, W > 256, sinks, SMALL grid (5 blocks x 2 KV heads x 4
                                                                 # sequences < CUs): the rule takes the compiled kernel, no sink split
    (16, 16, 128, 4, 600, [0, 4500, 9000]),                      # ... and a pack whose grid fills the chip: hand-placed under the rule too
    (16, 2, 80, 0, 128, [0, 300, 301, 1500, 2100])])             # gpt-oss sliding layer packed: group of 8, no sinks, W = 128 - the skewed
                                                                 # dK/dV sweep under the rule (sequences of 300 / 1 / 1199 / 600 rows)
def test_varlen_native_kernels_one_launch(Hq, Hkv, D, ns, W, cu, dkdv):
    """The packed kernels (cu_seqlens inside the grid) against the per-sequence oracle, forward and backward, and
    against the sequence-by-sequence path of the same library."""
    from sink_attention import _native
    from sink_attention.varlen import sink_flash_attention_varlen
    from sink_attention.sink_flash_attention import _sink_flash_attention_ex
    g = torch.Generator().manual_seed(33)
    T = cu[-1]
    dtype = torch.bfloat16
    q, k, v = rand((1, Hq, T, D), g, dtype), rand((1, Hkv, T, D), g, dtype), rand((1, Hkv, T, D), g, dtype)
    do = rand((1, Hq, T, D), g, dtype)
    sa = rand((Hq,), g, torch.float32, 0.5)
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    sad = sa.to(DEV).requires_grad_(True)
    out = sink_flash_attention_varlen(qd, kd, vd, cu, num_sink=ns, window_size=W, s_aux=sad)
    assert _native.last_path().startswith("fwd_mfma")
    out.backward(do.to(DEV))
    longest = max(b - a for a, b in zip(cu[:-1], cu[1:]))
    want = dkdv_kernel_name(dkdv, len(cu) - 1, Hkv, longest, longest, D, W, packed=True, ns=ns)
    assert want in _native.last_path(), (want, _native.last_path())
    o_r, dq_r, dk_r, dv_r, dsa_r = _per_seq_oracle(q, k, v, do, cu, ns, W, sa)
    assert maxdiff(out, o_r) < 2e-2
    assert maxdiff(qd.grad, dq_r) < 1.5e-1 and maxdiff(kd.grad, dk_r) < 1.5e-1 and maxdiff(vd.grad, dv_r) < 1.5e-1
    assert maxdiff(sad.grad, dsa_r) < 1.5
    # same numbers as running each sequence alone through the ordinary op (same kernels, same tile walk)
    for a, b in zip(cu[:-1], cu[1:]):
        if b > a:
            alone = _sink_flash_attention_ex(qd.detach()[:, :, a:b], kd.detach()[:, :, a:b], vd.detach()[:, :, a:b],
                                             ns, W, s_aux=sad.detach())
            assert torch.equal(alone, out.detach()[:, :, a:b])


def test_varlen_device_cu_seqlens_no_host_sync():
    from sink_attention.varlen import sink_flash_attention_varlen
    g = torch.Generator().manual_seed(34)
    cu = [0, 100, 356]
    q, k, v = (rand((1, 4, 356, 64), g, torch.float16).to(DEV) for _ in range(3))
    ref = sink_flash_attention_varlen(q, k, v, cu, num_sink=2, window_size=32)
    cud = torch.tensor(cu, dtype=torch.int32, device=DEV)
    out = sink_flash_attention_varlen(q, k, v, cud, num_sink=2, window_size=32, max_seqlen=300)   # upper bound is fine
    assert torch.equal(out, ref)


def test_boundary_routes_packed_batches_when_enabled():
    import sink_attention.verl_patch as vp
    g = torch.Generator().manual_seed(32)
    T, Hq, Hkv, D = 96, 4, 2, 64
    cu = [0, 40, 96]
    qs, ks, vs = rand((1, T, Hq, D), g, torch.float32), rand((1, T, Hkv, D), g, torch.float32), rand(
        (1, T, Hkv, D), g, torch.float32)
    sa = rand((Hq,), g, torch.float32, 0.5)
    pid = torch.cat([torch.arange(40), torch.arange(56)]).view(1, T)
    t = lambda x: x.transpose(1, 2)
    ref, *_ = _per_seq_oracle(t(qs), t(ks), t(vs), torch.zeros(1, Hq, T, D), cu, 0, 24, sa)
    old = vp.ENABLE_VARLEN
    vp.ENABLE_VARLEN = True
    try:
        out = vp._sink_flash_attention_forward(qs.to(DEV), ks.to(DEV), vs.to(DEV), None, T, is_causal=True,
                                               position_ids=pid.to(DEV), sliding_window=24, s_aux=sa.to(DEV))
        assert out.shape == (1, T, Hq, D) and maxdiff(t(out), ref) < 2e-5
        cuq = torch.tensor(cu, dtype=torch.int32, device=DEV)
        out2 = vp._sink_flash_attention_forward(qs.to(DEV), ks.to(DEV), vs.to(DEV), None, T, is_causal=True,
                                                cu_seq_lens_q=cuq, cu_seq_lens_k=cuq, max_length_q=56, max_length_k=56,
                                                sliding_window=24, s_aux=sa.to(DEV))
        assert maxdiff(t(out2), ref) < 2e-5
    finally:
        vp.ENABLE_VARLEN = old
