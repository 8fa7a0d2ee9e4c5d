"""GPU parity: sink_flash_attention forward/backward vs the CPU oracle.

Ports of the reference's test rows with THE SAME tolerances:
tests/test_sink_attention.py, tests/test_s_aux.py, tests/benchmark.py (extended configs).
Everything calls the product op, i.e. goes through the C ABI of libsfa.so.
"""
import math

import pytest
import torch

import golden_util as G
from util import dkdv_kernel_name, assert_close, make_qkv, maxdiff, oracle_bwd, oracle_fwd, rand

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _op():
    from sink_attention import sink_flash_attention
    return sink_flash_attention


def _ex():
    from sink_attention.sink_flash_attention import _sink_flash_attention_ex
    return _sink_flash_attention_ex


def _path():
    from sink_attention import _native
    return _native.last_path()


# ---------------------------------------------------------------- tests/test_sink_attention.py
FWD_CONFIGS = [  # :187-194
    (1, 4, 4, 128, 64, 4, 32),
    (1, 4, 4, 256, 64, 4, 64),
    (1, 8, 2, 256, 64, 4, 64),
    (2, 4, 4, 128, 64, 1, 64),
    (1, 4, 4, 256, 128, 4, 64),
    (1, 4, 4, 512, 64, 16, 128),
]


@pytest.mark.parametrize("B,Hq,Hkv,N,D,ns,W", FWD_CONFIGS)
def test_forward_correctness(B, Hq, Hkv, N, D, ns, W):
    q, k, v, _ = make_qkv(B, Hq, Hkv, N, D, torch.float16)
    out = _op()(q.to(DEV), k.to(DEV), v.to(DEV), num_sink=ns, window_size=W)
    ref, _ = oracle_fwd(q, k, v, ns, W)
    assert_close(out, ref.half(), 1e-2, 1e-2, "fwd fp16")          # :68


@pytest.mark.parametrize("B,Hq,Hkv,N,D,ns,W", [(1, 4, 4, 128, 64, 4, 32), (1, 4, 2, 128, 64, 4, 64)])  # :201-204
def test_backward_correctness(B, Hq, Hkv, N, D, ns, W, dkdv):
    q, k, v, g = make_qkv(B, Hq, Hkv, N, D, torch.float32)
    do = rand((B, Hq, N, D), g, torch.float32)
    # oracle on the fp32 values (the reference compares an fp16 kernel with fp32 autograd, :71-96)
    dq_r, dk_r, dv_r, _ = oracle_bwd(q, k, v, do, ns, W)
    qt, kt, vt = (t.half().to(DEV).requires_grad_(True) for t in (q, k, v))
    out = _op()(qt, kt, vt, num_sink=ns, window_size=W)
    out.backward(do.half().to(DEV))
    assert_close(qt.grad.float(), dq_r, 5e-2, 5e-2, "dq")
    assert_close(kt.grad.float(), dk_r, 5e-2, 5e-2, "dk")
    assert_close(vt.grad.float(), dv_r, 5e-2, 5e-2, "dv")


def test_degenerate_full_attention():   # :99-116
    B, H, N, D = 1, 4, 128, 64
    q, k, v, _ = make_qkv(B, H, H, N, D, torch.float16)
    out = _op()(q.to(DEV), k.to(DEV), v.to(DEV), num_sink=0, window_size=N)
    s = torch.matmul(q.double(), k.double().transpose(-2, -1)) / math.sqrt(D)
    s.masked_fill_(torch.triu(torch.ones(N, N), diagonal=1).bool(), float("-inf"))
    ref = torch.matmul(torch.softmax(s, -1), v.double()).half()
    assert_close(out, ref, 1e-2, 1e-2, "degenerate")


def test_sink_only():                   # :119-131
    q, k, v, _ = make_qkv(1, 2, 2, 64, 64, torch.float16)
    out = _op()(q.to(DEV), k.to(DEV), v.to(DEV), num_sink=4, window_size=1)
    ref, _ = oracle_fwd(q, k, v, 4, 1)
    assert_close(out, ref.half(), 1e-2, 1e-2, "sink only")


def test_memory_efficiency():           # :134-158
    B, H, N, D = 1, 4, 4096, 128
    q, k, v, _ = make_qkv(B, H, H, N, D, torch.float16)
    q, k, v = q.to(DEV), k.to(DEV), v.to(DEV)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    before = torch.cuda.max_memory_allocated()
    out = _op()(q, k, v, num_sink=4, window_size=256)
    torch.cuda.synchronize()
    used = torch.cuda.max_memory_allocated() - before
    assert used < 0.25 * B * H * N * N * 2, f"{used / 1e6:.1f} MB"
    assert torch.isfinite(out).all()


# ---------------------------------------------------------------- tests/test_s_aux.py
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Hq,Hkv", [(8, 8), (8, 2)])
def test_full_causal_with_s_aux(dtype, Hq, Hkv):   # :78-101
    B, N, D = 1, 128, 64
    q, k, v, g = make_qkv(B, Hq, Hkv, N, D, dtype)
    s_aux = rand((Hq,), g, torch.float32, 0.5)
    out = _op()(q.to(DEV), k.to(DEV), v.to(DEV), num_sink=0, window_size=N, s_aux=s_aux.to(DEV))
    ref, _ = oracle_fwd(q, k, v, 0, N, s_aux)
    assert_close(out, ref.to(dtype), 2e-2 if dtype == torch.float16 else 3e-2, 1e-2, "s_aux full")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_sliding_window_with_s_aux(dtype):          # :103-123
    B, Hq, Hkv, N, D, W = 1, 8, 2, 256, 64, 128
    q, k, v, g = make_qkv(B, Hq, Hkv, N, D, dtype)
    s_aux = rand((Hq,), g, torch.float32, 0.5)
    out = _op()(q.to(DEV), k.to(DEV), v.to(DEV), num_sink=0, window_size=W, s_aux=s_aux.to(DEV))
    ref, _ = oracle_fwd(q, k, v, 0, W, s_aux)
    assert_close(out, ref.to(dtype), 2e-2 if dtype == torch.float16 else 3e-2, 1e-2, "s_aux window")


def test_without_s_aux_unchanged():                 # :125-142
    q, k, v, _ = make_qkv(1, 8, 2, 128, 64, torch.float16)
    out = _op()(q.to(DEV), k.to(DEV), v.to(DEV), num_sink=0, window_size=128, s_aux=None)
    ref, _ = oracle_fwd(q, k, v, 0, 128)
    assert_close(out, ref.half(), 2e-2, 1e-2, "no s_aux")


def test_s_aux_absorbs_mass():                      # :144-170
    q, k, v, _ = make_qkv(1, 4, 4, 64, 32, torch.float16)
    q, k, v = q.to(DEV), k.to(DEV), v.to(DEV)
    small = _op()(q, k, v, num_sink=0, window_size=64, s_aux=torch.zeros(4, device=DEV))
    large = _op()(q, k, v, num_sink=0, window_size=64, s_aux=torch.full((4,), 10.0, device=DEV))
    assert large.float().norm().item() < small.float().norm().item()


def test_ds_aux_gradient_exists():                  # :176-195  (fp32 inputs)
    q, k, v, g = make_qkv(1, 4, 4, 64, 32, torch.float32)
    q, k, v = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    s_aux = rand((4,), g, torch.float32).to(DEV).requires_grad_(True)
    out = _op()(q, k, v, num_sink=0, window_size=64, s_aux=s_aux)
    out.sum().backward()
    assert s_aux.grad is not None and s_aux.grad.shape == (4,) and torch.isfinite(s_aux.grad).all()
    for t in (q, k, v):
        assert t.grad is not None and torch.isfinite(t.grad).all()


def test_ds_aux_gradient_numerical():               # :197-239  (fp32, central differences eps=1e-3)
    B, Hq, Hkv, N, D = 1, 2, 2, 32, 16
    q, k, v, g = make_qkv(B, Hq, Hkv, N, D, torch.float32)
    q, k, v = q.to(DEV), k.to(DEV), v.to(DEV)
    s_aux = rand((Hq,), g, torch.float32).to(DEV).requires_grad_(True)
    op = _op()
    op(q, k, v, num_sink=0, window_size=N, s_aux=s_aux).sum().backward()
    analytical = s_aux.grad.clone()
    eps = 1e-3
    numerical = torch.zeros_like(analytical)
    for i in range(Hq):
        sp, sm = s_aux.detach().clone(), s_aux.detach().clone()
        sp[i] += eps
        sm[i] -= eps
        lp = op(q, k, v, num_sink=0, window_size=N, s_aux=sp).sum()
        lm = op(q, k, v, num_sink=0, window_size=N, s_aux=sm).sum()
        numerical[i] = (lp - lm) / (2 * eps)
    assert (analytical - numerical).abs().max().item() < 5e-2
    # and against the oracle's exact gradient
    _, _, _, dsa = oracle_bwd(q, k, v, torch.ones(B, Hq, N, D), 0, N, s_aux.detach())
    assert maxdiff(analytical, dsa) < 1e-4


def test_match_gpt_oss_eager():                     # :267-293
    B, Hq, Hkv, N, D = 1, 16, 4, 128, 64
    q, k, v, g = make_qkv(B, Hq, Hkv, N, D, torch.float16)
    s_aux = rand((Hq,), g, torch.float32, 0.3)
    for W in (N, 128):
        out = _op()(q.to(DEV), k.to(DEV), v.to(DEV), num_sink=0, window_size=W, s_aux=s_aux.to(DEV))
        ref, _ = oracle_fwd(q, k, v, 0, W, s_aux)
        assert maxdiff(out, ref.half()) < 0.05


def test_gpt_oss_head_dim_80():                     # :295-314 (the reference's own kernel cannot run D=80)
    B, Hq, Hkv, N, D = 1, 8, 2, 64, 80
    q, k, v, g = make_qkv(B, Hq, Hkv, N, D, torch.bfloat16)
    s_aux = rand((Hq,), g, torch.float32, 0.5)
    out = _op()(q.to(DEV), k.to(DEV), v.to(DEV), num_sink=0, window_size=N, s_aux=s_aux.to(DEV))
    ref, _ = oracle_fwd(q, k, v, 0, N, s_aux)
    assert maxdiff(out, ref.bfloat16()) < 0.05


# ---------------------------------------------------------------- tests/benchmark.py extended configs (:283-337)
EXT_FWD = [(1, 4, 4, n, 64, 4, 64, torch.float16) for n in (64, 128, 256, 512, 1024)] + [
    (1, 4, 4, 256, 64, 0, 64, torch.float16),
    (1, 4, 4, 256, 64, 4, 1, torch.float16),
    (1, 4, 4, 256, 64, 1, 32, torch.float16),
    (1, 4, 4, 256, 64, 32, 64, torch.float16),
    (1, 4, 4, 256, 128, 4, 64, torch.float16),
    (1, 8, 1, 256, 64, 4, 64, torch.float16),
    (1, 8, 2, 256, 64, 4, 64, torch.float16),
    (1, 32, 8, 256, 64, 4, 64, torch.float16),
    (4, 4, 4, 128, 64, 4, 32, torch.float16),
    (1, 4, 4, 256, 64, 4, 64, torch.bfloat16),
]


@pytest.mark.parametrize("B,Hq,Hkv,N,D,ns,W,dtype", EXT_FWD)
def test_extended_forward(B, Hq, Hkv, N, D, ns, W, dtype):
    q, k, v, _ = make_qkv(B, Hq, Hkv, N, D, dtype)
    out = _op()(q.to(DEV), k.to(DEV), v.to(DEV), num_sink=ns, window_size=W)
    ref, _ = oracle_fwd(q, k, v, ns, W)
    tol = 2e-2 if dtype == torch.float16 else 5e-3      # benchmark.py:63-64
    assert_close(out, ref.to(dtype), tol, tol, "ext fwd")


EXT_BWD = [(1, 4, 4, n, 64, 4, 32) for n in (64, 128, 256)] + [
    (1, 8, 2, 128, 64, 4, 32), (1, 4, 4, 128, 128, 4, 32), (1, 4, 4, 128, 64, 0, 64), (1, 4, 4, 128, 64, 4, 1)]


@pytest.mark.parametrize("B,Hq,Hkv,N,D,ns,W", EXT_BWD)
def test_extended_backward(B, Hq, Hkv, N, D, ns, W, dkdv):
    q, k, v, g = make_qkv(B, Hq, Hkv, N, D, torch.float32)
    do = rand((B, Hq, N, D), g, torch.float32)
    dq_r, dk_r, dv_r, _ = oracle_bwd(q, k, v, do, ns, W)
    qt, kt, vt = (t.half().to(DEV).requires_grad_(True) for t in (q, k, v))
    _op()(qt, kt, vt, num_sink=ns, window_size=W).backward(do.half().to(DEV))
    assert dkdv_kernel_name(dkdv, B, Hkv, N, N, D, W, ns=ns) in _path(), _path()
    assert_close(qt.grad.float(), dq_r, 5e-2, 5e-2, "dq")
    assert_close(kt.grad.float(), dk_r, 5e-2, 5e-2, "dk")
    assert_close(vt.grad.float(), dv_r, 5e-2, 5e-2, "dv")


@pytest.mark.parametrize("N", [2048, 4096, 8192, 16384])
def test_forward_long_seq(N):                       # benchmark.py:91-104 -- here WITH an oracle (banded)
    B, Hq, Hkv, D = 1, 8, 2, 128
    q, k, v, _ = make_qkv(B, Hq, Hkv, N, D, torch.float16)
    out = _op()(q.to(DEV), k.to(DEV), v.to(DEV), num_sink=4, window_size=4096)
    assert out.shape == (B, Hq, N, D) and torch.isfinite(out).all()
    if N <= 4096:
        ref, _ = oracle_fwd(q, k, v, 4, 4096)
        assert_close(out, ref.half(), 2e-2, 2e-2, "long fwd")


@pytest.mark.parametrize("N", [2048, 4096, 8192])
def test_backward_long_seq(N, dkdv):                      # benchmark.py:107-119
    q, k, v, _ = make_qkv(1, 4, 4, N, 64, torch.float16)
    q, k, v = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    _op()(q, k, v, num_sink=4, window_size=4096).sum().backward()
    for t in (q, k, v):
        assert torch.isfinite(t.grad).all()


# ---------------------------------------------------------------- golden vectors captured from the reference
GOLD = [n for n in G.names("f") if n[:2] in ("f1", "f2", "f3", "f4", "f5")]


@pytest.mark.parametrize("force_generic", [False, True])
@pytest.mark.parametrize("name", GOLD)
def test_golden_fwd_bwd(name, force_generic, dkdv):
    g = G.load(name)
    m = G.fwd_bwd_meta(g)
    if "fp16" in name:
        dt, tol_o, tol_g = torch.float16, 4e-3, 2e-2
    elif "bf16" in name or name.startswith("f4"):
        dt, tol_o, tol_g = torch.bfloat16, 2e-2, 8e-2
    else:
        dt, tol_o, tol_g = torch.float32, 2e-5, 1e-4
    q, k, v = (g[x].to(dt).to(DEV).requires_grad_(True) for x in ("q", "k", "v"))
    sa = g["s_aux"].to(DEV).requires_grad_(True) if "s_aux" in g else None
    out = _ex()(q, k, v, m["ns"], m["W"], s_aux=sa, force_generic=force_generic)
    out.backward(g["do"].to(dt).to(DEV))
    tag = "eager" if "o_eager" in g else "kernel"
    assert maxdiff(out, g["o_" + tag]) < tol_o, "o"
    assert maxdiff(q.grad, g["dq_" + tag]) < tol_g, "dq"
    assert maxdiff(k.grad, g["dk_" + tag]) < tol_g, "dk"
    assert maxdiff(v.grad, g["dv_" + tag]) < tol_g, "dv"
    if sa is not None:
        assert maxdiff(sa.grad, g["ds_aux_" + tag]) < tol_g * 10, "ds_aux"


# ---------------------------------------------------------------- shapes / layouts the reference's kernel cannot take
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("B,Hq,Hkv,N,D,ns,W,aux", [
    (2, 4, 2, 77, 128, 4, 7, True),       # ragged N
    (1, 8, 1, 333, 64, 3, 50, False),     # MQA, ragged
    (1, 8, 2, 200, 80, 0, 16, True),      # D = 80
    (1, 2, 2, 130, 32, 2, 1000, True),    # W >= N
    (1, 2, 1, 65, 16, 100, 5, False),     # num_sink >= N
    (1, 4, 4, 96, 256, 4, 20, True),      # D = 256
    (1, 8, 2, 700, 128, 4, 128, True),    # several KV tiles, window edges inside tiles
    (1, 4, 4, 513, 64, 70, 100, False),   # sink range spanning more than one tile
])
def test_shapes_fwd_bwd(B, Hq, Hkv, N, D, ns, W, aux, dtype, dkdv):
    q, k, v, g = make_qkv(B, Hq, Hkv, N, D, dtype)
    do = rand((B, Hq, N, D), g, dtype)
    sa = rand((Hq,), g, torch.float32, 0.5) if aux else None
    o_r, _ = oracle_fwd(q, k, v, ns, W, sa)
    dq_r, dk_r, dv_r, dsa_r = oracle_bwd(q, k, v, do, ns, W, sa)
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    sad = sa.to(DEV).requires_grad_(True) if aux else None
    out = _op()(qd, kd, vd, num_sink=ns, window_size=W, s_aux=sad)
    out.backward(do.to(DEV))
    want = dkdv_kernel_name(dkdv, B, Hkv, N, N, D, W, dtype=dtype, ns=ns)
    assert want is None or want in _path(), (want, _path())
    to, tg = {torch.float32: (2e-5, 2e-4), torch.float16: (4e-3, 3e-2), torch.bfloat16: (2e-2, 1.5e-1)}[dtype]
    assert maxdiff(out, o_r) < to, f"o ({_path()})"
    assert maxdiff(qd.grad, dq_r) < tg, "dq"
    assert maxdiff(kd.grad, dk_r) < tg, "dk"
    assert maxdiff(vd.grad, dv_r) < tg, "dv"
    if aux:
        assert maxdiff(sad.grad, dsa_r) < tg * 10, "ds_aux"


def test_strided_bnhd_inputs_and_output():
    """[B,N,H,D] activations passed as transposed views; output produced in [B,N,H,D] memory."""
    B, N, Hq, Hkv, D = 2, 160, 4, 2, 128
    g = torch.Generator().manual_seed(3)
    qs, ks, vs = rand((B, N, Hq, D), g, torch.bfloat16), rand((B, N, Hkv, D), g, torch.bfloat16), rand(
        (B, N, Hkv, D), g, torch.bfloat16)
    out = _ex()(qs.to(DEV).transpose(1, 2), ks.to(DEV).transpose(1, 2), vs.to(DEV).transpose(1, 2), 2, 33,
                out_bnhd=True)
    assert out.transpose(1, 2).is_contiguous()
    ref, _ = oracle_fwd(qs.transpose(1, 2), ks.transpose(1, 2), vs.transpose(1, 2), 2, 33)
    assert maxdiff(out, ref) < 2e-2


def test_deterministic_bitwise(dkdv):
    q, k, v, g = make_qkv(2, 8, 2, 384, 128, torch.bfloat16)
    do = rand((2, 8, 384, 128), g, torch.bfloat16).to(DEV)
    sa = rand((8,), g, torch.float32).to(DEV)
    res = []
    for _ in range(2):
        qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
        sad = sa.clone().requires_grad_(True)
        out = _op()(qd, kd, vd, num_sink=4, window_size=100, s_aux=sad)
        out.backward(do)
        res.append((out.detach(), qd.grad, kd.grad, vd.grad, sad.grad))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_mfma_path_is_used_for_headline_shapes():
    """bf16/fp16 with D in {64, 128} must run the MFMA kernels, not the generic VALU path."""
    for dt in (torch.bfloat16, torch.float16):
        for D in (64, 128):
            q, k, v, _ = make_qkv(1, 2, 1, 128, D, dt)
            qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
            out = _op()(qd, kd, vd, num_sink=4, window_size=32)
            assert "mfma" in _path(), _path()
            out.sum().backward()
            assert "mfma" in _path(), _path()


@pytest.mark.parametrize("B,Hkv", [(2, 4), (4, 4), (2, 8), (3, 8)])
def test_backward_with_many_kv_groups(B, Hkv, dkdv):
    """B * H_kv = 8, 16 (and 24: three groups per XCD share): the dK/dV kernels order their workgroups per XCD (heavy
    sink blocks first) with index arithmetic that only engages when the (batch, KV head) count is a multiple of 8 -
    the branch the C3 bench runs (B * H_kv = 32).  Several key blocks, sinks, a window shorter than N, GQA."""
    Hq, N, D, ns, W = 2 * Hkv, 600, 128, 4, 200
    q, k, v, g = make_qkv(B, Hq, Hkv, N, D, torch.bfloat16, seed=7 + B * Hkv)
    do = rand((B, Hq, N, D), g, torch.bfloat16)
    sa = rand((Hq,), g, torch.float32, 0.5)
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    sad = sa.to(DEV).requires_grad_(True)
    out = _op()(qd, kd, vd, num_sink=ns, window_size=W, s_aux=sad)
    out.backward(do.to(DEV))
    assert dkdv_kernel_name(dkdv, B, Hkv, N, N, D, W, ns=ns) in _path(), _path()      # (dense batch: the hand-placed kernel + row split either way)
    o_r, _ = oracle_fwd(q, k, v, ns, W, sa)
    dq_r, dk_r, dv_r, dsa_r = oracle_bwd(q, k, v, do, ns, W, sa)
    assert_close(out, o_r.bfloat16(), 2e-2, 2e-2, "fwd")
    assert_close(qd.grad, dq_r, 5e-2, 5e-2, "dq")
    assert_close(kd.grad, dk_r, 5e-2 * max(1.0, dk_r.abs().max().item()), 5e-2, "dk")
    assert_close(vd.grad, dv_r, 5e-2 * max(1.0, dv_r.abs().max().item()), 5e-2, "dv")
    assert maxdiff(sad.grad, dsa_r) < 5e-2 * max(1.0, dsa_r.abs().max().item())


@pytest.mark.parametrize("D,dtype", [(64, torch.bfloat16), (80, torch.bfloat16), (96, torch.float16), (64, torch.float16)])
def test_hand_placed_kernels_other_head_dims(D, dtype, dkdv):
    """head dims 64 / 80 / 96 with a window long enough for the hand-placed kernels (the compiled ones keep the short
    windows): GQA, sinks, s_aux, ragged N, strided BNHD inputs, against the oracle"""
    B, Hq, Hkv, N, ns, W = 2, 8, 2, 777, 4, 400
    g = torch.Generator().manual_seed(D)
    qs, ks, vs = rand((B, N, Hq, D), g, dtype), rand((B, N, Hkv, D), g, dtype), rand((B, N, Hkv, D), g, dtype)
    q, k, v = qs.transpose(1, 2), ks.transpose(1, 2), vs.transpose(1, 2)
    do = rand((B, Hq, N, D), g, dtype)
    sa = rand((Hq,), g, torch.float32, 0.5)
    qd, kd, vd = (t.to(DEV).transpose(1, 2).requires_grad_(True) for t in (qs, ks, vs))
    sad = sa.to(DEV).requires_grad_(True)
    out = _op()(qd, kd, vd, num_sink=ns, window_size=W, s_aux=sad)
    assert "asm4x64" in _path(), _path()
    out.backward(do.to(DEV))
    assert "dkdvasm4x64" in _path() and "dqasm4x64" in _path(), _path()
    o_r, _ = oracle_fwd(q, k, v, ns, W, sa)
    dq_r, dk_r, dv_r, dsa_r = oracle_bwd(q, k, v, do, ns, W, sa)
    to = 2e-2 if dtype == torch.bfloat16 else 1e-2
    assert_close(out, o_r, to, to, "fwd")
    assert_close(qd.grad, dq_r, 5e-2, 5e-2, "dq")
    assert_close(kd.grad, dk_r, 5e-2 * max(1.0, dk_r.abs().max().item()), 5e-2, "dk")
    assert_close(vd.grad, dv_r, 5e-2 * max(1.0, dv_r.abs().max().item()), 5e-2, "dv")
    assert maxdiff(sad.grad, dsa_r) < 5e-2 * max(1.0, dsa_r.abs().max().item())


@pytest.mark.parametrize("D,dtype,W", [(256, torch.bfloat16, 300), (256, torch.float16, 0), (32, torch.bfloat16, 200),
                                       (32, torch.float16, 50)])
def test_mfma_kernels_head_dims_32_and_256(D, dtype, W):
    """head dims the reference lists beside 64 / 128 (README "Head dims: 64, 128, 256") and a small one: forward and
    backward run MFMA kernels for both (256: split-column dK/dV kernel); GQA, sinks, s_aux, ragged N, against the oracle"""
    B, Hq, Hkv, N, ns = 2, 4, 2, 333, 4
    W = W or N
    g = torch.Generator().manual_seed(D + 1)
    q, k, v = rand((B, Hq, N, D), g, dtype), rand((B, Hkv, N, D), g, dtype), rand((B, Hkv, N, D), g, dtype)
    do = rand((B, Hq, N, D), g, dtype)
    sa = rand((Hq,), g, torch.float32, 0.5)
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    sad = sa.to(DEV).requires_grad_(True)
    out = _op()(qd, kd, vd, num_sink=ns, window_size=W, s_aux=sad)
    assert "fwd_mfma" in _path() and "d%d" % D in _path(), _path()
    out.backward(do.to(DEV))
    assert "bwd_mfma" in _path() and "d%d" % D in _path(), _path()
    o_r, _ = oracle_fwd(q, k, v, ns, W, sa)
    dq_r, dk_r, dv_r, dsa_r = oracle_bwd(q, k, v, do, ns, W, sa)
    to = 2e-2 if dtype == torch.bfloat16 else 1e-2
    assert_close(out, o_r, to, to, "fwd")
    assert_close(qd.grad, dq_r, 5e-2, 5e-2, "dq")
    assert_close(kd.grad, dk_r, 5e-2 * max(1.0, dk_r.abs().max().item()), 5e-2, "dk")
    assert_close(vd.grad, dv_r, 5e-2 * max(1.0, dv_r.abs().max().item()), 5e-2, "dv")
    assert maxdiff(sad.grad, dsa_r) < 5e-2 * max(1.0, dsa_r.abs().max().item())


def test_head_dim_256_backward_larger_shapes():
    """head dim 256 backward on shapes with several key blocks, a window shorter than the sequence, MHA and GQA, and
    N_q < N_kv (queries are the last rows), against the banded oracle"""
    for (B, Hq, Hkv, N, Nk, ns, W) in ((1, 4, 4, 700, 700, 4, 200), (2, 4, 1, 515, 515, 0, 515), (1, 2, 1, 200, 456, 3, 128)):
        g = torch.Generator().manual_seed(N)
        dt = torch.bfloat16
        q, do = rand((B, Hq, N, 256), g, dt), rand((B, Hq, N, 256), g, dt)
        k, v = rand((B, Hkv, Nk, 256), g, dt), rand((B, Hkv, Nk, 256), g, dt)
        qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
        out = _op()(qd, kd, vd, num_sink=ns, window_size=W)
        out.backward(do.to(DEV))
        assert "bwd_mfma" in _path() and "d256" in _path(), _path()
        o_r, _ = oracle_fwd(q, k, v, ns, W)
        dq_r, dk_r, dv_r, _ = oracle_bwd(q, k, v, do, ns, W)
        assert_close(out, o_r, 2e-2, 2e-2, "fwd")
        assert_close(qd.grad, dq_r, 5e-2, 5e-2, "dq")
        assert_close(kd.grad, dk_r, 5e-2 * max(1.0, dk_r.abs().max().item()), 5e-2, "dk")
        assert_close(vd.grad, dv_r, 5e-2 * max(1.0, dv_r.abs().max().item()), 5e-2, "dv")


@pytest.mark.parametrize("B,Hq,Hkv,N,Nk,D,ns,W,dtype", [
    (1, 4, 1, 2100, 2100, 128, 4, 128, torch.bfloat16),     # 7 chunks of block 0's sweep, GQA, ragged last chunk
    (2, 2, 2, 1500, 1500, 64, 70, 300, torch.float16),      # MHA, head dim 64, sinks beyond one 64-key wave
    (1, 2, 1, 1000, 1300, 128, 4, 64, torch.bfloat16),      # N_q < N_kv
    (1, 2, 1, 3000, 3000, 96, 300, 512, torch.bfloat16),    # more sink keys than a key block holds
    (1, 4, 1, 2100, 2100, 64, 4, 100, torch.bfloat16),      # short window below head dim 128: the compiled kernel's split
    (2, 2, 1, 900, 1200, 80, 130, 64, torch.float16),       # the same with N_q < N_kv and sinks beyond a 128-key block
    (1, 2, 1, 20000, 20000, 128, 4, 32, torch.bfloat16),    # more chunks than the cap of 64 (banded oracle)
])
def test_sink_split_of_the_dkdv_sweep(B, Hq, Hkv, N, Nk, D, ns, W, dtype, dkdv):
    """key block 0 (the sink keys see every row) is swept by several workgroups whose partial dK / dV are added up:
    against the oracle, and bitwise deterministic across runs"""
    from sink_attention import _native
    g = torch.Generator().manual_seed(N + D)
    q, do = rand((B, Hq, N, D), g, dtype), rand((B, Hq, N, D), g, dtype)
    k, v = rand((B, Hkv, Nk, D), g, dtype), rand((B, Hkv, Nk, D), g, dtype)
    sa = rand((Hq,), g, torch.float32, 0.5)
    grads = []
    for _ in range(2):
        qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
        sad = sa.to(DEV).requires_grad_(True)
        out = _op()(qd, kd, vd, num_sink=ns, window_size=W, s_aux=sad)
        out.backward(do.to(DEV))
        assert dkdv_kernel_name(dkdv, B, Hkv, N, Nk, D, W, ns=ns) in _path(), _path()
        grads.append((qd.grad.clone(), kd.grad.clone(), vd.grad.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*grads))
    dq_r, dk_r, dv_r, _ = oracle_bwd(q, k, v, do, ns, W, sa)
    assert_close(grads[0][0], dq_r, 5e-2, 5e-2, "dq")
    assert_close(grads[0][1], dk_r, 5e-2 * max(1.0, dk_r.abs().max().item()), 5e-2, "dk")
    assert_close(grads[0][2], dv_r, 5e-2 * max(1.0, dv_r.abs().max().item()), 5e-2, "dv")


def test_small_grids_row_split_or_compiled_dkdv_kernel():
    """a grid of 256-key blocks smaller than the chip: a dense batch (N_q = N_kv) keeps the hand-placed dK/dV kernel and
    cuts EVERY block's sweep into chunks (row split, partial dK / dV added up in a fixed order); with N_q < N_kv the
    workspace query cannot size the partials and the compiled kernel (128-key blocks: twice the workgroups) takes over.
    Both against the oracle, with the library's own rule (the other GPU tests force the hand-placed kernel)."""
    from sink_attention import set_backward_options
    prev = set_backward_options(dkdv="rule")
    try:
        for (B, Hq, Hkv, N, Nk, D, ns, W, want) in ((1, 4, 1, 2048, 2048, 128, 4, 512, "dkdvasm4x64"),      # one KV head: 8 blocks
                                                    (1, 4, 1, 1500, 1500, 64, 300, 700, "dkdvasm4x64"),     # sinks over two blocks
                                                    (2, 4, 2, 1000, 1000, 128, 0, 1000, "dkdvasm4x64"),     # causal, no sinks
                                                    (1, 4, 1, 257, 257, 128, 130, 0, "dkdvasm4x64"),        # window 0: a split block NO row sees
                                                    (1, 8, 2, 1024, 1200, 128, 4, 4096, "dkdvws8"),         # N_q < N_kv
                                                    (2, 4, 1, 300, 1500, 128, 4, 700, "dkdvws8"),           # N_q << N_kv, window edge inside
                                                    (1, 8, 8, 512, 700, 96, 0, 400, "dkdvws8"),             # MHA, no sinks, head dim 96
                                                    (1, 4, 2, 77, 2000, 64, 130, 900, "dkdvws8")):          # ragged rows, sinks over three 64-key tiles
            q, do = rand((B, Hq, N, D), torch.Generator().manual_seed(N), torch.bfloat16), None
            g = torch.Generator().manual_seed(N + 1)
            do = rand((B, Hq, N, D), g, torch.bfloat16)
            k, v = rand((B, Hkv, Nk, D), g, torch.bfloat16), rand((B, Hkv, Nk, D), g, torch.bfloat16)
            grads = []
            for _ in range(2):
                qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
                _op()(qd, kd, vd, num_sink=ns, window_size=W).backward(do.to(DEV))
                assert want in _path(), _path()
                grads.append((qd.grad.clone(), kd.grad.clone(), vd.grad.clone()))
            assert all(torch.equal(a, b) for a, b in zip(*grads))          # deterministic
            dq_r, dk_r, dv_r, _ = oracle_bwd(q, k, v, do, ns, W)
            assert_close(grads[0][0], dq_r, 5e-2, 5e-2, "dq")
            assert_close(grads[0][1], dk_r, 5e-2 * max(1.0, dk_r.abs().max().item()), 5e-2, "dk")
            assert_close(grads[0][2], dv_r, 5e-2 * max(1.0, dv_r.abs().max().item()), 5e-2, "dv")
    finally:
        set_backward_options(overlap=prev[0], dkdv=prev[1] or "rule")


def test_baseline_c4_full_shape(dkdv):
    """BASELINE.json configs[3] at its REAL shape: gpt-oss-20b sliding layer, bf16, H_q=64, H_kv=8, D=80, N=8192,
    window=128, s_aux, fwd+bwd including ds_aux, against the banded oracle (cheap at W=128)."""
    B, Hq, Hkv, N, D, ns, W = 1, 64, 8, 8192, 80, 0, 128
    q, k, v, g = make_qkv(B, Hq, Hkv, N, D, torch.bfloat16, seed=20)
    do = rand((B, Hq, N, D), g, torch.bfloat16)
    sa = rand((Hq,), g, torch.float32, 0.5)
    o_r, _ = oracle_fwd(q, k, v, ns, W, sa, banded=True)
    dq_r, dk_r, dv_r, dsa_r = oracle_bwd(q, k, v, do, ns, W, sa, banded=True)
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    sad = sa.to(DEV).requires_grad_(True)
    out = _op()(qd, kd, vd, num_sink=ns, window_size=W, s_aux=sad)
    out.backward(do.to(DEV))
    # D = 80, W = 128, no sink keys: the skewed-sweep kernel under the rule; forced: the plain hand-placed kernel would need a grid
    # that fills the chip twice (256 blocks here), so the compiled one
    assert dkdv_kernel_name(dkdv, B, Hkv, N, N, D, W, ns=ns) in _path(), _path()
    assert ("dkdvasmskew" in _path()) == (dkdv == "rule"), _path()
    assert_close(out, o_r.bfloat16(), 2e-2, 2e-2, "C4 fwd")
    assert_close(qd.grad, dq_r, 5e-2, 5e-2, "C4 dq")
    assert_close(kd.grad, dk_r, 5e-2 * max(1.0, dk_r.abs().max().item()), 5e-2, "C4 dk")
    assert_close(vd.grad, dv_r, 5e-2 * max(1.0, dv_r.abs().max().item()), 5e-2, "C4 dv")
    assert maxdiff(sad.grad, dsa_r) < 5e-2 * max(1.0, dsa_r.abs().max().item())


@pytest.mark.parametrize("cfg", ["C2", "C3slice", "C4slice"])
def test_baseline_config_shapes_against_banded_oracle(cfg, dkdv):
    """BASELINE.json configs at full N on a slice of (batch, heads) small enough for the CPU oracle."""
    if cfg == "C2":      # fwd MHA bf16 N=4096 D=128 ns=4 W=1024
        B, Hq, Hkv, N, D, ns, W, aux = 1, 2, 2, 4096, 128, 4, 1024, False
    elif cfg == "C3slice":  # fwd+bwd GQA bf16 N=8192 D=128 ns=4 W=4096, one KV group
        B, Hq, Hkv, N, D, ns, W, aux = 1, 4, 1, 8192, 128, 4, 4096, False
    else:                # gpt-oss shape: D=80, W=128, s_aux, one KV group of 8 q heads, N=2048
        B, Hq, Hkv, N, D, ns, W, aux = 1, 8, 1, 2048, 80, 0, 128, True
    q, k, v, g = make_qkv(B, Hq, Hkv, N, D, torch.bfloat16)
    do = rand((B, Hq, N, D), g, torch.bfloat16)
    sa = rand((Hq,), g, torch.float32, 0.5) if aux else None
    o_r, _ = oracle_fwd(q, k, v, ns, W, sa, banded=True)
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    sad = sa.to(DEV).requires_grad_(True) if aux else None
    out = _op()(qd, kd, vd, num_sink=ns, window_size=W, s_aux=sad)
    assert_close(out, o_r.bfloat16(), 2e-2, 2e-2, cfg + " fwd")
    if cfg != "C2":
        dq_r, dk_r, dv_r, dsa_r = oracle_bwd(q, k, v, do, ns, W, sa, banded=True)
        out.backward(do.to(DEV))
        assert_close(qd.grad, dq_r, 5e-2, 5e-2, cfg + " dq")
        assert_close(kd.grad, dk_r, 5e-2 * max(1.0, dk_r.abs().max().item()), 5e-2, cfg + " dk")
        assert_close(vd.grad, dv_r, 5e-2 * max(1.0, dv_r.abs().max().item()), 5e-2, cfg + " dv")
        if aux:
            assert maxdiff(sad.grad, dsa_r) < 5e-2 * max(1.0, dsa_r.abs().max().item())


@pytest.mark.parametrize("B,Hq,Hkv,N,D,W,dtype,layout", [
    (1, 8, 1, 777, 80, 128, torch.bfloat16, "bhnd"),      # the gpt-oss group of 8, ragged last block / last slice
    (2, 2, 2, 1000, 64, 33, torch.float16, "bnhd"),       # MHA, short window (T rounded up to 6), [B, N, H, D] views
    (1, 6, 2, 530, 96, 200, torch.bfloat16, "bhnd"),      # T = 9 > 6 (trips without a B slice), group of 3
    (3, 4, 1, 256, 64, 512, torch.bfloat16, "bnhd"),      # window longer than the sequence, one block
    (1, 2, 1, 40, 80, 1, torch.float16, "bhnd"),          # window of one key, a sequence shorter than a wave's keys
    (2, 16, 2, 2048, 64, 128, torch.bfloat16, "bnhd")])   # many blocks per group
def test_short_window_skewed_dkdv_sweep(B, Hq, Hkv, N, D, W, dtype, layout):
    """No sink keys + a short window at head dims 64 / 80 / 96: bwd_dkdv_skew_asm_kernel (tools/asmgen/dkdv_skew.py) under
    the rule, against the oracle and against the plain hand-placed / compiled kernels of the same library (per-call flags);
    bitwise deterministic."""
    from sink_attention import set_backward_options
    q, k, v, g = make_qkv(B, Hq, Hkv, N, D, dtype, seed=N + W)
    do = rand((B, Hq, N, D), g, dtype)
    sa = rand((Hq,), g, torch.float32, 0.5)
    dq_r, dk_r, dv_r, _ = oracle_bwd(q, k, v, do, 0, W, sa, banded=N > 600)
    res = {}
    prev = set_backward_options(dkdv="rule")
    try:
        for mode in ("rule", "rule", "ws"):
            set_backward_options(dkdv=mode)
            if layout == "bnhd":
                qd, kd, vd = (t.transpose(1, 2).contiguous().to(DEV).transpose(1, 2).requires_grad_(True) for t in (q, k, v))
                dod = do.transpose(1, 2).contiguous().to(DEV).transpose(1, 2)
            else:
                qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
                dod = do.to(DEV)
            sad = sa.to(DEV).requires_grad_(True)
            _op()(qd, kd, vd, num_sink=0, window_size=W, s_aux=sad).backward(dod)
            assert ("dkdvasmskew" in _path()) == (mode == "rule"), _path()
            res.setdefault(mode, []).append((qd.grad.clone(), kd.grad.clone(), vd.grad.clone()))
    finally:
        set_backward_options(overlap=prev[0], dkdv=prev[1] or "rule")
    a, b = res["rule"]
    assert all(torch.equal(x, y) for x, y in zip(a, b))                      # deterministic
    tol = 5e-2
    assert_close(a[0], dq_r, tol, tol, "dq")
    assert_close(a[1], dk_r, tol * max(1.0, dk_r.abs().max().item()), tol, "dk")
    assert_close(a[2], dv_r, tol * max(1.0, dv_r.abs().max().item()), tol, "dv")
    # ... and no further from the oracle than the compiled kernel of the same call
    c = res["ws"][0]
    assert maxdiff(a[1], dk_r) <= 1.5 * maxdiff(c[1], dk_r) + 1e-3 and maxdiff(a[2], dv_r) <= 1.5 * maxdiff(c[2], dv_r) + 1e-3


def test_very_long_sequence_and_chunked_tail():
    """N = 65536 (offsets far beyond 16 bits, 1024 key tiles): forward against the banded oracle, forward+backward
    finite, and the last 8192 queries run alone against all keys (N_q < N_kv) reproduce the tail of the full run."""
    from sink_attention import _native, sink_flash_attention
    from sink_attention.sink_flash_attention import _sink_flash_attention_ex
    g = torch.Generator().manual_seed(123)
    B, Hq, Hkv, N, D, ns, W = 1, 4, 1, 65536, 128, 4, 4096
    q, k, v = rand((B, Hq, N, D), g, torch.bfloat16), rand((B, Hkv, N, D), g, torch.bfloat16), rand(
        (B, Hkv, N, D), g, torch.bfloat16)
    sa = rand((Hq,), g, torch.float32, 0.5)
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    sad = sa.to(DEV).requires_grad_(True)
    out = sink_flash_attention(qd, kd, vd, ns, W, sad)
    assert "mfma" in _native.last_path()
    out.float().sum().backward()
    for t in (out, qd.grad, kd.grad, vd.grad, sad.grad):
        assert torch.isfinite(t).all()
    # oracle on two windows of rows (head 1): fp64 masked softmax over just the keys those rows can see
    from oracle import sink_oracle as O
    for r0 in (0, N - 512):
        rows = torch.arange(r0, r0 + 512)
        cols = torch.cat([torch.arange(0, min(ns, r0 + 512)), torch.arange(max(r0 - W + 1, ns, 0), r0 + 512)])
        s = (q[0, 1, rows].double() @ k[0, 0, cols].double().T) / D ** 0.5
        s = s.masked_fill(~O.valid_mask(rows, cols, ns, W), float("-inf"))
        lse = torch.logsumexp(torch.cat([s, sa[1].double().expand(512, 1)], dim=1), dim=1, keepdim=True)
        o_r = torch.exp(s - lse) @ v[0, 0, cols].double()
        assert maxdiff(out[0, 1, r0:r0 + 512], o_r) < 2e-2
    tail = _sink_flash_attention_ex(qd.detach()[:, :, N - 8192:], kd.detach(), vd.detach(), ns, W, s_aux=sad.detach())
    assert torch.equal(tail, out.detach()[:, :, N - 8192:])


def test_row_split_partials_are_f32_one_rounding():
    """Small grids: every key block's sweep is cut into chunks whose partial dK / dV are f32 and rounded once
    (bwd_part_reduce_kernel), so the split result carries no more rounding than the unsplit kernel's.  B = 1, H = 4 / 1,
    N = 8192, D = 128 (a tensor-parallel shard: 32 key blocks on 256 CUs, row split active) against the SAME problem as
    batch 0 of a 16-fold batch (512 workgroups: no row split): max |dK - oracle| and max |dV - oracle| of the split run
    within 1.25 x the unsplit run's."""
    Hq, Hkv, N, D, ns, W = 4, 1, 8192, 128, 4, 4096
    q, k, v, g = make_qkv(1, Hq, Hkv, N, D, torch.bfloat16, seed=23)
    do = rand((1, Hq, N, D), g, torch.bfloat16)
    dq_r, dk_r, dv_r, _ = oracle_bwd(q, k, v, do, ns, W, banded=True)
    res = {}
    for B in (1, 16):
        qd, kd, vd = (t.expand(B, -1, -1, -1).contiguous().to(DEV).requires_grad_(True) for t in (q, k, v))
        _op()(qd, kd, vd, num_sink=ns, window_size=W).backward(do.expand(B, -1, -1, -1).contiguous().to(DEV))
        assert ("dkdvasm4x64rs" in _path()) == (B == 1), _path()
        res[B] = (maxdiff(kd.grad[:1], dk_r), maxdiff(vd.grad[:1], dv_r), maxdiff(qd.grad[:1], dq_r))
    assert res[1][0] <= 1.25 * res[16][0] and res[1][1] <= 1.25 * res[16][1], res
    assert res[1][2] == res[16][2], res                     # dQ: the same kernel either way
    assert res[1][0] < 5e-2 * max(1.0, dk_r.abs().max().item()) and res[1][1] < 5e-2 * max(1.0, dv_r.abs().max().item()), res
