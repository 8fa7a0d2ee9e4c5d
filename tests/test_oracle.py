"""Pins oracle/sink_oracle.py to the reference: every golden vector produced by
running the reference (its eager oracles AND its Triton kernels under the
interpreter; tests/golden/make_golden.py) must be reproduced by the CPU
restatement.  CPU only."""
import pytest
import torch

import golden_util as G
from oracle import sink_oracle as O

FWD_BWD = [n for n in G.names("f") if n[:2] in ("f1", "f2", "f3", "f4", "f5")]
DECODE = G.names("f6")


def _maxdiff(a, b):
    if a.numel() == 0:
        return 0.0
    return (a.double() - b.double()).abs().max().item()


@pytest.mark.parametrize("name", FWD_BWD)
def test_oracle_matches_reference_fwd_bwd(name):
    g = G.load(name)
    m = G.fwd_bwd_meta(g)
    s_aux = g.get("s_aux")
    o, lse = O.sink_attention_dense(g["q"], g["k"], g["v"], m["ns"], m["W"], s_aux)
    dq, dk, dv, dsa = O.sink_attention_bwd_dense(g["q"], g["k"], g["v"], g["do"], m["ns"], m["W"], s_aux)
    lowp = "fp16" in name
    # eager oracle of the reference ran in fp32; its Triton kernel in fp32 or fp16
    tol_e = 2e-5
    tol_k = 2e-2 if lowp else 2e-5
    checked = 0
    for tag, tol in (("eager", tol_e), ("kernel", tol_k)):
        if "o_" + tag in g:
            assert _maxdiff(o, g["o_" + tag]) < tol, (name, tag, "o")
            assert _maxdiff(dq, g["dq_" + tag]) < tol * 5, (name, tag, "dq")
            assert _maxdiff(dk, g["dk_" + tag]) < tol * 5, (name, tag, "dk")
            assert _maxdiff(dv, g["dv_" + tag]) < tol * 5, (name, tag, "dv")
            if s_aux is not None:
                assert _maxdiff(dsa, g["ds_aux_" + tag]) < tol * 20, (name, tag, "ds_aux")
            checked += 1
    assert checked >= 1


@pytest.mark.parametrize("name", FWD_BWD)
def test_banded_equals_dense(name):
    g = G.load(name)
    m = G.fwd_bwd_meta(g)
    s_aux = g.get("s_aux")
    o, lse = O.sink_attention_dense(g["q"], g["k"], g["v"], m["ns"], m["W"], s_aux)
    ob, lb = O.sink_attention_banded(g["q"], g["k"], g["v"], m["ns"], m["W"], s_aux, block=32)
    assert _maxdiff(o, ob) < 1e-12
    fin = torch.isfinite(lse)
    assert torch.equal(fin, torch.isfinite(lb)) and _maxdiff(lse[fin], lb[fin]) < 1e-12
    a = O.sink_attention_bwd_dense(g["q"], g["k"], g["v"], g["do"], m["ns"], m["W"], s_aux)
    b = O.sink_attention_bwd_banded(g["q"], g["k"], g["v"], g["do"], m["ns"], m["W"], s_aux, block=32)
    for x, y in zip(a, b):
        if x is not None:
            assert _maxdiff(x, y) < 1e-11


@pytest.mark.parametrize("name", DECODE)
def test_oracle_matches_reference_decode(name):
    g = G.load(name)
    dt = int(g["meta"][6])
    o = O.decode_dense(g["q"], g["k"], g["v"], g.get("s_aux"))
    # reference outputs were rounded to the input dtype (fp32 / fp16 / bf16)
    tol = {0: 2e-6, 1: 2e-3, 2: 1.6e-2}[dt]
    assert _maxdiff(o, g["o_kernel"]) < tol
    assert _maxdiff(o, g["o_eager"]) < tol
    if name == "f6_dec_saux100":
        assert g["o_kernel"].abs().max().item() < 0.01 and o.abs().max().item() < 0.01


def test_autograd_agrees_with_explicit_backward():
    g = G.load("f3_mixed_ragged")
    m = G.fwd_bwd_meta(g)
    q, k, v, sa = (g[x].double().requires_grad_(True) for x in ("q", "k", "v", "s_aux"))
    o, _ = O.sink_attention_dense(q, k, v, m["ns"], m["W"], sa)
    gr = torch.autograd.grad(o, [q, k, v, sa], g["do"].double())
    ex = O.sink_attention_bwd_dense(g["q"], g["k"], g["v"], g["do"], m["ns"], m["W"], g["s_aux"])
    for a, b in zip(gr, ex):
        assert _maxdiff(a, b) < 1e-12


def test_known_answers():
    """Analytic KATs of the reference tests: window=N, ns=0 == plain causal softmax
    (tests/test_sink_attention.py:99-116); norm(out|s_aux=10) < norm(out|s_aux=0)
    (tests/test_s_aux.py:144-169)."""
    gen = torch.Generator().manual_seed(42)
    B, H, N, D = 1, 4, 128, 64
    q, k, v = (torch.randn(B, H, N, D, generator=gen) for _ in range(3))
    o, _ = O.sink_attention_dense(q, k, v, 0, N)
    s = (q.double() @ k.double().transpose(-2, -1)) / D ** 0.5
    s = s.masked_fill(torch.triu(torch.ones(N, N), 1).bool(), float("-inf"))
    assert _maxdiff(o, torch.softmax(s, -1) @ v.double()) < 1e-12
    o0, _ = O.sink_attention_dense(q, k, v, 0, N, torch.zeros(H))
    o10, _ = O.sink_attention_dense(q, k, v, 0, N, torch.full((H,), 10.0))
    assert o10.norm() < o0.norm()


def test_pair_count_matches_baseline_table():
    assert O.pair_count_closed(8192, 4, 4096) == 25184250        # C3
    assert O.pair_count_closed(4096, 4, 1024) == 3682810         # C2
    assert O.pair_count_closed(8192, 0, 128) == 1040448          # C4
    assert O.pair_count_closed(128, 4, 32) == 3978               # C1
    for N, ns, W in [(50, 7, 3), (20, 30, 5), (33, 4, 0), (17, 0, 100), (64, 4, 1)]:
        assert O.pair_count(N, ns, W) == O.pair_count_closed(N, ns, W)
