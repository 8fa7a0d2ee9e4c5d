"""Shared helpers for the GPU parity tests."""
import torch

from oracle import sink_oracle as O


def rand(shape, seed_gen, dtype, scale=1.0):
    """CPU-generator randn (reproducible everywhere), rounded to ``dtype``."""
    return (torch.randn(*shape, generator=seed_gen, dtype=torch.float32) * scale).to(dtype)


def make_qkv(B, Hq, Hkv, N, D, dtype, seed=42):
    g = torch.Generator().manual_seed(seed)
    q = rand((B, Hq, N, D), g, dtype)
    k = rand((B, Hkv, N, D), g, dtype)
    v = rand((B, Hkv, N, D), g, dtype)
    return q, k, v, g


def oracle_fwd(q, k, v, ns, W, s_aux=None, banded=None):
    """fp64 oracle on the exact low-precision inputs (upcast)."""
    N = q.shape[2]
    if banded is None:
        banded = N > 1024
    fn = O.sink_attention_banded if banded else O.sink_attention_dense
    o, lse = fn(q.cpu(), k.cpu(), v.cpu(), ns, W, None if s_aux is None else s_aux.cpu().float())
    return o, lse


def oracle_bwd(q, k, v, do, ns, W, s_aux=None, banded=None):
    N = q.shape[2]
    if banded is None:
        banded = N > 1024
    fn = O.sink_attention_bwd_banded if banded else O.sink_attention_bwd_dense
    return fn(q.cpu(), k.cpu(), v.cpu(), do.cpu(), ns, W, None if s_aux is None else s_aux.cpu().float())


def maxdiff(a, b):
    if a.numel() == 0:
        return 0.0
    return (a.detach().double().cpu() - b.detach().double().cpu()).abs().max().item()


def assert_close(actual, expected, atol, rtol, what=""):
    a = actual.detach().double().cpu()
    e = expected.detach().double().cpu()
    err = (a - e).abs()
    tol = atol + rtol * e.abs()
    bad = err > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance "
                           f"(atol={atol}, rtol={rtol}); max abs err {err.max().item():.3e}")
