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


def dkdv_kernel_name(mode, B, Hkv, Nq, Nk, D, window, packed=False, dtype=torch.bfloat16, ns=1):
    """Name of the dK/dV kernel sfa_bwd must dispatch to, as it appears in sfa_last_path(): restates dkdv_asm() of
    csrc/sfa_bwd_mfma.hip.  mode: "rule" (the library's rule) / "asm" / "ws" (the per-call overrides).  None where the
    choice does not exist (fp32, head dims without a hand-placed body).  ns = num_sink of the call.  Packed batches: B = sequences, Nq = Nk = the
    longest one (the launch problem of sfa_bwd_varlen)."""
    if dtype == torch.float32 or D not in (64, 80, 96, 128):
        return None
    W = min(max(window, 0), Nk)
    # short windows without sink keys, self-attention, head dims below 128: the skewed sweep (dkdv_skew() of the library),
    # unless the call names one of the other kernels
    if mode == "rule" and D in (64, 80, 96) and ns <= 0 and (packed or Nq == Nk) and 1 <= W <= 512:
        return "dkdvasmskew"
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    wgs = -(-Nk // 256) * Hkv * B
    if not (D == 128 or W > 256 or wgs >= 2 * n_cu):   # short windows below head dim 128 on a grid that does not fill the chip twice: compiled kernels only
        return "dkdvws8"
    if mode == "asm":
        return "dkdvasm4x64"
    if mode == "ws":
        return "dkdvws8"
    dense = (not packed) and Nq == Nk      # row split available: the hand-placed kernel fills the chip whatever the grid
    fills = -(-Nk // 256) * Hkv * B >= n_cu
    return "dkdvasm4x64" if (dense or fills) else "dkdvws8"
