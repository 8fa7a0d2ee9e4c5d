"""CPU: the generated (hand-placed) dK/dV kernel body, run instruction by instruction in the gfx950 emulator
(tools/asmgen/emu.py), against the oracle.  The SAME instruction list is what csrc/gen/dkdv_asm.inc carries to the GPU
(tools/asmgen/emit.py); the emulator additionally checks every s_waitcnt (no register touched while a load into it is
outstanding) and the LDS-DMA / barrier protocol of the slice ring.  Also: the list assembles with the ROCm assembler,
and the scheduler's output computes bit for bit what the program order computes."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from asmgen.asmcheck import CLANG, assemble          # noqa: E402
from asmgen.dkdv import DkdvGen                       # noqa: E402
from asmgen.dq import DqGen                           # noqa: E402
from asmgen.fwd import FwdGen                         # noqa: E402
from asmgen.harness import run_dkdv, run_dq, run_fwd  # noqa: E402
from oracle import sink_oracle as O                   # noqa: E402

_PROGS = {}


def _prog(dtype, sched, gen=DkdvGen, D=128, persist=True):
    """persist (forward, dQ): the work-list form of the body (what ships) or the one-item-per-workgroup form"""
    key = (dtype, sched, gen, D, persist)
    if key not in _PROGS:
        kw = {} if gen is DkdvGen else {"persist": persist}
        _PROGS[key] = gen(dtype, sched=sched, D=D, **kw).build()
    return _PROGS[key]


def _case(B, Hq, Hkv, N, Nk, ns, W, dtype, seed, D=128):
    g = torch.Generator().manual_seed(seed)
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    q, do = (torch.randn(B, Hq, N, D, generator=g).to(td) for _ in range(2))
    k, v = (torch.randn(B, Hkv, Nk, D, generator=g).to(td) for _ in range(2))
    o, lse = O.sink_attention_dense(q, k, v, ns, W)
    delta = (do.double() * o).sum(-1)
    dq, dk, dv, _ = O.sink_attention_bwd_dense(q, k, v, do, ns, W)
    _case.dq = dq
    return q, k, v, do, lse, delta, dk, dv


@pytest.mark.parametrize("B,Hq,Hkv,N,Nk,ns,W,dtype", [
    (1, 4, 1, 300, 300, 4, 100, "bf16"),      # two key blocks, GQA group of 4, sink + window, ragged last slice
    (1, 2, 2, 333, 333, 70, 50, "f16"),       # sinks spanning three 32-key sub-blocks
    (2, 2, 1, 77, 200, 3, 64, "bf16"),        # N_q < N_kv (queries are the last 77 positions)
    (1, 2, 1, 40, 40, 4, 1, "bf16"),          # window of one key
])
def test_dkdv_body_in_emulator_matches_oracle(B, Hq, Hkv, N, Nk, ns, W, dtype):
    q, k, v, do, lse, delta, dk, dv = _case(B, Hq, Hkv, N, Nk, ns, W, dtype, seed=N)
    dk_e, dv_e = run_dkdv(_prog(dtype, True), q, k, v, do, lse, delta, ns, W, dtype)
    # tolerance of the GPU parity tests for 16-bit gradients (tests/test_gpu_prefill.py): atol 1e-1 + rtol 5e-2
    for got, ref, name in ((dk_e, dk, "dk"), (dv_e, dv, "dv")):
        err = (got.double() - ref).abs()
        assert (err <= 1e-1 + 5e-2 * ref.abs()).all(), (name, err.max().item())
        assert err.max().item() < 4e-2, (name, err.max().item())


def test_dkdv_f32_partials_of_a_split_sweep():
    """the second epilogue (split sweeps): the f32 partial dK / dV of a block's first rows is what the 16-bit output rounds
    (same accumulators, no rounding), rows beyond fall outside the descriptor and stay untouched"""
    q, k, v, do, lse, delta, dk, dv = _case(1, 2, 1, 96, 290, 4, 70, "bf16", seed=5)
    rows = 40
    dk_e, dv_e, pk, pv = run_dkdv(_prog("bf16", True), q, k, v, do, lse, delta, 4, 70, "bf16", part_rows=rows)
    for kb in range(2):
        n = min(rows, 290 - 256 * kb)
        for part, out, ref in ((pk, dk_e, dk), (pv, dv_e, dv)):
            sl = slice(256 * kb, 256 * kb + n)
            assert torch.equal(part[0, 0, kb, :n].bfloat16().float(), out[0, 0, sl])          # the output IS the rounded partial
            assert (part[0, 0, kb, :n].double() - ref[0, 0, sl]).abs().max().item() < 2e-2
            assert (part[0, 0, kb, n:] == 0).all()


def test_scheduled_body_equals_program_order_bitwise():
    q, k, v, do, lse, delta, _, _ = _case(1, 2, 1, 96, 290, 4, 70, "bf16", seed=5)
    a = run_dkdv(_prog("bf16", False), q, k, v, do, lse, delta, 4, 70, "bf16")
    b = run_dkdv(_prog("bf16", True), q, k, v, do, lse, delta, 4, 70, "bf16")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm assembler not installed")
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_body_assembles_for_gfx950(dtype):
    ok, err = assemble(_prog(dtype, True))
    assert ok, err[:4000]


def test_stamped_diagnostic_body_runs_and_agrees():
    """the s_memtime-stamped build of the body (tools/stamps_dkdv.py) stores only into its own debug records and
    computes the same gradients"""
    q, k, v, do, lse, delta, _, _ = _case(1, 2, 1, 96, 290, 4, 70, "bf16", seed=5)
    a = run_dkdv(_prog("bf16", True), q, k, v, do, lse, delta, 4, 70, "bf16")
    b = run_dkdv(DkdvGen("bf16", stamps=True).build(), q, k, v, do, lse, delta, 4, 70, "bf16", stamped=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("B,Hq,Hkv,N,Nk,ns,W,dtype", [
    (1, 4, 1, 300, 300, 4, 100, "bf16"),      # 4 heads x 64 rows per workgroup, sink tile + window tiles, ragged rows
    (1, 2, 2, 333, 333, 70, 50, "f16"),       # MHA: one head x 256 rows per workgroup; sinks over two key tiles
    (2, 2, 1, 77, 200, 3, 64, "bf16"),        # 2 heads x 128 rows; N_q < N_kv
    (1, 2, 1, 40, 40, 4, 1, "bf16"),          # window of one key
    (1, 1, 1, 700, 700, 4, 192, "bf16"),      # W >= 128: window-edge and causal-edge tiles (typed bodies)
    (1, 4, 1, 520, 650, 0, 256, "f16"),       # the same with N_q < N_kv and no sinks
])
def test_dq_body_in_emulator_matches_oracle(B, Hq, Hkv, N, Nk, ns, W, dtype):
    """the work-list (persistent) body: two workgroups share the items, so every item transition (K / V ring running on
    into the next item, requests before the stores) and the last-item exit are executed, under the emulator's wait and
    LDS-DMA race checks"""
    q, k, v, do, lse, delta, _, _ = _case(B, Hq, Hkv, N, Nk, ns, W, dtype, seed=N + 1)
    ref = _case.dq
    got = run_dq(_prog(dtype, True, DqGen), q, k, v, do, lse, delta, ns, W, dtype, persist=True, n_wg=2)
    err = (got.double() - ref).abs()
    assert (err <= 5e-2 + 5e-2 * ref.abs()).all() and err.max().item() < 3e-2, err.max().item()


@pytest.mark.parametrize("n_wg", [1, 3])
def test_dq_work_list_body_equals_one_item_body_bitwise(n_wg):
    """every item of a work list computes what the one-item-per-workgroup body computes (the A/B partner kept in the
    library), whatever the number of workgroups the list is cut for; scheduled = program order for both"""
    q, k, v, do, lse, delta, _, _ = _case(1, 4, 1, 200, 200, 4, 70, "bf16", seed=6)
    a = run_dq(_prog("bf16", False, DqGen, persist=False), q, k, v, do, lse, delta, 4, 70, "bf16")
    b = run_dq(_prog("bf16", True, DqGen, persist=False), q, k, v, do, lse, delta, 4, 70, "bf16")
    c = run_dq(_prog("bf16", True, DqGen), q, k, v, do, lse, delta, 4, 70, "bf16", persist=True, n_wg=n_wg)
    assert torch.equal(a, b) and torch.equal(a, c)


@pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm assembler not installed")
@pytest.mark.parametrize("dtype,persist", [("bf16", True), ("f16", True), ("bf16", False)])
def test_dq_body_assembles_for_gfx950(dtype, persist):
    ok, err = assemble(_prog(dtype, True, DqGen, persist=persist))
    assert ok, err[:4000]


@pytest.mark.parametrize("B,Hq,Hkv,N,Nk,ns,W,dtype,aux,spike", [
    (1, 4, 1, 300, 300, 4, 100, "bf16", True, False),     # 4 heads x 64 rows, s_aux, sink tile + window tiles, ragged rows
    (1, 2, 2, 333, 333, 70, 50, "f16", True, False),      # MHA: one head x 256 rows per workgroup; sinks over two tiles
    (2, 2, 1, 77, 200, 3, 64, "bf16", False, False),      # N_q < N_kv, no s_aux (rows start from m = -inf)
    (1, 1, 1, 600, 600, 0, 600, "bf16", True, True),      # causal; a spiked key forces a late move of the reference point
    (1, 2, 1, 40, 40, 4, 1, "bf16", False, False),        # window of one key
    (1, 1, 1, 700, 700, 4, 192, "bf16", True, False),     # W >= 128: window-edge and causal-edge tiles take the typed bodies
    (1, 4, 1, 520, 650, 0, 256, "f16", False, False),     # the same with N_q < N_kv and no sinks
])
def test_fwd_body_in_emulator_matches_oracle(B, Hq, Hkv, N, Nk, ns, W, dtype, aux, spike):
    g = torch.Generator().manual_seed(N + 2)
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    q = torch.randn(B, Hq, N, 128, generator=g).to(td)
    k, v = (torch.randn(B, Hkv, Nk, 128, generator=g).to(td) for _ in range(2))
    if spike:    # rule 26 of the CDNA guide: the rescale branch needs an input that takes it (row maximum jumps by > 2^8)
        k[:, :, Nk - 40] = q[:, 0, N - 1] * 3
    sa = torch.randn(Hq, generator=g) * 0.5 if aux else None
    o_ref, lse_ref = O.sink_attention_dense(q, k, v, ns, W, sa)
    o, lse = run_fwd(_prog(dtype, True, FwdGen), q, k, v, ns, W, sa, dtype, persist=True, n_wg=2)      # the work-list body
    assert (o.double() - o_ref).abs().max().item() < (1e-2 if dtype == "bf16" else 2e-3)
    fin = torch.isfinite(lse_ref)
    assert (lse.double()[fin] - lse_ref[fin]).abs().max().item() < 5e-3
    assert (lse[~fin] == float("-inf")).all()


@pytest.mark.parametrize("n_wg", [1, 3])
def test_fwd_work_list_body_equals_one_item_body_bitwise(n_wg):
    g = torch.Generator().manual_seed(8)
    q = torch.randn(1, 4, 200, 128, generator=g).bfloat16()
    k, v = (torch.randn(1, 1, 200, 128, generator=g).bfloat16() for _ in range(2))
    sa = torch.randn(4, generator=g) * 0.5
    a = run_fwd(_prog("bf16", False, FwdGen, persist=False), q, k, v, 4, 70, sa, "bf16")
    b = run_fwd(_prog("bf16", True, FwdGen, persist=False), q, k, v, 4, 70, sa, "bf16")
    c = run_fwd(_prog("bf16", True, FwdGen), q, k, v, 4, 70, sa, "bf16", persist=True, n_wg=n_wg)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])


@pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm assembler not installed")
@pytest.mark.parametrize("dtype,persist", [("bf16", True), ("f16", True), ("bf16", False)])
def test_fwd_body_assembles_for_gfx950(dtype, persist):
    ok, err = assemble(_prog(dtype, True, FwdGen, persist=persist))
    assert ok, err[:4000]


@pytest.mark.parametrize("D", [64, 80, 96])
def test_other_head_dims_in_emulator(D):
    """head dims 64 / 80 / 96: fewer k-steps and output blocks, padding chunks of the 256-byte LDS rows fetched as zeros
    (80, 96) or not at all (64); all three bodies against the oracle, and they assemble"""
    B, Hq, Hkv, N, Nk, ns, W = 1, 4, 1, 200, 200, 4, 90
    q, k, v, do, lse, delta, dk, dv = _case(B, Hq, Hkv, N, Nk, ns, W, "bf16", seed=D, D=D)
    dq = _case.dq
    dk_e, dv_e = run_dkdv(_prog("bf16", True, DkdvGen, D), q, k, v, do, lse, delta, ns, W, "bf16")
    dq_e = run_dq(_prog("bf16", True, DqGen, D), q, k, v, do, lse, delta, ns, W, "bf16", persist=True, n_wg=2)
    g = torch.Generator().manual_seed(1)
    sa = torch.randn(Hq, generator=g) * 0.5
    o_ref, lse_ref = O.sink_attention_dense(q, k, v, ns, W, sa)
    o_e, lse_e = run_fwd(_prog("bf16", True, FwdGen, D), q, k, v, ns, W, sa, "bf16", persist=True, n_wg=2)
    assert (dk_e.double() - dk).abs().max().item() < 4e-2 and (dv_e.double() - dv).abs().max().item() < 4e-2
    assert (dq_e.double() - dq).abs().max().item() < 3e-2
    assert (o_e.double() - o_ref).abs().max().item() < 1e-2 and (lse_e.double() - lse_ref).abs().max().item() < 5e-3
    if os.path.exists(CLANG):
        for gen in (DkdvGen, DqGen, FwdGen):
            ok, err = assemble(_prog("bf16", True, gen, D))
            assert ok, err[:2000]


@pytest.mark.parametrize("Hq,Hkv,N,D,W,NT,strip,dtype,aux,spike", [
    (4, 1, 400, 80, 128, 3, 3, "bf16", True, False),      # gpt-oss sliding layer shape (C4): 3 tiles per item, strips of 3 + a ragged last
    (8, 2, 300, 64, 100, 3, 5, "f16", False, False),      # no s_aux (rows start from m = -inf), window not a tile multiple, two KV heads
    (4, 1, 260, 96, 64, 2, 2, "bf16", True, True),        # two tiles per item; a spiked key forces the out-of-line rescale
    (4, 1, 330, 64, 192, 4, 6, "bf16", True, False),      # four tiles per item (5-slot ring), one strip over the whole sequence
])
def test_short_window_strip_forward_in_emulator(Hq, Hkv, N, D, W, NT, strip, dtype, aux, spike):
    """tools/asmgen/fwd_strip.py: strips of consecutive query tiles over a sliding K / V ring, double-buffered Q fragments,
    the finished item's stores under the next item's first MFMAs; against the oracle, under the emulator's wait / LDS-DMA
    race checks; the first items of a sequence (tile indices below 0) go through the same blocks"""
    from asmgen.fwd_strip import FwdStripGen
    from asmgen.harness import run_fwd_strip
    g = torch.Generator().manual_seed(N + D)
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    q = torch.randn(1, Hq, N, D, generator=g).to(td)
    k, v = (torch.randn(1, Hkv, N, D, generator=g).to(td) for _ in range(2))
    if spike:
        k[:, :, N - 30] = q[:, 0, N - 1] * 3
    sa = torch.randn(Hq, generator=g) * 0.5 if aux else None
    prog = FwdStripGen(dtype, D=D, NT=NT).build()
    if os.path.exists(CLANG):
        ok, err = assemble(prog)
        assert ok, err[:2000]
    o_ref, lse_ref = O.sink_attention_dense(q, k, v, 0, W, sa)
    o, lse = run_fwd_strip(prog, q, k, v, W, sa, dtype, NT=NT, strip=strip)
    assert (o.double() - o_ref).abs().max().item() < (1e-2 if dtype == "bf16" else 2e-3)
    assert (lse.double() - lse_ref).abs().max().item() < 5e-3


@pytest.mark.parametrize("Hq,Hkv,N,D,W,NT,strip,dtype", [
    (4, 1, 400, 80, 128, 3, 3, "bf16"),      # gpt-oss sliding layer shape (C4): 3 tiles per item, steady blocks after the first two items
    (8, 2, 300, 64, 100, 3, 5, "f16"),       # window not a tile multiple (general blocks throughout), two KV heads
    (4, 1, 330, 64, 192, 4, 6, "bf16"),      # four tiles per item (5-slot ring), one strip over the whole sequence
])
def test_short_window_strip_dq_in_emulator(Hq, Hkv, N, D, W, NT, strip, dtype):
    """tools/asmgen/dq_strip.py: the backward twin of the strip forward (sliding K / V ring, double-buffered Q / dO fragments
    and row constants, the finished item's dQ stores under the next item's MFMAs), against the oracle under the emulator's
    wait / LDS-DMA race checks"""
    from asmgen.dq_strip import DqStripGen
    from asmgen.harness import run_dq_strip
    g = torch.Generator().manual_seed(N + D)
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    q, do = (torch.randn(1, Hq, N, D, generator=g).to(td) for _ in range(2))
    k, v = (torch.randn(1, Hkv, N, D, generator=g).to(td) for _ in range(2))
    o, lse = O.sink_attention_dense(q, k, v, 0, W)
    delta = (do.double() * o).sum(-1)
    dq_ref, _, _, _ = O.sink_attention_bwd_dense(q, k, v, do, 0, W)
    prog = DqStripGen(dtype, D=D, NT=NT).build()
    if os.path.exists(CLANG):
        ok, err = assemble(prog)
        assert ok, err[:2000]
    dq = run_dq_strip(prog, q, k, v, do, lse, delta, W, dtype, NT=NT, strip=strip)
    err = (dq.double() - dq_ref).abs()
    assert (err <= 5e-2 + 5e-2 * dq_ref.abs()).all() and err.max().item() < 3e-2, err.max().item()


# ------------------------------------------------------------------------------------------------ skewed dK/dV sweep
@pytest.mark.parametrize("B,Hq,Hkv,N,D,W,dtype", [
    (1, 2, 1, 300, 80, 128, "bf16"),      # the gpt-oss window: T = 6, two key blocks, ragged last slice
    (1, 3, 1, 520, 64, 100, "bf16"),      # three blocks, group of 3
    (1, 2, 2, 333, 64, 200, "f16"),       # T = 9: trips without a B slice; MHA
    (2, 1, 1, 257, 96, 33, "bf16")])      # T rounded up to 6; a last block of one key
def test_dkdv_skew_body_in_emulator_matches_oracle(B, Hq, Hkv, N, D, W, dtype):
    """the short-window dK/dV body (tools/asmgen/dkdv_skew.py: wave w on slice r of head h or slice T + r of head h - 1)"""
    from asmgen.dkdv_skew import DkdvSkewGen
    from asmgen.harness import run_dkdv_skew
    q, k, v, do, lse, delta, dk, dv = _case(B, Hq, Hkv, N, N, 0, W, dtype, seed=N, D=D)
    dk_e, dv_e = run_dkdv_skew(DkdvSkewGen(dtype, D=D).build(), q, k, v, do, lse, delta, W, dtype)
    for got, ref, name in ((dk_e, dk, "dk"), (dv_e, dv, "dv")):
        err = (got.double() - ref).abs()
        assert (err <= 1e-1 + 5e-2 * ref.abs()).all(), (name, err.max().item())
        assert err.max().item() < 4e-2, (name, err.max().item())


def test_dkdv_skew_scheduled_equals_program_order_and_assembles():
    from asmgen.dkdv_skew import DkdvSkewGen
    from asmgen.harness import run_dkdv_skew
    q, k, v, do, lse, delta, _, _ = _case(1, 2, 1, 300, 300, 0, 128, "bf16", seed=300, D=80)
    a = run_dkdv_skew(DkdvSkewGen("bf16", D=80, sched=False).build(), q, k, v, do, lse, delta, 128, "bf16")
    b = run_dkdv_skew(DkdvSkewGen("bf16", D=80).build(), q, k, v, do, lse, delta, 128, "bf16")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    if os.path.exists(CLANG):
        for D in (64, 80, 96):
            ok, err = assemble(DkdvSkewGen("bf16", D=D).build())
            assert ok, err[:4000]
