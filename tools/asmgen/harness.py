"""Runs the generated dK/dV kernel body in the CPU emulator on torch tensors (TEST INFRASTRUCTURE).

`dkdv_block_params` restates, in Python, the scalar prologue of the HIP shell around the asm statement
(csrc/sfa_bwd_asm.hip: block -> slice range, base pointers, ranges); tests/test_asm_emu.py runs every key block of a
small problem through the emulator and compares dK / dV with the oracle.
"""
from __future__ import annotations

import math
import struct

import numpy as np
import torch

from . import dkdv as K
from .emu import Memory, Workgroup


def f32_bits(x: float) -> int:
    return struct.unpack("<I", struct.pack("<f", x))[0]


def to_u16(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def dkdv_block_params(N, Nk, D, Hq, Hkv, ns, window, kb):
    """slice range of the 256-key block `kb` (the shell's arithmetic; query row i sits at key position i + Nk - N)"""
    P = Nk - N
    W = min(max(window, 0), Nk)
    kb0 = kb * 256
    kb1 = min(kb0 + 256, Nk)
    if kb0 < ns:
        i_hi = N
    else:
        i_hi = min(max(kb1 - 1 + W - P, 0), N)
    qt_lo = max(kb0 - P, 0) // 32
    qt_hi = max((i_hi + 31) // 32, qt_lo)
    return dict(P=P, W=W, kb0=kb0, qt_lo=qt_lo, nq=qt_hi - qt_lo)


def run_dkdv(prog, q, k, v, do, lse, delta, ns, window, dtype="bf16", check_races=True, blocks=None, stats=None,
             stamped=False, part_rows=0):
    """q, do [B, Hq, N, D]; k, v [B, Hkv, Nk, D] (torch, bf16 / f16); lse, delta [B, Hq, N] float.
    Returns dk, dv [B, Hkv, Nk, D] float32 (as stored by the kernel: 16-bit values)."""
    B, Hq, N, D = q.shape
    Hkv, Nk = k.shape[1], k.shape[2]
    g = Hq // Hkv
    scale = 1.0 / math.sqrt(D)
    mem = Memory()
    aq, ak, av, ado = (mem.alloc(to_u16(t)) for t in (q, k, v, do))
    consts = torch.stack([-(lse.double() / scale), -delta.double()], dim=2).float().contiguous()   # [B, Hq, 2, N]
    ac = mem.alloc(consts.numpy())
    adk = mem.alloc_zero(B * Hkv * Nk * D * 2)
    adv = mem.alloc_zero(B * Hkv * Nk * D * 2)
    nkb = (Nk + 255) // 256
    dbg = mem.alloc_zero(B * Hkv * nkb * 64) if stamped else None
    # part_rows > 0: every block also leaves the f32 partial of its first part_rows keys (the split sweeps' epilogue),
    # [b, hk, block][dk | dv][part_rows][D]; returned as a third / fourth result
    apart = mem.alloc_zero(B * Hkv * nkb * 2 * part_rows * D * 4) if part_rows else None
    for b in range(B):
        for hk in range(Hkv):
            for kb in range(nkb):
                if blocks is not None and (b, hk, kb) not in blocks:
                    continue
                bp = dkdv_block_params(N, Nk, D, Hq, Hkv, ns, window, kb)
                head0 = hk * g
                qb = aq + ((b * Hq + head0) * N) * D * 2
                dob = ado + ((b * Hq + head0) * N) * D * 2
                cb = ac + ((b * Hq + head0) * 2 * N) * 4
                kbp = ak + ((b * Hkv + hk) * Nk) * D * 2
                vbp = av + ((b * Hkv + hk) * Nk) * D * 2
                dkb = adk + ((b * Hkv + hk) * Nk) * D * 2
                dvb = adv + ((b * Hkv + hk) * Nk) * D * 2
                rng_q = ((N - 1) * D + D) * 2
                rng_k = ((Nk - 1) * D + D) * 2
                params = dict(
                    q_lo=qb & 0xFFFFFFFF, q_hi=qb >> 32, do_lo=dob & 0xFFFFFFFF, do_hi=dob >> 32,
                    c_lo=cb & 0xFFFFFFFF, c_hi=cb >> 32, k_lo=kbp & 0xFFFFFFFF, k_hi=kbp >> 32,
                    v_lo=vbp & 0xFFFFFFFF, v_hi=vbp >> 32, dk_lo=dkb & 0xFFFFFFFF, dk_hi=dkb >> 32,
                    dv_lo=dvb & 0xFFFFFFFF, dv_hi=dvb >> 32,
                    q_rng=rng_q, do_rng=rng_q, c_rng=2 * N * 4, k_rng=rng_k, v_rng=rng_k, dk_rng=rng_k, dv_rng=rng_k,
                    q_sn=D * 2, do_sn=D * 2, k_sn=D * 2, v_sn=D * 2, dk_sn=D * 2, dv_sn=D * 2,
                    q_hs=N * D * 2, do_hs=N * D * 2, c_hs=2 * N * 4,
                    nq=bp["nq"], g=g, q_row0=bp["qt_lo"] * 32, kb0=bp["kb0"], pos0=bp["P"], W=bp["W"], ns=ns, nrows=N,
                    cdelta=N * 4, c_log2=f32_bits(scale * math.log2(math.e)), scale=f32_bits(scale),
                    pk_lo=0, pk_hi=0, pv_lo=0, pv_hi=0, p_rng=0)
                if apart is not None:
                    pb = apart + (((b * Hkv + hk) * nkb + kb) * 2 * part_rows * D) * 4 - bp["kb0"] * D * 4
                    pvb = pb + part_rows * D * 4
                    params.update(pk_lo=pb & 0xFFFFFFFF, pk_hi=pb >> 32, pv_lo=pvb & 0xFFFFFFFF, pv_hi=pvb >> 32,
                                  p_rng=(bp["kb0"] + part_rows) * D * 4)
                assert set(params) == set(K.PARAMS), set(params) ^ set(K.PARAMS)
                if dbg is not None:      # stamped diagnostic body
                    params.update(dbg_lo=dbg & 0xFFFFFFFF, dbg_hi=dbg >> 32, bid=(b * Hkv + hk) * nkb + kb)
                wg = Workgroup(prog, 4, mem, params, lds_bytes=160 * 1024, check_races=check_races)
                wg.run()
                if stats is not None:
                    stats.append({"block": (b, hk, kb), "nq": bp["nq"], "icount": [w.icount for w in wg.waves],
                                  "kinds": dict(wg.waves[0].stats)})
    def back(addr):
        raw = mem.read(addr).view(np.uint16).reshape(B, Hkv, Nk, D)
        t = torch.from_numpy(raw.view(np.int16).copy())
        return t.view(torch.bfloat16 if dtype == "bf16" else torch.float16).float()
    if apart is not None:
        pp = torch.from_numpy(mem.read(apart).view(np.float32).reshape(B, Hkv, nkb, 2, part_rows, D).copy())
        return back(adk), back(adv), pp[:, :, :, 0], pp[:, :, :, 1]
    return back(adk), back(adv)


# ------------------------------------------------------------------------------------------------ dQ kernel
def dq_block_params(N, Nk, ns, window, g, qt, hpw):
    """tile list of the workgroup that owns query tile `qt` (BM = 64 * 4 / hpw rows); the shell's arithmetic"""
    P = Nk - N
    W = min(max(window, 0), Nk)
    BM = 64 * (4 // hpw)
    q0 = qt * BM
    q1 = min(q0 + BM, N)
    ns_eff = min(ns, q1 + P)
    ts_hi = (ns_eff + 63) >> 6
    wlo = max(q0 + P - W + 1, 0)
    tw_lo = max(wlo >> 6, ts_hi)
    tw_hi = (q1 + P + 63) >> 6
    tw_lo = min(tw_lo, tw_hi)
    return dict(P=P, W=W, q0=q0, ts_hi=ts_hi, tw_off=tw_lo - ts_hi, nt=ts_hi + (tw_hi - tw_lo), BM=BM)


def worklist_of(n_ranks, n_wg, w):
    """ranks (cost-descending item order) of workgroup w of n_wg, in processing order: snake over the rounds (round j
    hands rank j n_wg + pos to the workgroup; pos runs forwards in even rounds, backwards in odd ones, inside the
    workgroup's XCD lane w % 8 when the grid is a multiple of 8) - the HIP shells' rule (csrc/sfa_worklist.hpp)"""
    out = []
    j = 0
    while j * n_wg < n_ranks:
        if j & 1:
            pos = ((n_wg // 8 - 1 - w // 8) * 8 + w % 8) if n_wg % 8 == 0 else n_wg - 1 - w
        else:
            pos = w
        r = j * n_wg + pos
        if r < n_ranks:
            out.append(r)
        j += 1
    return out


def run_worklist(prog, mem, all_items, consts, desc_map, desc_base, n_wg, check_races=True, stats=None):
    """all_items: per rank a dict of descriptor fields or None (an empty item: skipped, as the shell's compaction does);
    runs n_wg persistent workgroups"""
    for w in range(n_wg):
        mine = [all_items[r] for r in worklist_of(len(all_items), n_wg, w) if all_items[r] is not None]
        if not mine:
            continue
        assert len(mine) <= 256
        params = dict(consts, n_items=len(mine))
        wg = Workgroup(prog, 4, mem, params, lds_bytes=160 * 1024, check_races=check_races)
        tab = np.zeros((len(mine), 32), dtype=np.uint32)
        for i, it in enumerate(mine):
            assert set(it) == set(desc_map), set(it) ^ set(desc_map)
            for name, val in it.items():
                tab[i, desc_map[name]] = int(val) & 0xFFFFFFFF
        raw = tab.view(np.uint8).reshape(-1)
        wg.lds[desc_base:desc_base + raw.size] = raw
        wg.run()
        if stats is not None:
            stats.append({"wg": w, "n_items": len(mine), "icount": [x.icount for x in wg.waves]})


def run_dq(prog, q, k, v, do, lse, delta, ns, window, dtype="bf16", check_races=True, stats=None, persist=False, n_wg=3):
    """q, do [B, Hq, N, D]; k, v [B, Hkv, Nk, D]; lse, delta [B, Hq, N].  Returns dq [B, Hq, N, D] float32.
    persist: the work-list form of the body (n_wg persistent workgroups share the items)."""
    from . import dq as KQ
    B, Hq, N, D = q.shape
    Hkv, Nk = k.shape[1], k.shape[2]
    g = Hq // Hkv
    hpw = math.gcd(g, 4)
    scale = 1.0 / math.sqrt(D)
    mem = Memory()
    aq, ak, av, ado = (mem.alloc(to_u16(t)) for t in (q, k, v, do))
    alse = mem.alloc(lse.float().contiguous().numpy())
    adl = mem.alloc(delta.float().contiguous().numpy())
    adq = mem.alloc_zero(B * Hq * N * D * 2)
    BM = 64 * (4 // hpw)
    nqt = (N + BM - 1) // BM
    if persist:
        rng_q = ((N - 1) * D + D) * 2
        rng_k = ((Nk - 1) * D + D) * 2
        lo = lambda x: x & 0xFFFFFFFF
        ngrp = B * Hkv * (g // hpw)
        all_items = []
        for r in range(nqt * ngrp):                      # rank -> (query tile, descending; group)
            qt, grp = nqt - 1 - r // ngrp, r % ngrp
            hg = grp % (g // hpw)
            hk = (grp // (g // hpw)) % Hkv
            b = grp // (g // hpw) // Hkv
            bp = dq_block_params(N, Nk, ns, window, g, qt, hpw)
            head0 = hk * g + hg * hpw
            hb = (b * Hq + head0) * N
            kb = (b * Hkv + hk) * Nk * D * 2
            all_items.append(dict(
                q_lo=lo(aq + hb * D * 2), q_hi=(aq + hb * D * 2) >> 32, do_lo=lo(ado + hb * D * 2), do_hi=(ado + hb * D * 2) >> 32,
                lse_lo=lo(alse + hb * 4), lse_hi=(alse + hb * 4) >> 32, dl_lo=lo(adl + hb * 4), dl_hi=(adl + hb * 4) >> 32,
                dq_lo=lo(adq + hb * D * 2), dq_hi=(adq + hb * D * 2) >> 32, q0=bp["q0"], nrows=N,
                q_rng=rng_q, do_rng=rng_q, dq_rng=rng_q,
                k_lo=lo(ak + kb), k_hi=(ak + kb) >> 32, v_lo=lo(av + kb), v_hi=(av + kb) >> 32, k_rng=rng_k, v_rng=rng_k,
                nt=bp["nt"], ts_hi=bp["ts_hi"], tw_off=bp["tw_off"]))
        consts = dict(q_hs=N * D * 2, q_sn=D * 2, do_hs=N * D * 2, do_sn=D * 2, dq_hs=N * D * 2, dq_sn=D * 2,
                      k_sn=D * 2, v_sn=D * 2, ld_hs=N * 4, pos0=Nk - N, W=min(max(window, 0), Nk), ns=ns,
                      hpw_log2={1: 0, 2: 1, 4: 2}[hpw], c_log2=f32_bits(scale * math.log2(math.e)),
                      nlog2e=f32_bits(-math.log2(math.e)), scale=f32_bits(scale))
        assert set(consts) | {"n_items"} == set(KQ.PARAMS_PK)
        run_worklist(prog, mem, all_items, consts, KQ.DESC, KQ.DESC_BASE, n_wg, check_races, stats)
        nqt = 0                                          # (skip the one-item loop below)
    for b in range(B):
        for hk in range(Hkv):
            for hg in range(g // hpw):
                for qt in range(nqt):
                    bp = dq_block_params(N, Nk, ns, window, g, qt, hpw)
                    head0 = hk * g + hg * hpw
                    hb = (b * Hq + head0) * N
                    kb = (b * Hkv + hk) * Nk * D * 2
                    rng_q = ((N - 1) * D + D) * 2
                    rng_k = ((Nk - 1) * D + D) * 2
                    lo = lambda x: x & 0xFFFFFFFF
                    params = dict(
                        q_lo=lo(aq + hb * D * 2), q_hi=(aq + hb * D * 2) >> 32, q_hs=N * D * 2, q_sn=D * 2, q_rng=rng_q,
                        do_lo=lo(ado + hb * D * 2), do_hi=(ado + hb * D * 2) >> 32, do_hs=N * D * 2, do_sn=D * 2, do_rng=rng_q,
                        dq_lo=lo(adq + hb * D * 2), dq_hi=(adq + hb * D * 2) >> 32, dq_hs=N * D * 2, dq_sn=D * 2, dq_rng=rng_q,
                        k_lo=lo(ak + kb), k_hi=(ak + kb) >> 32, k_sn=D * 2, k_rng=rng_k,
                        v_lo=lo(av + kb), v_hi=(av + kb) >> 32, v_sn=D * 2, v_rng=rng_k,
                        lse_lo=lo(alse + hb * 4), lse_hi=(alse + hb * 4) >> 32, dl_lo=lo(adl + hb * 4), dl_hi=(adl + hb * 4) >> 32,
                        ld_hs=N * 4, q0=bp["q0"], nrows=N, pos0=bp["P"], W=bp["W"], ns=ns, nt=bp["nt"], ts_hi=bp["ts_hi"],
                        tw_off=bp["tw_off"], hpw_log2={1: 0, 2: 1, 4: 2}[hpw], c_log2=f32_bits(scale * math.log2(math.e)),
                        nlog2e=f32_bits(-math.log2(math.e)), scale=f32_bits(scale))
                    assert set(params) == set(KQ.PARAMS), set(params) ^ set(KQ.PARAMS)
                    wg = Workgroup(prog, 4, mem, params, lds_bytes=160 * 1024, check_races=check_races)
                    wg.run()
                    if stats is not None:
                        stats.append({"block": (b, hk, hg, qt), "nt": bp["nt"], "icount": [w.icount for w in wg.waves]})
    raw = mem.read(adq).view(np.uint16).reshape(B, Hq, N, D)
    t = torch.from_numpy(raw.view(np.int16).copy())
    return t.view(torch.bfloat16 if dtype == "bf16" else torch.float16).float()


# ------------------------------------------------------------------------------------------------ forward kernel
def run_fwd(prog, q, k, v, ns, window, s_aux=None, dtype="bf16", check_races=True, stats=None, persist=False, n_wg=3):
    """q [B, Hq, N, D]; k, v [B, Hkv, Nk, D]; s_aux [Hq] or None.  Returns o [B, Hq, N, D] float32, lse [B, Hq, N].
    persist: the work-list form of the body (n_wg persistent workgroups share the items)."""
    from . import fwd as KF
    B, Hq, N, D = q.shape
    Hkv, Nk = k.shape[1], k.shape[2]
    g = Hq // Hkv
    hpw = math.gcd(g, 4)
    scale = 1.0 / math.sqrt(D)
    log2e = math.log2(math.e)
    mem = Memory()
    aq, ak, av = (mem.alloc(to_u16(t)) for t in (q, k, v))
    ao = mem.alloc_zero(B * Hq * N * D * 2)
    alse = mem.alloc_zero(B * Hq * N * 4)
    BM = 64 * (4 // hpw)
    nqt = (N + BM - 1) // BM
    if persist:
        rng_q = ((N - 1) * D + D) * 2
        rng_k = ((Nk - 1) * D + D) * 2
        lo = lambda x: x & 0xFFFFFFFF
        ngrp = B * Hkv * (g // hpw)
        all_items = []
        for r in range(nqt * ngrp):
            qt, grp = nqt - 1 - r // ngrp, r % ngrp
            hg = grp % (g // hpw)
            hk = (grp // (g // hpw)) % Hkv
            b = grp // (g // hpw) // Hkv
            bp = dq_block_params(N, Nk, ns, window, g, qt, hpw)
            head0 = hk * g + hg * hpw
            hb = (b * Hq + head0) * N
            kb = (b * Hkv + hk) * Nk * D * 2
            m0 = [f32_bits(float(s_aux[head0 + i]) * log2e) if (s_aux is not None and i < hpw) else KF.NEG_INF for i in range(4)]
            all_items.append(dict(
                q_lo=lo(aq + hb * D * 2), q_hi=(aq + hb * D * 2) >> 32, o_lo=lo(ao + hb * D * 2), o_hi=(ao + hb * D * 2) >> 32,
                lse_lo=lo(alse + hb * 4), lse_hi=(alse + hb * 4) >> 32, q0=bp["q0"], nrows=N, q_rng=rng_q, o_rng=rng_q,
                m0_0=m0[0], m0_1=m0[1], m0_2=m0[2], m0_3=m0[3],
                k_lo=lo(ak + kb), k_hi=(ak + kb) >> 32, v_lo=lo(av + kb), v_hi=(av + kb) >> 32, k_rng=rng_k, v_rng=rng_k,
                nt=bp["nt"], ts_hi=bp["ts_hi"], tw_off=bp["tw_off"]))
        consts = dict(q_hs=N * D * 2, q_sn=D * 2, o_hs=N * D * 2, o_sn=D * 2, k_sn=D * 2, v_sn=D * 2, ld_hs=N * 4,
                      l0=f32_bits(1.0 if s_aux is not None else 0.0), pos0=Nk - N, W=min(max(window, 0), Nk), ns=ns,
                      hpw_log2={1: 0, 2: 1, 4: 2}[hpw], c_log2=f32_bits(scale * log2e), ln2=f32_bits(math.log(2.0)))
        assert set(consts) | {"n_items"} == set(KF.PARAMS_PK)
        run_worklist(prog, mem, all_items, consts, KF.DESC, KF.DESC_BASE, n_wg, check_races, stats)
        nqt = 0
    for b in range(B):
        for hk in range(Hkv):
            for hg in range(g // hpw):
                for qt in range(nqt):
                    bp = dq_block_params(N, Nk, ns, window, g, qt, hpw)
                    head0 = hk * g + hg * hpw
                    hb = (b * Hq + head0) * N
                    kb = (b * Hkv + hk) * Nk * D * 2
                    rng_q = ((N - 1) * D + D) * 2
                    rng_k = ((Nk - 1) * D + D) * 2
                    lo = lambda x: x & 0xFFFFFFFF
                    m0 = [f32_bits(float(s_aux[head0 + i]) * log2e) if (s_aux is not None and i < hpw) else KF.NEG_INF
                          for i in range(4)]
                    params = dict(
                        q_lo=lo(aq + hb * D * 2), q_hi=(aq + hb * D * 2) >> 32, q_hs=N * D * 2, q_sn=D * 2, q_rng=rng_q,
                        o_lo=lo(ao + hb * D * 2), o_hi=(ao + hb * D * 2) >> 32, o_hs=N * D * 2, o_sn=D * 2, o_rng=rng_q,
                        k_lo=lo(ak + kb), k_hi=(ak + kb) >> 32, k_sn=D * 2, k_rng=rng_k,
                        v_lo=lo(av + kb), v_hi=(av + kb) >> 32, v_sn=D * 2, v_rng=rng_k,
                        lse_lo=lo(alse + hb * 4), lse_hi=(alse + hb * 4) >> 32, ld_hs=N * 4,
                        m0_0=m0[0], m0_1=m0[1], m0_2=m0[2], m0_3=m0[3], l0=f32_bits(1.0 if s_aux is not None else 0.0),
                        q0=bp["q0"], nrows=N, pos0=bp["P"], W=bp["W"], ns=ns, nt=bp["nt"], ts_hi=bp["ts_hi"],
                        tw_off=bp["tw_off"], hpw_log2={1: 0, 2: 1, 4: 2}[hpw], c_log2=f32_bits(scale * log2e),
                        ln2=f32_bits(math.log(2.0)))
                    assert set(params) == set(KF.PARAMS), set(params) ^ set(KF.PARAMS)
                    wg = Workgroup(prog, 4, mem, params, lds_bytes=160 * 1024, check_races=check_races)
                    wg.run()
                    if stats is not None:
                        stats.append({"block": (b, hk, hg, qt), "nt": bp["nt"], "icount": [w.icount for w in wg.waves]})
    raw = mem.read(ao).view(np.uint16).reshape(B, Hq, N, D)
    t = torch.from_numpy(raw.view(np.int16).copy())
    o = t.view(torch.bfloat16 if dtype == "bf16" else torch.float16).float()
    lse = torch.from_numpy(mem.read(alse).view(np.float32).reshape(B, Hq, N).copy())
    return o, lse


# ------------------------------------------------------------------------------------------------ short-window forward (strips)
def run_fwd_strip(prog, q, k, v, window, s_aux=None, dtype="bf16", NT=3, strip=3, check_races=True, stats=None):
    """q [B, Hq, N, D]; k, v [B, Hkv, N, D] (N_q = N_kv, no sink keys, GQA group a multiple of 4); every workgroup walks a
    strip of `strip` consecutive 64-row query tiles of one (batch, KV head, head set).  Returns o f32, lse."""
    from . import fwd_strip as KS
    B, Hq, N, D = q.shape
    Hkv = k.shape[1]
    g = Hq // Hkv
    assert g % 4 == 0 and k.shape[2] == N
    scale = 1.0 / math.sqrt(D)
    log2e = math.log2(math.e)
    mem = Memory()
    aq, ak, av = (mem.alloc(to_u16(t)) for t in (q, k, v))
    ao = mem.alloc_zero(B * Hq * N * D * 2)
    alse = mem.alloc_zero(B * Hq * N * 4)
    nqt = (N + 63) // 64
    rng = ((N - 1) * D + D) * 2
    lo = lambda x: x & 0xFFFFFFFF
    for b in range(B):
        for hk in range(Hkv):
            for hg in range(g // 4):
                for s0 in range(0, nqt, strip):
                    n_it = min(strip, nqt - s0)
                    head0 = hk * g + hg * 4
                    hb = (b * Hq + head0) * N
                    kb = (b * Hkv + hk) * N * D * 2
                    m0 = [f32_bits(float(s_aux[head0 + i]) * log2e) if s_aux is not None else KS.NEG_INF for i in range(4)]
                    params = dict(
                        q_lo=lo(aq + hb * D * 2), q_hi=(aq + hb * D * 2) >> 32, q_hs=N * D * 2, q_sn=D * 2, q_rng=rng,
                        o_lo=lo(ao + hb * D * 2), o_hi=(ao + hb * D * 2) >> 32, o_hs=N * D * 2, o_sn=D * 2, o_rng=rng,
                        k_lo=lo(ak + kb), k_hi=(ak + kb) >> 32, k_sn=D * 2, k_rng=rng,
                        v_lo=lo(av + kb), v_hi=(av + kb) >> 32, v_sn=D * 2, v_rng=rng,
                        lse_lo=lo(alse + hb * 4), lse_hi=(alse + hb * 4) >> 32, ld_hs=N * 4,
                        m0_0=m0[0], m0_1=m0[1], m0_2=m0[2], m0_3=m0[3], l0=f32_bits(1.0 if s_aux is not None else 0.0),
                        q0=64 * s0, n_it=n_it, nrows=N, W=min(max(window, 0), N), c_log2=f32_bits(scale * log2e),
                        ln2=f32_bits(math.log(2.0)))
                    assert set(params) == set(KS.PARAMS), set(params) ^ set(KS.PARAMS)
                    wg = Workgroup(prog, 4, mem, params, lds_bytes=160 * 1024, check_races=check_races)
                    wg.run()
                    if stats is not None:
                        stats.append({"strip": (b, hk, hg, s0), "n_it": n_it, "icount": [w.icount for w in wg.waves]})
    raw = mem.read(ao).view(np.uint16).reshape(B, Hq, N, D)
    t = torch.from_numpy(raw.view(np.int16).copy())
    o = t.view(torch.bfloat16 if dtype == "bf16" else torch.float16).float()
    lse = torch.from_numpy(mem.read(alse).view(np.float32).reshape(B, Hq, N).copy())
    return o, lse


def run_dq_strip(prog, q, k, v, do, lse, delta, window, dtype="bf16", NT=3, strip=3, check_races=True):
    """q, do [B, Hq, N, D]; k, v [B, Hkv, N, D]; lse, delta [B, Hq, N] (N_q = N_kv, no sink keys, GQA group a multiple of 4):
    every workgroup walks a strip of `strip` consecutive 64-row query tiles.  Returns dq f32."""
    from . import dq_strip as KS
    B, Hq, N, D = q.shape
    Hkv = k.shape[1]
    g = Hq // Hkv
    assert g % 4 == 0 and k.shape[2] == N
    scale = 1.0 / math.sqrt(D)
    mem = Memory()
    aq, ak, av, ado = (mem.alloc(to_u16(t)) for t in (q, k, v, do))
    alse = mem.alloc(lse.float().contiguous().numpy())
    adl = mem.alloc(delta.float().contiguous().numpy())
    adq = mem.alloc_zero(B * Hq * N * D * 2)
    nqt = (N + 63) // 64
    rng = ((N - 1) * D + D) * 2
    lo = lambda x: x & 0xFFFFFFFF
    for b in range(B):
        for hk in range(Hkv):
            for hg in range(g // 4):
                for s0 in range(0, nqt, strip):
                    n_it = min(strip, nqt - s0)
                    head0 = hk * g + hg * 4
                    hb = (b * Hq + head0) * N
                    kb = (b * Hkv + hk) * N * D * 2
                    params = dict(
                        q_lo=lo(aq + hb * D * 2), q_hi=(aq + hb * D * 2) >> 32, q_hs=N * D * 2, q_sn=D * 2, q_rng=rng,
                        do_lo=lo(ado + hb * D * 2), do_hi=(ado + hb * D * 2) >> 32, do_hs=N * D * 2, do_sn=D * 2, do_rng=rng,
                        dq_lo=lo(adq + hb * D * 2), dq_hi=(adq + hb * D * 2) >> 32, dq_hs=N * D * 2, dq_sn=D * 2, dq_rng=rng,
                        k_lo=lo(ak + kb), k_hi=(ak + kb) >> 32, k_sn=D * 2, k_rng=rng,
                        v_lo=lo(av + kb), v_hi=(av + kb) >> 32, v_sn=D * 2, v_rng=rng,
                        lse_lo=lo(alse + hb * 4), lse_hi=(alse + hb * 4) >> 32, dl_lo=lo(adl + hb * 4), dl_hi=(adl + hb * 4) >> 32,
                        ld_hs=N * 4, q0=64 * s0, n_it=n_it, nrows=N, W=min(max(window, 0), N),
                        c_log2=f32_bits(scale * math.log2(math.e)), nlog2e=f32_bits(-math.log2(math.e)), scale=f32_bits(scale))
                    assert set(params) == set(KS.PARAMS), set(params) ^ set(KS.PARAMS)
                    wg = Workgroup(prog, 4, mem, params, lds_bytes=160 * 1024, check_races=check_races)
                    wg.run()
    raw = mem.read(adq).view(np.uint16).reshape(B, Hq, N, D)
    t = torch.from_numpy(raw.view(np.int16).copy())
    return t.view(torch.bfloat16 if dtype == "bf16" else torch.float16).float()


# ------------------------------------------------------------------------------------------------ skewed dK/dV kernel
def run_dkdv_skew(prog, q, k, v, do, lse, delta, window, dtype="bf16", check_races=True, blocks=None, stats=None):
    """the short-window body of dkdv_skew.py (no sink keys, N_q = N_kv): q, do [B, Hq, N, D]; k, v [B, Hkv, N, D];
    lse, delta [B, Hq, N] float.  Returns dk, dv [B, Hkv, N, D] float32 (the 16-bit values the kernel stored)."""
    from . import dkdv_skew as KS
    B, Hq, N, D = q.shape
    Hkv = k.shape[1]
    assert k.shape[2] == N
    g = Hq // Hkv
    scale = 1.0 / math.sqrt(D)
    mem = Memory()
    aq, ak, av, ado = (mem.alloc(to_u16(t)) for t in (q, k, v, do))
    consts = torch.stack([-(lse.double() / scale), -delta.double()], dim=2).float().contiguous()   # [B, Hq, 2, N]
    ac = mem.alloc(consts.numpy())
    adk = mem.alloc_zero(B * Hkv * N * D * 2)
    adv = mem.alloc_zero(B * Hkv * N * D * 2)
    W = min(max(window, 0), N)
    T = max(KS.SKEW, (62 + W) // 32 + 1)
    for b in range(B):
        for hk in range(Hkv):
            for kb in range((N + 255) // 256):
                if blocks is not None and (b, hk, kb) not in blocks:
                    continue
                head0 = hk * g
                qb = aq + ((b * Hq + head0) * N) * D * 2
                dob = ado + ((b * Hq + head0) * N) * D * 2
                cb = ac + ((b * Hq + head0) * 2 * N) * 4
                kbp = ak + ((b * Hkv + hk) * N) * D * 2
                vbp = av + ((b * Hkv + hk) * N) * D * 2
                dkb = adk + ((b * Hkv + hk) * N) * D * 2
                dvb = adv + ((b * Hkv + hk) * N) * D * 2
                rng = N * D * 2
                params = dict(
                    q_lo=qb & 0xFFFFFFFF, q_hi=qb >> 32, do_lo=dob & 0xFFFFFFFF, do_hi=dob >> 32,
                    c_lo=cb & 0xFFFFFFFF, c_hi=cb >> 32, k_lo=kbp & 0xFFFFFFFF, k_hi=kbp >> 32,
                    v_lo=vbp & 0xFFFFFFFF, v_hi=vbp >> 32, dk_lo=dkb & 0xFFFFFFFF, dk_hi=dkb >> 32,
                    dv_lo=dvb & 0xFFFFFFFF, dv_hi=dvb >> 32,
                    q_rng=rng, do_rng=rng, c_rng=2 * N * 4, k_rng=rng, v_rng=rng, dk_rng=rng, dv_rng=rng,
                    q_sn=D * 2, do_sn=D * 2, k_sn=D * 2, v_sn=D * 2, dk_sn=D * 2, dv_sn=D * 2,
                    q_hs=N * D * 2, do_hs=N * D * 2, c_hs=2 * N * 4,
                    nq=T, g=g, q_row0=kb * 256, kb0=kb * 256, W=W, nrows=N,
                    cdelta=N * 4, c_log2=f32_bits(scale * math.log2(math.e)), scale=f32_bits(scale))
                assert set(params) == set(KS.PARAMS), set(params) ^ set(KS.PARAMS)
                wg = Workgroup(prog, 4, mem, params, lds_bytes=160 * 1024, check_races=check_races)
                wg.run()
                if stats is not None:
                    stats.append({"block": (b, hk, kb), "icount": [w.icount for w in wg.waves], "kinds": dict(wg.waves[0].stats)})
    def back(addr):
        raw = mem.read(addr).view(np.uint16).reshape(B, Hkv, N, D)
        t = torch.from_numpy(raw.view(np.int16).copy())
        return t.view(torch.bfloat16 if dtype == "bf16" else torch.float16).float()
    return back(adk), back(adv)
