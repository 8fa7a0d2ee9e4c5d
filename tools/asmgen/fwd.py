"""Generator of the hand-placed forward kernel body (gfx950, head dim 128, bf16 / f16).

Replaces _sink_flash_attn_fwd_kernel of the reference (sink_attention/sink_flash_attention.py:93-194); same maths as
csrc/sfa_fwd_mfma.hip (swapped product, online softmax in the exp2 domain seeded with the s_aux logit, deferred
rescale), different machine mapping:

  workgroup = 4 waves; wave w = 64 query rows (two 32-row blocks rb) of ONE q head, HPW = gcd(group, 4) heads x 4 / HPW
  row groups share every 64-key K / V tile through LDS; one wave per SIMD, 512 registers: Q fragments (B operands) in 64
  accumulator registers, O^T [d, row] in 128, the row sums l in 32.
  The loop is software-pipelined by one tile; iteration i runs
      A(i+1)   S^T = K Q^T of the NEXT tile              32 MFMAs  (K row fragments from LDS feed both row blocks)
      E(i)     p = exp2(c s - m), packed in place         VALU, in the gaps of A(i+1)
      C(i)     O^T += V^T P^T and l += 1 P^T             40 MFMAs  (V^T by transposed LDS reads; the row sum is one more
                                                                    MFMA per k-step with an all-ones A operand: no VALU adds)
      M(i+1)   mask, row maximum, new reference m and the rescale factor alpha of the next tile     VALU, in the gaps of C(i)
  so the exp / pack work of a tile hides under the next tile's QK^T and the max / bookkeeping under its own PV.  The
  rescale of O and l by alpha (rare: the reference point moves only when the row maximum grows by more than 2^8) runs
  out of line at the loop head when any lane of the wave asks for it.  K / V tiles arrive by LDS-DMA three tiles ahead
  into a 4-deep ring; one s_barrier per tile.  Two copies of the body alternate (the S^T registers of tile i and i+1
  swap roles), times the mask class of the next tile (full / edge / edge with sink keys / none: last tile).
"""
from __future__ import annotations

from .core import A, Imm, Instr, M0, P, PV, Prog, Reg, S, V, VCC, fimm, imm
from .dkdv import Alloc
from .sched import finish_block, fix_hazards, insert_waits, schedule
from .worklist import STREAM, WorkList

STG_BYTES = 32768
NSTAGE = 4
LDS_BYTES = NSTAGE * STG_BYTES
NEG_INF = 0xFF800000

# ---- work-list (persistent) form, see worklist.py: one 128-byte descriptor per item in LDS behind the ring
DESC_BASE = LDS_BYTES
DESC = dict({"q_lo": 0, "q_hi": 1, "o_lo": 2, "o_hi": 3, "lse_lo": 4, "lse_hi": 5, "q0": 6, "nrows": 7, "q_rng": 8, "o_rng": 9,
             "m0_0": 12, "m0_1": 13, "m0_2": 14, "m0_3": 15}, **STREAM)
PARAMS_PK = ["q_hs", "q_sn", "o_hs", "o_sn", "k_sn", "v_sn", "ld_hs", "l0", "pos0", "W", "ns", "hpw_log2", "c_log2", "ln2", "n_items"]

PARAMS = [
    "q_lo", "q_hi", "q_hs", "q_sn", "q_rng",
    "o_lo", "o_hi", "o_hs", "o_sn", "o_rng",
    "k_lo", "k_hi", "k_sn", "k_rng", "v_lo", "v_hi", "v_sn", "v_rng",
    "lse_lo", "lse_hi", "ld_hs",
    "m0_0", "m0_1", "m0_2", "m0_3", "l0",          # initial reference (s_aux log2e or -inf) of the workgroup's heads; l0 = 1 / 0
    "q0", "nrows", "pos0", "W", "ns", "nt", "ts_hi", "tw_off",
    "hpw_log2", "c_log2", "ln2",
]


class FwdGen(WorkList):
    DESC, DESC_BASE = DESC, DESC_BASE

    def __init__(self, dtype="bf16", sched=True, vfirst=4, sfirst=None, npool=10, thr=8.0, dma_t0=500, dma_dt=180, D=128, ablate=(), lsum="mfma", kpre=True, kpre_dl=300, persist=True, trans_sched=True, dead=True, sinkfar=False):
        assert dtype in ("bf16", "f16") and D in (64, 80, 96, 128)
        self.dtype, self.do_sched, self.thr = dtype, sched, thr
        self.persist = persist
        self.trans_sched = trans_sched        # item transition placed by the gap scheduler (the pipeline fill's MFMAs beside the store tail)
        if persist:
            assert lsum == "mfma" and kpre
        if sfirst is None:
            sfirst = 44 if persist else 56
        # head dim: DK k-steps of 16, DB 32-wide output blocks, NCH valid 16-byte chunks per row (LDS rows stay 256 bytes:
        # chunks beyond the head dim are fetched as zeros or, when a whole 128-byte half is padding, not at all)
        self.D, self.DK, self.DB, self.NCH = D, D // 16, (D + 31) // 32, D // 8
        self.HALVES = 2 if D > 64 else 1
        # LDS-DMA deadlines inside a trip (early and staggered; placing them in the PV half measured 0.8 % slower)
        self.dma_t0, self.dma_dt = dma_t0, dma_dt
        self.ablate = set(ablate)         # timing-only knock-out builds (wrong results)
        self.dead = dead                  # tile class 4 (emit_class) and the iteration bodies that skip dead tiles
        self.dead_sel = True              # ... found by a selector behind the loop head that only edge tiles reach, not by every body
        # ... which can also tell sink tiles from edge tiles, so that the bodies compute "full or not" only (12 scalar
        # instructions fewer per iteration): measured no faster (C3 forward 1.484 vs 1.477 ms, same box), off
        self.range_cls = False
        # an item's sink tile without the exp2 / PV of its second key half when nobody sees it: measured no faster (C3 forward
        # 1.458 vs 1.445 ms, window 512 0.383 vs 0.383, in-process): off; the dQ kernel's version of it pays (dq.py)
        self.sinkfar = sinkfar and persist and dead
        # the first four K row fragments of the NEXT iteration's S^T chains are read at the end of the current one (their
        # latency passes under the loop head instead of in front of the first MFMA); needs tile i+2 landed at barrier i
        self.kpre, self.kpre_dl = kpre, kpre_dl
        self.lsum_valu = lsum == "valu"   # row sums: f32 adds beside the exponentials ("valu") or ones-MFMAs ("mfma")
        self.vfirst, self.sfirst = vfirst, sfirst
        va = self.va = Alloc("v", vfirst, 255)
        sa = self.sa = Alloc("s", sfirst, 99)
        # ---------------- VGPRs
        self.SS = [[[va("s%d_%d%d" % (par, kh, rb), 16, 4) for rb in range(2)] for kh in range(2)] for par in range(2)]
        self.POOL = [va("pool%d" % i, 4, 4) for i in range(npool)]
        self.m = [va("m%d" % rb) for rb in range(2)]          # reference point of the running softmax (log2 domain)
        self.nms = [va("nms%d" % rb) for rb in range(2)]      # -(m, or 0 where m = -inf): the exponent offset
        self.alpha = [va("alpha%d" % rb) for rb in range(2)]
        # per-lane partial row sums (the lane's own keys; the two half-waves are added in the epilogue)
        self.lsum = [va("lsum%d" % rb) for rb in range(2)] if self.lsum_valu else None
        self.lane, self.lane31 = va("lane"), va("lane31")
        self.l_row_e, self.l_tr0 = va("l_row_e"), va("l_tr0")
        self.a_k_e, self.a_k_o = va("a_k_e"), va("a_k_o")
        self.a_tr0, self.a_tr1 = va("a_tr0"), va("a_tr1")
        self.l_dma = [[va("l_dma%d%s" % (e, t)) for t in "kv"] for e in range(2)]
        # second 128-byte half of the rows: an out-of-range offset for lanes whose chunk lies beyond the head dim
        self.l_dma1 = [[va("l_dma1_%d%s" % (e, t)) for t in "kv"] for e in range(2)] if 64 < D < 128 else None
        self.vt = [va("vt%d" % i) for i in range(2)]
        self.v_oob = va("v_oob")
        self.v_pos = [va("v_pos%d" % rb) for rb in range(2)]
        self.v_d = [va("v_d%d" % rb) for rb in range(2)]
        self.v_w, self.v_2e31, self.v_nsh, self.v_weff, self.v_ninf = va("v_w"), va("v_2e31"), va("v_nsh"), va("v_weff"), va("v_ninf")
        self.tmp = [va("tmp%d" % i) for i in range(6)]
        self.vo = [va("vo%d" % rb) for rb in range(2)]
        # ---------------- AGPRs
        self.QF = [[A((rb * 8 + ks) * 4, 4) for ks in range(self.DK)] for rb in range(2)]
        self.LACC = [A(64 + rb * 16, 16) for rb in range(2)]
        self.ONES = A(96, 4)
        self.OACC = [[A(128 + (rb * 4 + db) * 16, 16) for db in range(self.DB)] for rb in range(2)]
        # ---------------- SGPRs
        self.d_k, self.d_v, self.d_x = sa("d_k", 4, 4), sa("d_v", 4, 4), sa("d_x", 4, 4)
        self.s_flag = sa("s_flag", 2, 2)
        self.s_f0 = sa("s_f0", 2, 2)
        self.s_wave, self.s_hh, self.s_rgi = sa("s_wave"), sa("s_hh"), sa("s_rgi")
        self.s_pw0, self.s_pwhi = sa("s_pw0"), sa("s_pwhi")
        self.s_flo, self.s_frng = sa("s_flo"), sa("s_frng")        # tiles starting in [flo, flo + frng) are "full" for this wave
        self.s_it, self.s_k0n = sa("s_it"), sa("s_k0n")
        self.s_st, self.s_stn, self.s_std = sa("s_st"), sa("s_stn"), sa("s_std")
        self.s_koff, self.s_voff = sa("s_koff"), sa("s_voff")
        self.s_wofs, self.s_cls = sa("s_wofs"), sa("s_cls")
        self.s_tmp = [sa("s_tmp%d" % i) for i in range(5)]
        if persist:
            self.wl_alloc(sa)
        self.r_nt = self.s_nt if persist else P("nt")
        self.r_ts_hi = self.s_ts_hi if persist else P("ts_hi")
        self.r_tw_off = self.s_tw_off if persist else P("tw_off")
        self.pool_next = 0

    def params(self):
        if self.persist:
            return list(PARAMS_PK)
        return list(PARAMS)

    def pool(self):
        r = self.POOL[self.pool_next % len(self.POOL)]
        self.pool_next += 1
        return r

    # ------------------------------------------------------------------ shared with dq.py (same tile ring)
    def emit_tile_of(self, p: Prog, dst, it):
        p.s_add_u32(dst, it, self.r_tw_off)
        p.s_cmp("lt_u32", it, self.r_ts_hi)
        p.s_cselect(dst, it, dst)

    def emit_dma_tile(self, p: Prog, it_reg, spread=False):
        t = self.s_tmp
        self.emit_tile_of(p, t[0], it_reg)
        p.s_lshl_b32(t[0], t[0], 6)
        p.s_mul_i32(self.s_koff, t[0], P("k_sn"))
        p.s_mul_i32(self.s_voff, t[0], P("v_sn"))
        p.s_cmp("lt_u32", it_reg, P("nt"))
        p.s_cselect(self.d_k[2], P("k_rng"), 0)
        p.s_cselect(self.d_v[2], P("v_rng"), 0)
        k = 0
        for img, desc, off, col in ((0, self.d_k, self.s_koff, 0), (16384, self.d_v, self.s_voff, 1)):
            for e in range(2):
                for half in range(self.HALVES):
                    vt = self.vt[k & 1]
                    if half and self.l_dma1 is not None:
                        p.v_add_u32(vt, off, self.l_dma1[e][col])
                    else:
                        p.v_add_u32(vt, off, self.l_dma[e][col])
                        if half:
                            p.v_add_u32(vt, 128, vt)
                    if k == 0:
                        p.s_add_u32(t[1], self.s_std, self.s_wofs)
                        p.s_mov_m0(t[1])
                    else:
                        p.s_add_m0(t[1], img + 2048 * e + 1024 * half)
                    ins = p.buffer_load_lds(16, vt, desc, 0, mem=("dma_stage",))
                    if spread:
                        ins.mods["alap"] = self.dma_t0 + self.dma_dt * k
                    k += 1

    def emit_class(self, p: Prog, k0):
        """s_cls of the tile starting at key k0: 0 full, 1 edge (no sink key in the tile), 2 edge with sink keys, 4 dead"""
        t = self.s_tmp
        if self.range_cls:
            # the scheduled bodies only ask "full or not": every key causal for every row and inside every row's window <=>
            # pwhi - W + 1 <= k0 <= pw0 - 63, one unsigned compare against two per-item constants; the selector behind the loop
            # head tells the rest apart (edge / sink keys / dead), for the tiles that are not full
            p.s_sub_u32(t[0], k0, self.s_flo)
            p.s_cmp("lt_u32", t[0], self.s_frng)
            p.s_cselect(self.s_cls, 0, 1)
            return
        p.s_add_u32(t[0], k0, 63)
        p.s_cmp("le_i32", t[0], self.s_pw0)
        p.s_cselect(t[1], 1, 0)
        p.s_cmp("lt_i32", t[0], P("ns"))
        p.s_cselect(t[2], 1, 0)
        p.s_sub_i32(t[0], self.s_pwhi, P("W"))
        p.s_add_i32(t[0], t[0], 1)
        p.s_cmp("ge_i32", k0, t[0])
        p.s_cselect(t[0], 1, 0)
        p.s_or_b32(t[0], t[0], t[2])
        p.s_and_b32(t[0], t[0], t[1])
        p.s_cmp("lt_i32", k0, P("ns"))
        p.s_cselect(t[1], 2, 1)
        p.s_cmp("lg_u32", t[0], 0)
        p.s_cselect(self.s_cls, 0, t[1])
        if self.dead and not self.dead_sel:
            # 4: no row of the wave sees any key of the tile - every key lies behind every row (k0 > pwhi), or the tile holds
            # no sink key and every key has left every row's window (k0 + 63 <= pw0 - W).  Workgroups whose waves own
            # different ROWS (MHA, groups of 2: 4 / hpw row groups) walk tiles that only the other waves' rows can see.
            p.s_cmp("gt_i32", k0, self.s_pwhi)
            p.s_cselect(t[0], 1, 0)
            p.s_sub_i32(t[1], self.s_pw0, P("W"))
            p.s_add_u32(t[2], k0, 63)
            p.s_cmp("le_i32", t[2], t[1])
            p.s_cselect(t[1], 1, 0)
            p.s_cmp("ge_i32", k0, P("ns"))
            p.s_cselect(t[2], 1, 0)
            p.s_and_b32(t[1], t[1], t[2])
            p.s_or_b32(t[0], t[0], t[1])
            p.s_cmp("lg_u32", t[0], 0)
            p.s_cselect(self.s_cls, 4, self.s_cls)

    # ------------------------------------------------------------------ phases
    def emit_k_prefetch(self, p: Prog, e, o, deadline=None):
        """first four K row fragments (key half 0, k-steps 0..3) of a tile, into pool slots 0..3"""
        for ks in range(4):      # (DK >= 4 for every supported head dim)
            ins = p.ds_read_b128(self.POOL[ks], o if ks & 1 else e, 512 * (ks >> 1), mem=("stage_r",), note="K rows, next trip")
            if deadline is not None:
                ins.mods["alap"] = deadline + 6 * ks

    # Sub-block classes of a tile (the short-window strip bodies, fwd_strip.py): self.sub = None, or sub[kh][rb] in ("mask",
    # "full", "dead") for key half kh x row block rb - a "dead" sub-block (no row sees any of its keys) is left out of every
    # phase, a "full" one (every row sees every key) gets no mask instructions.
    sub = None

    def sub_is(self, kh, rb, what):
        return self.sub is not None and self.sub[kh][rb] == what

    def emit_A(self, p: Prog, par_next: int, e, o, pre=False):
        """S^T of a tile whose K image row-read addresses are e / o: four chains of 8, into SS[par_next]; pre: pool slots
        0..3 already hold the first four fragments"""
        dt = self.dtype
        for kh in range(2):
            kf = []
            for ks in range(self.DK):
                f = self.pool()
                base = o if ks & 1 else e
                if not (pre and kh == 0 and ks < 4):
                    p.ds_read_b128(f, base, 8192 * kh + 512 * (ks >> 1), mem=("stage_r",), note="K rows")
                kf.append(f)
            for rb in range(2):
                if self.sub_is(kh, rb, "dead"):
                    continue
                acc = self.SS[par_next][kh][rb]
                for ks in range(self.DK):
                    p.mfma(dt, acc, kf[ks], self.QF[rb][ks], acc if ks else 0, tag="S")

    def emit_M(self, p: Prog, par: int, cls: int, k0):
        """mask (class cls of the tile at key k0), row maximum, new reference point and rescale factor for the tile in
        SS[par]; sets s_flag (lane mask: some row's reference moved)"""
        t = self.tmp
        for rb in range(2):
            if cls and "mask" not in self.ablate:
                p.v_lshrrev(t[1], 5, self.lane)
                p.v_lshlrev(t[1], 2, t[1])
                p.v_add_u32(t[1], k0, t[1])
                p.v_sub_u32(self.v_d[rb], self.v_pos[rb], t[1])        # pos - k0 - 4 h
                if cls == 2 and rb == 0:
                    p.v_sub_u32(self.v_nsh, P("ns"), t[1])
                if cls == 4:                                            # window only: key valid <=> c > (pos - k0 - 4 h) - W
                    p.v_sub_u32(self.v_d[rb], self.v_d[rb], self.v_w)
                for kh in range(2):
                    if self.sub_is(kh, rb, "dead") or self.sub_is(kh, rb, "full"):
                        continue
                    for v in range(16):
                        x = self.SS[par][kh][rb][v]
                        c = 32 * kh + (v & 3) + 8 * (v >> 2)
                        if cls in (3, 4):
                            # typed edges (one compare of the element's constant offset against a per-lane threshold): 3 = the
                            # tile's keys are inside every row's window, only the causal test c <= pos - k0 - 4 h; 4 = every key
                            # is causal, only the window test
                            p.v_cmp("le_i32" if cls == 3 else "gt_i32", c, self.v_d[rb])
                            p.v_cndmask(x, self.v_ninf, x)
                            continue
                        p.v_sub_u32(t[0], self.v_d[rb], c)
                        if cls == 2:
                            p.v_cmp("lt_i32", c, self.v_nsh)
                            p.v_cndmask(self.v_weff, self.v_w, self.v_2e31)
                            p.v_cmp("lt_u32", t[0], self.v_weff)
                        else:
                            p.v_cmp("lt_u32", t[0], self.v_w)
                        p.v_cndmask(x, self.v_ninf, x)
            # row maximum over the lane's 32 scores (two trees of max3), then across the two half-waves
            regs = [self.SS[par][kh][rb][v] for kh in range(2) if not self.sub_is(kh, rb, "dead") for v in range(16)]
            mx, mx2 = t[2], t[3]
            p.v_max3_f32(mx, regs[0], regs[1], regs[2])
            p.v_max3_f32(mx2, regs[3], regs[4], regs[5])
            i = 6
            while i + 3 < len(regs):
                p.v_max3_f32(mx, mx, regs[i], regs[i + 1])
                p.v_max3_f32(mx2, mx2, regs[i + 2], regs[i + 3])
                i += 4
            p.v_max3_f32(mx, mx, regs[-2], regs[-1])
            p.v_max_f32(mx, mx, mx2)
            p.v_mov(mx2, mx)
            p.v_permlane32_swap(mx, mx2, note="lanes 32..63 of mx <-> lanes 0..31 of mx2: each lane now holds both halves")
            p.v_max_f32(mx, mx, mx2)
            # m_cand = max(m, c mx); the reference moves only when it would grow by more than thr (or from -inf)
            mc, tt = t[4], t[5]
            p.v_mul_f32(mc, P("c_log2"), mx)
            p.v_max_f32(mc, mc, self.m[rb])
            p.v_add_f32(tt, fimm(self.thr), self.m[rb])
            p.v_cmp("gt_f32", mc, tt)
            p.v_cndmask(mc, self.m[rb], mc)                            # m_new
            p.v_cmp("neq_f32", self.v_ninf, mc)
            p.v_cndmask(tt, 0, mc)                                     # m_safe = m_new, 0 where m_new = -inf
            p.v_sub_f32(self.alpha[rb], self.m[rb], tt)
            p.v_exp_f32(self.alpha[rb], self.alpha[rb])
            if getattr(self, "inf_no_rescale", False):
                # a row that has seen nothing yet (m = -inf: O = 0, l = 0) needs no rescale when its reference point moves:
                # alpha = 1 there, so the lane does not ask for the out-of-line pass (short windows without s_aux: the rows
                # whose first tile is wholly outside their window would ask for it in every item)
                p.v_cmp("neq_f32", self.v_ninf, self.m[rb])
                p.v_cndmask(self.alpha[rb], fimm(1.0), self.alpha[rb])
            p.v_sub_f32(self.nms[rb], 0, tt)
            p.v_mov(self.m[rb], mc)
            p.v_cmp("neq_f32", fimm(1.0), self.alpha[rb])
            if rb == 0:
                p.s_mov_b64(self.s_f0, VCC)
            else:
                p.s_or_b64(self.s_flag, self.s_f0, VCC)

    def emit_E(self, p: Prog, par: int):
        """p = exp2(c s - m) of the tile in SS[par], packed in place: [kh][rb][4 s + j]; key half 0 of both row blocks
        first (the PV of that half starts while the second half is still being exponentiated)"""
        for kh in range(2):
            for rb in range(2):
                if self.sub_is(kh, rb, "dead"):
                    continue
                acc = self.SS[par][kh][rb]
                for v in range(16):
                    if "fma" not in self.ablate:
                        p.v_fma_f32(acc[v], acc[v], P("c_log2"), self.nms[rb])
                    p.v_exp_f32(acc[v], acc[v])
                    if self.lsum_valu:
                        p.v_add_f32(self.lsum[rb], self.lsum[rb], acc[v])
                for s in range(2):
                    for j in range(4):
                        p.v_cvt_pk(self.dtype, acc[4 * s + j], acc[8 * s + 2 * j], acc[8 * s + 2 * j + 1])

    def emit_C(self, p: Prog, par: int):
        """O^T += V^T P^T, l += 1 P^T for the tile in SS[par] (V image transposed-read addresses a_tr0 / a_tr1); every
        V^T fragment feeds both row blocks"""
        dt = self.dtype
        for kh in range(2):
            for s in range(2):
                pf = [self.SS[par][kh][rb][4 * s:4 * s + 4] for rb in range(2)]
                for rb in range(2):
                    if not self.lsum_valu and not self.sub_is(kh, rb, "dead"):
                        p.mfma(dt, self.LACC[rb], self.ONES, pf[rb], self.LACC[rb], tag="l")
                if self.sub_is(kh, 0, "dead") and self.sub_is(kh, 1, "dead"):
                    continue                                   # nobody takes this key half's V^T fragments
                for db in range(self.DB):
                    f = self.pool()
                    off = 16384 + 8192 * kh + 512 * db
                    p.ds_read_b64_tr_b16(f[0:2], self.a_tr0, off + 2048 * (2 * s), mem=("stage_r",))
                    p.ds_read_b64_tr_b16(f[2:4], self.a_tr1, off + 2048 * (2 * s + 1), mem=("stage_r",))
                    for rb in range(2):
                        if not self.sub_is(kh, rb, "dead"):
                            p.mfma(dt, self.OACC[rb][db], f, pf[rb], self.OACC[rb][db], tag="PV")

    def emit_next_class(self, p: Prog, it_next, cur_dead=False):
        """state of the trip that processes tile it_next in E / C: first key and class of the tile after it (the one
        that trip computes S^T and the maximum of), folded with the trip's parity and with whether tile it_next itself is
        dead (cur_dead: known where this is emitted) into the dispatch code (2 cur_dead + par) 8 + class (class 3: there is
        no next tile, 4: the next tile is dead)"""
        t = self.s_tmp
        p.s_add_u32(t[3], it_next, 1)
        self.emit_tile_of(p, t[4], t[3])
        p.s_lshl_b32(self.s_k0n, t[4], 6)
        self.emit_class(p, self.s_k0n)
        p.s_cmp("lt_u32", t[3], self.r_nt)
        p.s_cselect(self.s_cls, self.s_cls, 3)
        p.s_and_b32(t[0], it_next, 1)
        p.s_lshl_b32(t[0], t[0], 3)
        p.s_add_u32(self.s_cls, self.s_cls, t[0])
        if cur_dead:
            p.s_add_u32(self.s_cls, self.s_cls, 16)

    # ------------------------------------------------------------------ prologue
    def prologue(self) -> Prog:
        p = Prog()
        t0, t1, t2, t3 = self.tmp[:4]
        st = self.s_tmp
        lane, wv = self.lane, self.s_wave
        p.v_and(lane, 63, PV("tid"))
        p.v_lshrrev(t0, 6, PV("tid"))
        p.v_readfirstlane(wv, t0)
        p.v_and(self.lane31, 31, lane)
        p.v_mov(self.v_oob, imm(0x7FFFF000))
        p.s_lshl_b32(st[0], 1, P("hpw_log2"))
        p.s_sub_u32(st[0], st[0], 1)
        p.s_and_b32(self.s_hh, wv, st[0])
        p.s_lshr_b32(self.s_rgi, wv, P("hpw_log2"))
        p.v_lshrrev(t0, 3, self.lane31)
        p.v_lshlrev(t0, 11, t0)
        p.v_and(t1, 7, lane)
        p.v_lshl_add_u32(t0, t1, 6, t0)
        p.v_bfe_u32(t1, lane, 2, 2)
        p.v_lshrrev(t2, 5, lane)                              # h
        p.v_xor(t1, t1, t2)
        p.v_lshl_add_u32(self.l_row_e, t1, 4, t0)
        p.v_bfe_u32(t0, lane, 2, 2)
        p.v_lshl_add_u32(t0, t2, 2, t0)
        p.v_lshlrev(t0, 6, t0)
        p.v_bfe_u32(t1, lane, 4, 1)
        p.v_bfe_u32(t3, lane, 1, 1)
        p.v_lshl_add_u32(t1, t1, 1, t3)
        p.v_xor(t1, t1, t2)
        p.v_lshl_add_u32(t0, t1, 4, t0)
        p.v_and(t1, 1, lane)
        p.v_lshl_add_u32(self.l_tr0, t1, 3, t0)
        p.s_lshl_b32(st[0], self.s_rgi, 6)
        p.s_add_u32(st[0], st[0], P("q0"))                    # qw0
        p.s_add_u32(self.s_pw0, st[0], P("pos0"))
        p.s_add_u32(st[1], st[0], 63)
        p.s_sub_u32(st[2], P("nrows"), 1)
        p.s_min_i32(st[1], st[1], st[2])
        p.s_add_u32(self.s_pwhi, st[1], P("pos0"))
        p.v_add_u32(t0, st[0], self.lane31)                   # row, rb = 0
        p.v_add_u32(self.v_pos[0], P("pos0"), t0)
        p.v_add_u32(self.v_pos[1], 32, self.v_pos[0])
        # ---- Q fragments -> accumulator registers
        p.s_mul_i32(st[1], self.s_hh, P("q_hs"))
        p.s_mul_hi_u32(st[2], self.s_hh, P("q_hs"))
        p.s_add_u32(self.d_x[0], P("q_lo"), st[1])
        p.s_addc_u32(self.d_x[1], P("q_hi"), st[2])
        p.s_mov(self.d_x[2], P("q_rng"))
        p.s_mov(self.d_x[3], 0x00020000)
        p.v_mul_lo_u32(t1, t0, P("q_sn"))
        p.v_lshl_add_u32(self.vo[0], t2, 4, t1)
        p.s_lshl_b32(st[1], P("q_sn"), 5)
        p.v_add_u32(self.vo[1], st[1], self.vo[0])
        # ---- K / V streams (as dq.py)
        for d, nm in ((self.d_k, "k"), (self.d_v, "v")):
            p.s_mov(d[0], P(nm + "_lo"))
            p.s_mov(d[1], P(nm + "_hi"))
            p.s_mov(d[2], P(nm + "_rng"))
            p.s_mov(d[3], 0x00020000)
        rr, slot = t0, t1
        p.v_bfe_u32(rr, lane, 2, 3)
        p.v_and(slot, 3, lane)
        p.s_lshl_b32(st[0], wv, 4)
        for e in range(2):
            p.v_lshrrev(t3, 2, rr)
            p.v_add_u32(t3, 2 * e, t3)
            p.v_and(t3, 3, t3)
            p.v_xor(t3, t3, slot)
            p.v_lshl_add_u32(t3, t2, 2, t3)
            p.v_lshlrev(t3, 4, t3)
            p.s_add_u32(st[1], st[0], 8 * e)
            p.v_add_u32(self.vt[0], st[1], rr)
            for col, nm in ((0, "k"), (1, "v")):
                p.v_mul_lo_u32(self.l_dma[e][col], self.vt[0], P(nm + "_sn"))
                p.v_add_u32(self.l_dma[e][col], self.l_dma[e][col], t3)
                if self.l_dma1 is not None:      # chunk 8 + (t3 >> 4) of the row must be < NCH
                    p.v_lshrrev(self.vt[1], 4, t3)
                    p.v_add_u32(self.l_dma1[e][col], 128, self.l_dma[e][col])
                    p.v_cmp("gt_u32", self.NCH - 8, self.vt[1])
                    p.v_cndmask(self.l_dma1[e][col], self.v_oob, self.l_dma1[e][col])
        p.s_lshl_b32(self.s_wofs, wv, 12)
        # requests in the order their data is needed: tile 0, the Q fragments, tiles 1 and 2; the softmax state and the
        # constants are set up while they are in flight
        p.s_mov(self.s_std, 0)
        p.s_mov(st[3], 0)
        self.emit_dma_tile(p, st[3])
        for rb in range(2):
            for ks in range(self.DK):
                p.buffer_load(self.QF[rb][ks], self.vo[rb], self.d_x, 0, offset=32 * ks)
        for j in (1, 2):
            p.s_mov(self.s_std, j * STG_BYTES)
            p.s_mov(st[3], j)
            self.emit_dma_tile(p, st[3])
        # ---- softmax state: m = m0 of the wave's head, l = l0, O = 0 ; constants
        p.s_cmp("eq_u32", self.s_hh, 1)
        p.s_cselect(st[1], P("m0_1"), P("m0_0"))
        p.s_cmp("eq_u32", self.s_hh, 2)
        p.s_cselect(st[1], P("m0_2"), st[1])
        p.s_cmp("eq_u32", self.s_hh, 3)
        p.s_cselect(st[1], P("m0_3"), st[1])
        if self.lsum_valu:
            p.v_mov(t1, P("l0"))
            p.v_cmp("gt_u32", 32, lane)
        for rb in range(2):
            p.v_mov(self.m[rb], st[1])
            if self.lsum_valu:
                p.v_cndmask(self.lsum[rb], 0, t1)                 # l0 once per row: lanes 0..31
            else:
                for i in range(16):
                    p.v_accvgpr_write(self.LACC[rb][i], P("l0"))
            for db in range(self.DB):
                for i in range(16):
                    p.v_accvgpr_write(self.OACC[rb][db][i], 0)
        ones = 0x3F803F80 if self.dtype == "bf16" else 0x3C003C00
        p.v_mov(t1, imm(ones))
        for i in range(4):
            p.v_accvgpr_write(self.ONES[i], t1)
        p.v_mov(self.v_w, P("W"))
        p.v_mov(self.v_2e31, imm(0x80000000))
        p.v_mov(self.v_ninf, imm(NEG_INF))
        p.s_waitcnt(vmcnt=2 * 4 * self.HALVES, note="Q fragments, tile 0 landed (tiles 1, 2 in flight)")
        p.s_barrier()
        # ---- pipeline fill: S^T of tile 0 and its softmax bookkeeping (general mask path: any tile class)
        p.v_mov(self.a_k_e, self.l_row_e)
        p.v_xor(self.a_k_o, 32, self.a_k_e)
        self.pool_next = 0
        self.emit_A(p, 0, self.a_k_e, self.a_k_o)
        p.s_mov(st[3], 0)
        self.emit_tile_of(p, st[4], st[3])
        p.s_lshl_b32(self.s_k0n, st[4], 6)
        self.emit_M(p, 0, 2, self.s_k0n)
        p.s_mov(self.s_it, 0)
        self.emit_next_class(p, self.s_it)
        p.s_mov(self.s_st, 0)
        p.s_mov(self.s_stn, STG_BYTES)
        p.s_mov(self.s_std, 3 * STG_BYTES)
        if self.kpre:
            p.s_waitcnt(vmcnt=0, note="tiles 1, 2 landed (own pieces)")
            p.s_barrier()
            p.v_add_u32(self.a_k_e, self.s_stn, self.l_row_e)
            p.v_xor(self.a_k_o, 32, self.a_k_e)
            self.emit_k_prefetch(p, self.a_k_e, self.a_k_o)
        return p

    # ------------------------------------------------------------------ loop head
    def loop_top(self) -> Prog:
        """common path: exit test, rescale test, wait + barrier, dispatch (the full-tile bodies are tested first)"""
        p = Prog()
        p.label("L_top%=")
        if self.persist:
            self.emit_stream_advance(p)
        p.s_cmp("ge_u32", self.s_it, self.r_nt)
        p.s_cbranch("scc1", "L_done%=")
        p.s_cmp_lg_u64(self.s_flag, 0)
        p.s_cbranch("scc1", "L_rescale%=")
        p.label("L_top_a%=")
        if self.kpre:
            p.s_waitcnt(vmcnt=0, note="tile it+2 landed (own pieces; issued a whole iteration ago)")
            p.s_barrier()
            p.s_waitcnt(lgkmcnt=0, note="the K fragments fetched at the end of the last iteration")
        else:
            p.s_waitcnt(vmcnt=4 * self.HALVES, note="tile it+1 landed (own pieces); tile it+2 may be in flight")
            p.s_barrier()
        codes = self.body_codes()
        if self.dead and self.dead_sel:
            codes = [c for c in codes if (c & 7) != 4]          # the dead-next bodies are entered through the selectors
        for code in codes[:-1]:
            p.s_cmp("eq_u32", self.s_cls, code)
            p.s_cbranch("scc1", ("L_sel%d%%=" if (self.dead and self.dead_sel and (code & 7) == 1) else "L_body%d%%=") % code)
        last = codes[-1]
        p.s_branch(("L_sel%d%%=" if (self.dead and self.dead_sel and (last & 7) == 1) else "L_body%d%%=") % last)
        return p

    def dead_selectors(self) -> Prog:
        """behind the loop head, for iterations whose NEXT tile is an edge tile without sink keys: is it out of reach of every
        row of the wave - every key behind every row (k0n > pwhi) or past every row's window (k0n + 63 <= pw0 - W)?  Then the
        body without its S^T chains and softmax bookkeeping (class 4), else the edge body.  Waves that own different ROWS (MHA,
        groups of 2) walk tiles that only the other waves' rows can see; the scheduled bodies carry none of this test."""
        p = Prog()
        t = self.s_tmp
        for code in self.body_codes():
            if (code & 7) != 1:
                continue
            p.label("L_sel%d%%=" % code)
            if self.range_cls:
                # a tile with sink keys: all of them sinks and causal for every row - full after all; else the sink-edge body
                l_ns = "L_selns%d%%=" % code
                p.s_cmp("lt_i32", self.s_k0n, P("ns"))
                p.s_cbranch("scc0", l_ns)
                p.s_add_u32(t[2], self.s_k0n, 63)
                p.s_cmp("lt_i32", t[2], P("ns"))
                p.s_cbranch("scc0", "L_body%d%%=" % (code + 1))
                p.s_cmp("le_i32", t[2], self.s_pw0)
                p.s_cbranch("scc1", "L_body%d%%=" % (code - 1))
                p.s_branch("L_body%d%%=" % (code + 1))
                p.label(l_ns)
            p.s_cmp("gt_i32", self.s_k0n, self.s_pwhi)
            p.s_cbranch("scc1", "L_body%d%%=" % (code + 3))
            p.s_sub_i32(t[1], self.s_pw0, P("W"))
            p.s_add_u32(t[2], self.s_k0n, 63)
            p.s_cmp("le_i32", t[2], t[1])
            p.s_cbranch("scc1", "L_body%d%%=" % (code + 3))
            p.s_branch("L_body%d%%=" % code)
        return p

    def body_codes(self):
        """dispatch codes (2 cur_dead + par) 8 + class of the iteration bodies, the frequent ones first"""
        live = [8 * par + c for c in (0, 1, 3, 2) for par in (0, 1)]
        if not self.dead:
            return live
        codes = live + [8 * par + 4 for par in (0, 1)] + [16 + 8 * par + c for c in (4, 0, 1, 3, 2) for par in (0, 1)]
        if self.sinkfar:
            codes += [32 + c for c in (0, 1, 4, 3, 2)]           # the item's first iteration, tile 0 = a far sink tile (parity 0)
        return codes

    def rescale(self) -> Prog:
        """out of line: O and l of the rows whose reference point moved are multiplied by alpha (1 elsewhere)"""
        p = Prog()
        p.label("L_rescale%=")
        t = self.tmp
        k = 0
        for rb in range(2):
            regs = [self.OACC[rb][db][i] for db in range(self.DB) for i in range(16)]
            if self.lsum_valu:
                p.v_mul_f32(self.lsum[rb], self.alpha[rb], self.lsum[rb])
            else:
                regs = [self.LACC[rb][i] for i in range(16)] + regs
            for a in regs:
                r = t[k % 4]
                k += 1
                p.v_accvgpr_read(r, a)
                p.v_mul_f32(r, self.alpha[rb], r)
                p.v_accvgpr_write(a, r)
        p.s_branch("L_top_a%=")
        return p

    # ------------------------------------------------------------------ one iteration
    def body(self, par: int, cls_next: int, cur_dead: bool = False, cur_sinkfar: bool = False) -> Prog:
        """par: SS[par] holds tile i (its exp / PV run here), SS[par ^ 1] receives tile i+1; cls_next 0..2, 3 = none, 4 = tile
        i+1 is dead for this wave (no S^T, no softmax bookkeeping); cur_dead: tile i is (nothing to exponentiate or add)"""
        p = Prog()
        self.pool_next = 0
        st = self.s_tmp
        if not self.kpre:        # (with the prefetch the previous iteration left the addresses of tile i+1's K image)
            p.v_add_u32(self.a_k_e, self.s_stn, self.l_row_e)
            p.v_xor(self.a_k_o, 32, self.a_k_e)
        p.v_add_u32(self.a_tr0, self.s_st, self.l_tr0)
        p.v_xor(self.a_tr1, 32, self.a_tr0)
        if self.persist:         # the next tile of the K / V stream (it may belong to the next item)
            self.emit_dma_stream_tile(p, spread=True)
        else:
            p.s_add_u32(st[4], self.s_it, 3)
            self.emit_dma_tile(p, st[4], spread=True)
        if cls_next not in (3, 4):
            self.emit_A(p, par ^ 1, self.a_k_e, self.a_k_o, pre=self.kpre)
        if not cur_dead:
            self.sub = [["mask", "mask"], ["dead", "dead"]] if cur_sinkfar else None      # (sub[kh][rb])
            self.emit_E(p, par)
            self.emit_C(p, par)
            self.sub = None
        if cls_next not in (3, 4):
            self.emit_M(p, par ^ 1, cls_next, self.s_k0n)
        else:
            p.s_mov_b64(self.s_flag, 0)
        p.s_add_u32(self.s_it, self.s_it, 1)
        t0 = st[3]
        p.s_mov(self.s_st, self.s_stn)
        p.s_add_u32(t0, self.s_stn, STG_BYTES)
        p.s_and_b32(self.s_stn, t0, LDS_BYTES - 1)
        p.s_add_u32(t0, self.s_std, STG_BYTES)
        p.s_and_b32(self.s_std, t0, LDS_BYTES - 1)
        self.emit_next_class(p, self.s_it, cur_dead=(cls_next == 4))
        if self.kpre:            # K image of tile i+2 (landed before this iteration's barrier): addresses + first fragments
            p.v_add_u32(self.a_k_e, self.s_stn, self.l_row_e)
            p.v_xor(self.a_k_o, 32, self.a_k_e)
            self.emit_k_prefetch(p, self.a_k_e, self.a_k_o, deadline=(8 * self.DK + 8 * self.DB + 8) * 32 - self.kpre_dl)
        if self.ablate:          # knock-out builds for tools/ab.py (wrong results): what does an iteration cost without ...
            kill = lambda it: (("dma" in self.ablate and it.kind == "dma") or ("exp" in self.ablate and it.kind == "trans") or
                               ("ldsr" in self.ablate and it.kind == "ds_read") or
                               ("mfma_pv" in self.ablate and it.kind == "mfma" and it.tag in ("PV", "l")) or
                               ("mfma_s" in self.ablate and it.kind == "mfma" and it.tag == "S"))
            p.items = [it for it in p.items if not kill(it)]
        return p

    # ------------------------------------------------------------------ epilogue
    def epilogue(self) -> Prog:
        p = Prog()
        dt = self.dtype
        t0, t1, t2, t3, t4, t5 = self.tmp
        st = self.s_tmp
        p.label("L_done%=")
        p.s_waitcnt(vmcnt=0, lgkmcnt=0)
        # O[row, d] = O^T[d, row] / l (l = 0 -> 1), LSE = ln2 (m + log2 l)
        p.s_mul_i32(st[1], self.s_hh, P("o_hs"))
        p.s_mul_hi_u32(st[2], self.s_hh, P("o_hs"))
        p.s_add_u32(self.d_x[0], P("o_lo"), st[1])
        p.s_addc_u32(self.d_x[1], P("o_hi"), st[2])
        p.s_mov(self.d_x[2], P("o_rng"))
        p.s_mov(self.d_x[3], 0x00020000)
        p.v_sub_u32(t0, self.v_pos[0], P("pos0"))             # row, rb = 0
        p.v_mul_lo_u32(t1, t0, P("o_sn"))
        p.v_lshrrev(t2, 5, self.lane)
        p.v_lshl_add_u32(self.vo[0], t2, 4, t1)
        p.s_lshl_b32(st[1], P("o_sn"), 5)
        p.v_add_u32(self.vo[1], st[1], self.vo[0])
        inv = [self.alpha[0], self.alpha[1]]
        lg = [self.nms[0], self.nms[1]]
        for rb in range(2):
            if self.lsum_valu:
                p.v_mov(t3, self.lsum[rb])
                p.v_mov(t4, t3)
                p.v_permlane32_swap(t3, t4)
                p.v_add_f32(t3, t3, t4)
            else:
                p.v_accvgpr_read(t3, self.LACC[rb][0])
            p.v_mov(t4, fimm(1.0))
            p.v_cmp("eq_f32", 0, t3)
            p.v_cndmask(t3, t3, t4)                           # l = 0 -> 1
            p.v_rcp_f32(inv[rb], t3)
            p.v_log_f32(lg[rb], t3)
            p.v_add_f32(lg[rb], self.m[rb], lg[rb])
            p.v_mul_f32(lg[rb], P("ln2"), lg[rb])
        # a lane holds columns 8k .. 8k+3 (half-wave 0) or 8k+4 .. 8k+7 (half-wave 1) of its row for every column group
        # k: one half exchange per dword between groups k and k+1 leaves 16 contiguous bytes in every lane (lanes 0..31
        # columns 8k .. 8k+7, lanes 32..63 columns 8k+8 .. 8k+15): half as many store instructions for the same bytes
        # (the store tail is issue-bound: T21 of the CDNA guide)
        npair = 0
        for rb in range(2):
            for db in range(self.DB):
                for gp in range(2):
                    if 32 * db + 16 * gp >= self.D:
                        continue                              # padding columns of the last block
                    X, Y = self.POOL[(2 * npair) % 8], self.POOL[(2 * npair + 1) % 8]
                    npair += 1
                    for e in range(4):
                        p.v_accvgpr_read(X[e], self.OACC[rb][db][8 * gp + e])
                        p.v_accvgpr_read(Y[e], self.OACC[rb][db][8 * gp + 4 + e])
                    for e in range(4):
                        p.v_mul_f32(X[e], inv[rb], X[e])
                        p.v_mul_f32(Y[e], inv[rb], Y[e])
                    p.v_cvt_pk(dt, X[0], X[0], X[1])
                    p.v_cvt_pk(dt, X[1], X[2], X[3])
                    p.v_cvt_pk(dt, X[2], Y[0], Y[1])
                    p.v_cvt_pk(dt, X[3], Y[2], Y[3])
                    p.v_permlane32_swap(X[0], X[2])
                    p.v_permlane32_swap(X[1], X[3])
                    p.buffer_store(X[0:4], self.vo[rb], self.d_x, 0, offset=64 * db + 32 * gp)
        # LSE [head, row] f32: lanes 0..31 (h = 0) of each row block; rows >= nrows fall outside the descriptor
        p.s_mul_i32(st[1], self.s_hh, P("ld_hs"))
        p.s_mul_hi_u32(st[2], self.s_hh, P("ld_hs"))
        p.s_add_u32(self.d_x[0], P("lse_lo"), st[1])
        p.s_addc_u32(self.d_x[1], P("lse_hi"), st[2])
        p.s_lshl_b32(self.d_x[2], P("nrows"), 2)
        p.v_lshlrev(t1, 2, t0)
        p.v_mov(t3, imm(0x7FFFFFF0))
        p.v_cmp("gt_u32", 32, self.lane)
        p.v_cndmask(t1, t3, t1)                               # lanes >= 32: out of range
        p.v_add_u32(t5, 128, t1)
        p.buffer_store(lg[0], t1, self.d_x, 0)
        p.buffer_store(lg[1], t5, self.d_x, 0)
        p.s_waitcnt(vmcnt=0)
        return p


    # ================================================================== work-list (persistent) form
    def setup_pk(self) -> Prog:
        """once per workgroup: lane constants, wave -> (head, row group), K / V stream offsets, constants"""
        p = Prog()
        t0, t1, t2, t3 = self.tmp[:4]
        st = self.s_tmp
        lane, wv = self.lane, self.s_wave
        p.v_and(lane, 63, PV("tid"))
        p.v_lshrrev(t0, 6, PV("tid"))
        p.v_readfirstlane(wv, t0)
        p.v_and(self.lane31, 31, lane)
        p.v_mov(self.v_oob, imm(0x7FFFF000))
        p.s_lshl_b32(st[0], 1, P("hpw_log2"))
        p.s_sub_u32(st[0], st[0], 1)
        p.s_and_b32(self.s_hh, wv, st[0])
        p.s_lshr_b32(self.s_rgi, wv, P("hpw_log2"))
        p.v_lshrrev(t0, 3, self.lane31)
        p.v_lshlrev(t0, 11, t0)
        p.v_and(t1, 7, lane)
        p.v_lshl_add_u32(t0, t1, 6, t0)
        p.v_bfe_u32(t1, lane, 2, 2)
        p.v_lshrrev(t2, 5, lane)                              # h
        p.v_xor(t1, t1, t2)
        p.v_lshl_add_u32(self.l_row_e, t1, 4, t0)
        p.v_bfe_u32(t0, lane, 2, 2)
        p.v_lshl_add_u32(t0, t2, 2, t0)
        p.v_lshlrev(t0, 6, t0)
        p.v_bfe_u32(t1, lane, 4, 1)
        p.v_bfe_u32(t3, lane, 1, 1)
        p.v_lshl_add_u32(t1, t1, 1, t3)
        p.v_xor(t1, t1, t2)
        p.v_lshl_add_u32(t0, t1, 4, t0)
        p.v_and(t1, 1, lane)
        p.v_lshl_add_u32(self.l_tr0, t1, 3, t0)
        rr, slot = t0, t1
        p.v_bfe_u32(rr, lane, 2, 3)
        p.v_and(slot, 3, lane)
        p.s_lshl_b32(st[0], wv, 4)
        for e in range(2):
            p.v_lshrrev(t3, 2, rr)
            p.v_add_u32(t3, 2 * e, t3)
            p.v_and(t3, 3, t3)
            p.v_xor(t3, t3, slot)
            p.v_lshl_add_u32(t3, t2, 2, t3)
            p.v_lshlrev(t3, 4, t3)
            p.s_add_u32(st[1], st[0], 8 * e)
            p.v_add_u32(self.vt[0], st[1], rr)
            for col, nm in ((0, "k"), (1, "v")):
                p.v_mul_lo_u32(self.l_dma[e][col], self.vt[0], P(nm + "_sn"))
                p.v_add_u32(self.l_dma[e][col], self.l_dma[e][col], t3)
                if self.l_dma1 is not None:
                    p.v_lshrrev(self.vt[1], 4, t3)
                    p.v_add_u32(self.l_dma1[e][col], 128, self.l_dma[e][col])
                    p.v_cmp("gt_u32", self.NCH - 8, self.vt[1])
                    p.v_cndmask(self.l_dma1[e][col], self.v_oob, self.l_dma1[e][col])
        p.s_lshl_b32(self.s_wofs, wv, 12)
        ones = 0x3F803F80 if self.dtype == "bf16" else 0x3C003C00
        p.v_mov(t1, imm(ones))
        for i in range(4):
            p.v_accvgpr_write(self.ONES[i], t1)
        p.v_mov(self.v_w, P("W"))
        p.v_mov(self.v_2e31, imm(0x80000000))
        p.v_mov(self.v_ninf, imm(NEG_INF))
        return p

    def land_new(self):
        """scratch VGPR quads for the descriptor groups of the item being opened (S^T tile registers of parity 1: dead
        between two items; the pipeline fill of the new item writes parity 0)"""
        regs = [self.SS[1][0][rb][4 * i:4 * i + 4] for rb in range(2) for i in range(4)]
        return {g: regs[g] for g in range(8)}

    def land_old(self):
        regs = [self.SS[1][1][0][4 * i:4 * i + 4] for i in range(4)]
        return {g: regs[g] for g in range(4)}

    def emit_item_begin(self, p: Prog):
        """open item s_item: tile-list scalars, row positions, the requests for its Q fragments (into registers that are
        dead once the previous item's last S^T chains have run)"""
        land = self.land_new()
        t0, t1, t2, t3 = self.tmp[:4]
        st = self.s_tmp
        self.desc_read(p, self.s_item, (0, 1, 2, 3, 5, 6), land, t0)
        self.desc_get(p, self.s_nt, land, "nt")
        self.desc_get(p, self.s_ts_hi, land, "ts_hi")
        self.desc_get(p, self.s_tw_off, land, "tw_off")
        self.desc_get(p, st[3], land, "q0")
        self.desc_get(p, st[4], land, "nrows")
        p.s_lshl_b32(st[0], self.s_rgi, 6)
        p.s_add_u32(st[0], st[0], st[3])                      # qw0 = q0 + 64 rgi
        p.s_add_u32(self.s_pw0, st[0], P("pos0"))
        p.s_add_u32(st[1], st[0], 63)
        p.s_sub_u32(st[2], st[4], 1)
        p.s_min_i32(st[1], st[1], st[2])
        p.s_add_u32(self.s_pwhi, st[1], P("pos0"))
        if self.range_cls:
            p.s_sub_i32(self.s_flo, self.s_pwhi, P("W"))
            p.s_add_i32(self.s_flo, self.s_flo, 1)
            p.s_sub_i32(st[1], self.s_pw0, self.s_flo)
            p.s_sub_i32(st[1], st[1], 62)
            p.s_max_i32(self.s_frng, st[1], 0)
        p.v_add_u32(t0, st[0], self.lane31)                   # row, rb = 0
        p.v_add_u32(self.v_pos[0], P("pos0"), t0)
        p.v_add_u32(self.v_pos[1], 32, self.v_pos[0])
        p.v_lshrrev(t2, 5, self.lane)                         # h
        vl = [self.a_tr0, self.a_tr1]                         # load offsets (the bodies recompute these registers)
        self.desc_get(p, self.d_y[0], land, "q_lo")
        self.desc_get(p, self.d_y[1], land, "q_hi")
        p.s_mul_i32(st[1], self.s_hh, P("q_hs"))
        p.s_mul_hi_u32(st[2], self.s_hh, P("q_hs"))
        p.s_add_u32(self.d_y[0], self.d_y[0], st[1])
        p.s_addc_u32(self.d_y[1], self.d_y[1], st[2])
        self.desc_get(p, self.d_y[2], land, "q_rng")
        p.s_mov(self.d_y[3], 0x00020000)
        p.v_mul_lo_u32(t1, t0, P("q_sn"))
        p.v_lshl_add_u32(vl[0], t2, 4, t1)
        p.s_lshl_b32(st[1], P("q_sn"), 5)
        p.v_add_u32(vl[1], st[1], vl[0])
        for rb in range(2):
            for ks in range(self.DK):
                p.buffer_load(self.QF[rb][ks], vl[rb], self.d_y, 0, offset=32 * ks)

    def emit_item_init(self, p: Prog, first=False):
        """softmax state of the new item (m = the head's s_aux logit or -inf, l = l0, O = 0), then the pipeline fill: S^T
        of its tile 0 (ring slot s_st: landed and barrier-visible; waits for the Q fragments) and that tile's softmax
        bookkeeping, the first K fragments of its tile 1"""
        land = self.land_new()
        st = self.s_tmp
        for i in range(4):
            self.desc_get(p, st[i], land, "m0_%d" % i)
        p.s_cmp("eq_u32", self.s_hh, 1)
        p.s_cselect(st[0], st[1], st[0])
        p.s_cmp("eq_u32", self.s_hh, 2)
        p.s_cselect(st[0], st[2], st[0])
        p.s_cmp("eq_u32", self.s_hh, 3)
        p.s_cselect(st[0], st[3], st[0])
        for rb in range(2):
            p.v_mov(self.m[rb], st[0])
            for i in range(16):
                p.v_accvgpr_write(self.LACC[rb][i], P("l0"))
            for db in range(self.DB):
                for i in range(16):
                    p.v_accvgpr_write(self.OACC[rb][db][i], 0)
        p.v_add_u32(self.a_k_e, self.s_st, self.l_row_e)
        p.v_xor(self.a_k_o, 32, self.a_k_e)
        self.pool_next = 0
        self.emit_A(p, 0, self.a_k_e, self.a_k_o)
        p.s_mov(st[3], 0)
        self.emit_tile_of(p, st[4], st[3])
        p.s_lshl_b32(self.s_k0n, st[4], 6)
        self.emit_M(p, 0, 2, self.s_k0n)
        p.s_mov(self.s_it, 0)
        self.emit_next_class(p, self.s_it)
        if self.sinkfar:
            # tile 0 is a sink tile whose second 32-key half holds no sink key and lies outside every row's window (rows far
            # behind the sinks: every item but the first few): the first iteration skips that half's exp2 / PV / row sums
            # (dispatch code + 32; its S^T and mask above were computed in full - they are masked to -inf)
            p.s_mov(st[3], 0)
            self.emit_tile_of(p, st[4], st[3])
            p.s_lshl_b32(st[4], st[4], 6)                          # k0 of tile 0
            p.s_cmp("lt_i32", st[4], P("ns"))
            p.s_cselect(st[0], 1, 0)
            p.s_add_u32(st[1], st[4], 32)
            p.s_cmp("ge_i32", st[1], P("ns"))
            p.s_cselect(st[1], 1, 0)
            p.s_and_b32(st[0], st[0], st[1])
            p.s_sub_i32(st[1], self.s_pw0, P("W"))
            p.s_add_u32(st[2], st[4], 63)
            p.s_cmp("le_i32", st[2], st[1])
            p.s_cselect(st[1], 1, 0)
            p.s_and_b32(st[0], st[0], st[1])
            p.s_lshl_b32(st[0], st[0], 5)
            p.s_add_u32(self.s_cls, self.s_cls, st[0])
        if first:
            p.s_waitcnt(vmcnt=0, note="tiles 1, 2 landed (own pieces)")
            p.s_barrier()
        p.v_add_u32(self.a_k_e, self.s_stn, self.l_row_e)
        p.v_xor(self.a_k_o, 32, self.a_k_e)
        self.emit_k_prefetch(p, self.a_k_e, self.a_k_o)

    def prologue_pk(self) -> Prog:
        p = self.setup_pk()
        p.s_mov(self.s_st, 0)
        p.s_mov(self.s_stn, STG_BYTES)
        self.emit_stream_open(p)
        p.s_mov(self.s_std, 0)
        self.emit_dma_stream_tile(p)
        self.emit_stream_advance(p)
        p.s_mov(self.s_item, 0)
        self.emit_item_begin(p)
        for j in (1, 2):
            p.s_mov(self.s_std, j * STG_BYTES)
            self.emit_dma_stream_tile(p)
            self.emit_stream_advance(p)
        p.s_mov(self.s_std, 3 * STG_BYTES)
        p.s_waitcnt(vmcnt=2 * 4 * self.HALVES, note="Q fragments, tile 0 landed (two tiles in flight)")
        p.s_barrier()
        self.emit_item_init(p, first=True)
        return p

    def emit_store_setup(self, p: Prog):
        """the finished item's O descriptor (d_x), row offsets (vo) and LSE offsets (v_d), before its row positions are
        replaced; its descriptor groups stay in land_old() for the LSE stores"""
        land = self.land_old()
        t0, t1, t2, t3 = self.tmp[:4]
        st = self.s_tmp
        self.desc_read(p, self.s_item, (0, 1, 2), land, t0)
        self.desc_get(p, self.d_x[0], land, "o_lo")
        self.desc_get(p, self.d_x[1], land, "o_hi")
        p.s_mul_i32(st[1], self.s_hh, P("o_hs"))
        p.s_mul_hi_u32(st[2], self.s_hh, P("o_hs"))
        p.s_add_u32(self.d_x[0], self.d_x[0], st[1])
        p.s_addc_u32(self.d_x[1], self.d_x[1], st[2])
        self.desc_get(p, self.d_x[2], land, "o_rng")
        p.s_mov(self.d_x[3], 0x00020000)
        p.v_sub_u32(t0, self.v_pos[0], P("pos0"))             # row, rb = 0
        p.v_mul_lo_u32(t1, t0, P("o_sn"))
        p.v_lshrrev(t2, 5, self.lane)
        p.v_lshl_add_u32(self.vo[0], t2, 4, t1)
        p.s_lshl_b32(st[1], P("o_sn"), 5)
        p.v_add_u32(self.vo[1], st[1], self.vo[0])
        # LSE [head, row] f32: lanes 0..31 of each row block (the others: an out-of-range offset)
        p.v_lshlrev(t1, 2, t0)
        p.v_mov(t3, imm(0x7FFFFFF0))
        p.v_cmp("gt_u32", 32, self.lane)
        p.v_cndmask(self.v_d[0], t3, t1)
        p.v_add_u32(self.v_d[1], 128, self.v_d[0])

    def emit_stores(self, p: Prog):
        """O[row, d] = O^T[d, row] / l (l = 0 -> 1), LSE = ln2 (m + log2 l) of the finished item (as the one-item epilogue)"""
        dt = self.dtype
        t0, t1, t2, t3, t4, t5 = self.tmp
        st = self.s_tmp
        land = self.land_old()
        inv = [self.alpha[0], self.alpha[1]]
        lg = [self.nms[0], self.nms[1]]
        for rb in range(2):
            p.v_accvgpr_read(t3, self.LACC[rb][0])
            p.v_mov(t4, fimm(1.0))
            p.v_cmp("eq_f32", 0, t3)
            p.v_cndmask(t3, t3, t4)
            p.v_rcp_f32(inv[rb], t3)
            p.v_log_f32(lg[rb], t3)
            p.v_add_f32(lg[rb], self.m[rb], lg[rb])
            p.v_mul_f32(lg[rb], P("ln2"), lg[rb])
        npair = 0
        for rb in range(2):
            for db in range(self.DB):
                for gp in range(2):
                    if 32 * db + 16 * gp >= self.D:
                        continue
                    X, Y = self.POOL[(2 * npair) % 8], self.POOL[(2 * npair + 1) % 8]
                    npair += 1
                    for e in range(4):
                        p.v_accvgpr_read(X[e], self.OACC[rb][db][8 * gp + e])
                        p.v_accvgpr_read(Y[e], self.OACC[rb][db][8 * gp + 4 + e])
                    for e in range(4):
                        p.v_mul_f32(X[e], inv[rb], X[e])
                        p.v_mul_f32(Y[e], inv[rb], Y[e])
                    p.v_cvt_pk(dt, X[0], X[0], X[1])
                    p.v_cvt_pk(dt, X[1], X[2], X[3])
                    p.v_cvt_pk(dt, X[2], Y[0], Y[1])
                    p.v_cvt_pk(dt, X[3], Y[2], Y[3])
                    p.v_permlane32_swap(X[0], X[2])
                    p.v_permlane32_swap(X[1], X[3])
                    p.buffer_store(X[0:4], self.vo[rb], self.d_x, 0, offset=64 * db + 32 * gp)
        self.desc_get(p, self.d_x[0], land, "lse_lo")
        self.desc_get(p, self.d_x[1], land, "lse_hi")
        p.s_mul_i32(st[1], self.s_hh, P("ld_hs"))
        p.s_mul_hi_u32(st[2], self.s_hh, P("ld_hs"))
        p.s_add_u32(self.d_x[0], self.d_x[0], st[1])
        p.s_addc_u32(self.d_x[1], self.d_x[1], st[2])
        self.desc_get(p, st[0], land, "nrows")
        p.s_lshl_b32(self.d_x[2], st[0], 2)
        p.buffer_store(lg[0], self.v_d[0], self.d_x, 0)
        p.buffer_store(lg[1], self.v_d[1], self.d_x, 0)

    def build_pk(self):
        items = []
        items += finish_block(self.prologue_pk().items)
        items += insert_waits(self.loop_top().items)
        for code in self.body_codes():
            if True:
                b = self.body((code >> 3) & 1, code & 7, cur_dead=bool(code & 16), cur_sinkfar=bool(code & 32)).items
                items.append(Instr("label", mods={"label": "L_body%d%%=" % code}, kind="label", cost=0))
                if self.do_sched:
                    b = schedule(b)
                b = insert_waits(b)
                b = fix_hazards(b, loop=True)
                items += b
                items.append(Instr("s_branch", mods={"label": "L_top%="}, kind="branch"))
        items += finish_block(self.rescale().items)
        if self.dead and self.dead_sel:
            items += finish_block(self.dead_selectors().items)
        # ---- item transition: the next item's Q requests go out BEFORE the finished item's stores are formed
        p = Prog()
        p.label("L_done%=")
        p.s_waitcnt(lgkmcnt=0, note="the K fragments fetched at the end of the last iteration (re-read below)")
        self.emit_store_setup(p)
        p.s_add_u32(self.s_item, self.s_item, 1)
        p.s_cmp("ge_u32", self.s_item, P("n_items"))
        p.s_cbranch("scc1", "L_last%=")
        items += finish_block(p.items)
        p = Prog()
        self.emit_item_begin(p)
        self.emit_stores(p)
        self.emit_item_init(p)
        blk = schedule(p.items) if (self.trans_sched and self.do_sched) else p.items
        items += fix_hazards(insert_waits(blk, strict_tail=True))
        items.append(Instr("s_branch", mods={"label": "L_top%="}, kind="branch"))
        p = Prog()
        p.label("L_last%=")
        self.emit_stores(p)
        p.s_waitcnt(vmcnt=0)
        items += finish_block(p.items)
        return items

    def build(self):
        if self.persist:
            return self.build_pk()
        items = []
        items += finish_block(self.prologue().items)
        items += insert_waits(self.loop_top().items)
        for code in self.body_codes():
            if True:
                b = self.body((code >> 3) & 1, code & 7, cur_dead=bool(code & 16), cur_sinkfar=bool(code & 32)).items
                items.append(Instr("label", mods={"label": "L_body%d%%=" % code}, kind="label", cost=0))
                if self.do_sched:
                    b = schedule(b)
                b = insert_waits(b)
                b = fix_hazards(b, loop=True)
                # every load into a register is consumed inside the body: nothing outstanding at its end
                items += b
                if not self.kpre:
                    items.append(Instr("s_waitcnt", kind="wait", mods={"lgkmcnt": 0}))
                items.append(Instr("s_branch", mods={"label": "L_top%="}, kind="branch"))
        items += finish_block(self.rescale().items)
        if self.dead and self.dead_sel:
            items += finish_block(self.dead_selectors().items)
        items += finish_block(self.epilogue().items)
        return items

    def clobbers(self):
        c = ["v%d" % i for i in range(self.vfirst, 256)] + ["a%d" % i for i in range(256)]
        c += ["s%d" % i for i in range(self.sfirst, 100)] + ["vcc", "scc", "memory"]
        return c
