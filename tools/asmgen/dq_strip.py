"""Generator of the hand-placed SHORT-WINDOW dQ backward kernel body (gfx950, head dims 64 / 80, bf16 / f16).

Backward twin of fwd_strip.py (the gpt-oss sliding layers / BASELINE C4: no sink keys, window <= 193, GQA group a multiple
of 4): a workgroup (4 waves = 4 q heads x 64 rows) walks a STRIP of consecutive 64-row query tiles of one (batch, KV head,
head set); the K / V tiles slide through an LDS ring of NT + 1 slots (one new tile per item, one barrier per item); the Q and
dO fragments and the row constants of the NEXT item are requested while the current one computes (double buffered: head dims
64 / 80 leave the registers for it, 96 does not); the finished item's dQ is scaled, packed and stored in the gaps of the next
item's first MFMAs (its accumulators are free again when the next item's first dQ MFMA - srcC = 0 - writes them).

Maths per tile as dq.py (S^T = K Q^T, dP^T = V dO^T, P = exp2(c S - LSE log2e), dS = P (dP - Delta), dQ^T += K^T dS^T); an item's
NT tiles are unrolled into ONE block placed around its 64 NT (head dim 80) MFMAs by sched.py.  Two block sets as in the forward:
"steady" (typed masks: first tile window test, diagonal tile causal test, the tiles between unmasked; W = 64 (NT - 1) or one
more, every tile of the item exists) and "general" (one compare against a per-tile threshold: W, or 0 for tile indices below 0).
"""
from __future__ import annotations

from .core import A, Imm, Instr, P, PV, Prog, imm
from .dq import STG_BYTES, DqGen
from .sched import finish_block, fix_hazards, insert_waits, schedule

PARAMS = [
    "q_lo", "q_hi", "q_hs", "q_sn", "q_rng",
    "do_lo", "do_hi", "do_hs", "do_sn", "do_rng",
    "dq_lo", "dq_hi", "dq_hs", "dq_sn", "dq_rng",
    "k_lo", "k_hi", "k_sn", "k_rng", "v_lo", "v_hi", "v_sn", "v_rng",
    "lse_lo", "lse_hi", "dl_lo", "dl_hi", "ld_hs",
    "q0", "n_it", "nrows", "W", "c_log2", "nlog2e", "scale",
]


class DqStripGen(DqGen):
    def __init__(self, dtype="bf16", D=80, NT=3, sched=True, vfirst=4, sfirst=48, npool=12, ld_step=None, dma_from=None, st_from=None, st_step=None):
        assert D in (64, 80) and 2 <= NT <= 4
        # deadlines (cycles into the item's block) of the next item's loads, its tile's DMA pieces and the finished item's stores:
        # spread over the whole item, loads first, stores last
        # The finished item's stores come FIRST (its accumulators must be free when the first tile's dQ MFMAs start), then the
        # next item's loads, its tile's DMA pieces last; everything has landed at the next item's head (vmcnt(0)).
        # (positions are MFMA indices of the item's block)
        nm = NT * (4 * (D // 16) * 2 + 8 * ((D + 31) // 32))      # MFMAs of the block: S, dP chains + dQ per tile ...
        nm = nm * (4 * NT - 2) // (4 * NT)                        # ... less two dead sub-blocks per (steady) item, tile_sub()
        self.st_from = st_from if st_from is not None else 2
        self.st_step = st_step if st_step is not None else max(1, int(0.22 * nm / 10))
        self.ld_from = int(0.27 * nm)
        self.ld_step = ld_step if ld_step is not None else max(1, int(0.60 * nm / (4 * (D // 16) + 4)))
        self.dma_from = dma_from if dma_from is not None else int(0.90 * nm)
        DqGen.__init__(self, dtype, sched=sched, vfirst=vfirst, sfirst=sfirst, npool=npool, D=D, persist=False)
        self.NT, self.R = NT, NT + 1
        self.RING = self.R * STG_BYTES
        va, sa = self.va, self.sa
        DK, DB = self.DK, self.DB
        # ---------------- AGPRs: two buffers of Q and dO fragments, dQ^T
        self.QFB = [[[A(((b * 4 + rb) * DK + ks) * 4, 4) for ks in range(DK)] for rb in range(2)] for b in range(2)]
        self.DOFB = [[[A(((b * 4 + 2 + rb) * DK + ks) * 4, 4) for ks in range(DK)] for rb in range(2)] for b in range(2)]
        base = 2 * 4 * DK * 4
        self.DQ = [[A(base + (rb * DB + db) * 16, 16) for db in range(DB)] for rb in range(2)]
        assert base + 2 * DB * 16 <= 256
        # ---------------- VGPRs
        # (registers of the one-item body that this walk does not use are taken over: lse2 / nd as buffer 0, v_w / v_2e31 ...)
        self.lse2b = [self.lse2, [va("lse2_1%d" % rb) for rb in range(2)]]
        self.ndb = [self.nd, [va("nd_1%d" % rb) for rb in range(2)]]
        self.v_wt, self.v_wc = self.v_w, self.v_2e31
        self.vl = [[va("vl%s%d" % (t, rb)) for rb in range(2)] for t in "qd"]      # Q / dO load offsets (next item)
        self.v_lc = [self.v_nsh, self.v_weff]                                      # row-constant load offsets (next item)
        # two sets of LDS address registers and mask bases: consecutive tiles of an item alternate between them
        self.addr = [[self.a_k_e, self.a_k_o, self.a_v_e, self.a_v_o, self.a_tr0, self.a_tr1],
                     [self.a_kn_e, self.a_kn_o] + [va("ad1_%d" % i) for i in range(4)]]
        self.v_dj = [self.v_d, [va("vd1_%d" % rb) for rb in range(2)]]
        self.ep = [va("ep%d" % i, 4, 4) for i in range(4)]
        spare = [self.s_it, self.s_k0, self.s_st, self.s_stn, self.s_std, self.s_cls, self.s_rgi, self.s_pw0, self.s_pwhi]
        take = lambda name: spare.pop() if spare else sa(name)
        self.s_n, self.s_T, self.s_s0 = take("s_n"), take("s_T"), take("s_s0")
        self.s_slot = [take("s_slot%d" % j) for j in range(NT)]
        self.s_k0t = [take("s_k0_%d" % j) for j in range(NT)]
        self.s_par = take("s_par")
        self.d_q, self.d_do = sa("d_q", 4, 4), sa("d_do", 4, 4)
        self.d_l, self.d_d = sa("d_l", 4, 4), sa("d_d", 4, 4)
        self.steady = False

    def params(self):
        return list(PARAMS)

    @staticmethod
    def gate(ins, k):
        """not before the k-th MFMA of the block, and right there"""
        ins.mods["after_mfma"] = int(k)
        ins.mods["alap"] = 32 * (int(k) + 1)

    # ------------------------------------------------------------------ small pieces (as fwd_strip.py)
    def emit_wrap(self, p: Prog, dst, src, add):
        p.s_add_u32(dst, src, add)
        p.s_cmp("ge_u32", dst, self.RING)
        p.s_cselect(self.s_tmp[4], self.RING, 0)
        p.s_sub_u32(dst, dst, self.s_tmp[4])

    def emit_item_scalars(self, p: Prog):
        for j in range(self.NT):
            if j == 0:
                p.s_mov(self.s_slot[0], self.s_s0)
            else:
                self.emit_wrap(p, self.s_slot[j], self.s_slot[j - 1], STG_BYTES)
            p.s_add_i32(self.s_tmp[3], self.s_T, j)
            p.s_lshl_b32(self.s_k0t[j], self.s_tmp[3], 6)

    def emit_dma(self, p: Prog, tile, slot, spread_from=None):
        t = self.s_tmp
        p.s_lshl_b32(t[0], tile, 6)
        p.s_mul_i32(self.s_koff, t[0], P("k_sn"))
        p.s_mul_i32(self.s_voff, t[0], P("v_sn"))
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
                        p.s_add_u32(t[1], slot, self.s_wofs)
                        p.s_mov_m0(t[1])
                    else:
                        p.s_add_m0(t[1], img + 2048 * e + 1024 * half)
                    ins = p.buffer_load_lds(16, vt, desc, 0, mem=("dma_stage",))
                    if spread_from is not None:
                        self.gate(ins, spread_from + 2 * k)
                    k += 1

    def emit_loads(self, p: Prog, buf, spread_from=None):
        """Q / dO fragments and row constants of an item into buffer `buf` (offsets vl / v_lc point at its rows)"""
        n = 0
        for frags, vl, d in ((self.QFB[buf], self.vl[0], self.d_q), (self.DOFB[buf], self.vl[1], self.d_do)):
            for rb in range(2):
                for ks in range(self.DK):
                    ins = p.buffer_load(frags[rb][ks], vl[rb], d, 0, offset=32 * ks)
                    if spread_from is not None:      # (spread over the item: the address path takes ~56 cycles per such load and CU)
                        self.gate(ins, spread_from + self.ld_step * n)
                    n += 1
        for dst, d in ((self.ndb[buf], self.d_d), (self.lse2b[buf], self.d_l)):
            for rb in range(2):
                ins = p.buffer_load(dst[rb], self.v_lc[rb], d, 0)
                if spread_from is not None:
                    self.gate(ins, spread_from + self.ld_step * n)

    def emit_offsets_advance(self, p: Prog):
        """load offsets move on by one item (64 rows)"""
        for i, nm in enumerate(("q_sn", "do_sn")):
            p.s_lshl_b32(self.s_tmp[4], P(nm), 6)
            for rb in range(2):
                p.v_add_u32(self.vl[i][rb], self.s_tmp[4], self.vl[i][rb])
        for rb in range(2):
            p.v_add_u32(self.v_lc[rb], 256, self.v_lc[rb])

    def tile_sub(self, j):
        """sub-block classes of the item's tile j, sub[kh][rb] (key half x row block) in "mask" / "full" / "dead" (fwd_strip.py):
        the last tile is the diagonal one - rows 0..31 never see its keys 32..63; steady items: rows 32..63 never see the first
        tile's keys 0..31 and see every one of the diagonal tile's keys 0..31, rows 0..31 every one of the first tile's keys
        32..63.  A dead sub-block is left out of every phase, a full one gets no mask instructions."""
        sub = [["mask", "mask"], ["mask", "mask"]]
        if j == self.NT - 1:
            sub[1][0] = "dead"
            if self.steady:
                sub[0][1] = "full"
        elif j == 0 and self.steady:
            sub[0][1] = "dead"
            sub[1][0] = "full"
        return sub

    def emit_tile(self, p: Prog, buf, j, first):
        """tile j of the current item (ring offset s_slot[j], first key s_k0t[j]); first: dQ^T starts from 0 (srcC = 0)"""
        dt = self.dtype
        sub = self.tile_sub(j)
        started = [False, False]
        QF, DOF = self.QFB[buf], self.DOFB[buf]
        lse2, nd = self.lse2b[buf], self.ndb[buf]
        ke, ko, ve, vo_, tr0, tr1 = self.addr[j & 1]
        vdj = self.v_dj[j & 1]
        p.v_add_u32(ke, self.s_slot[j], self.l_row_e)
        p.v_xor(ko, 32, ke)
        p.v_add_u32(ve, 16384, ke)
        p.v_xor(vo_, 32, ve)
        p.v_add_u32(tr0, self.s_slot[j], self.l_tr0)
        p.v_xor(tr1, 32, tr0)
        NT = self.NT
        if self.steady:
            cls = 3 if j == NT - 1 else (4 if j == 0 else 0)
        else:
            cls = 1
            p.s_cmp("lt_i32", self.s_k0t[j], 0)
            p.s_cselect(self.s_tmp[4], 0, P("W"))
            p.v_mov(self.v_wt, self.s_tmp[4])
        if cls:
            for rb in range(2):
                p.v_lshrrev(self.tmp[1], 5, self.lane)
                p.v_lshlrev(self.tmp[1], 2, self.tmp[1])
                p.v_add_u32(self.tmp[1], self.s_k0t[j], self.tmp[1])
                p.v_sub_u32(vdj[rb], self.v_pos[rb], self.tmp[1])       # pos - k0 - 4 h
                if cls == 4:
                    p.v_sub_u32(vdj[rb], vdj[rb], self.v_wc)
        for kh in range(2):
            kf = []
            for ks in range(self.DK):
                f = self.pool()
                p.ds_read_b128(f, ko if ks & 1 else ke, 8192 * kh + 512 * (ks >> 1), mem=("stage_r",), note="K rows")
                kf.append(f)
            for rb in range(2):
                if sub[kh][rb] == "dead":
                    continue
                for ks in range(self.DK):
                    p.mfma(dt, self.SACC[kh][rb], kf[ks], QF[rb][ks], self.SACC[kh][rb] if ks else 0, tag="S")
            vf = []
            for ks in range(self.DK):
                f = self.pool()
                p.ds_read_b128(f, vo_ if ks & 1 else ve, 8192 * kh + 512 * (ks >> 1), mem=("stage_r",), note="V rows")
                vf.append(f)
            for rb in range(2):
                if sub[kh][rb] == "dead":
                    continue
                for ks in range(self.DK):
                    p.mfma(dt, self.DPACC[kh][rb], vf[ks], DOF[rb][ks], self.DPACC[kh][rb] if ks else 0, tag="dP")
            for rb in range(2):
                if sub[kh][rb] == "dead":
                    continue
                for v in range(16):
                    x, y = self.SACC[kh][rb][v], self.DPACC[kh][rb][v]
                    p.v_fma_f32(x, x, P("c_log2"), lse2[rb])
                    p.v_exp_f32(x, x)
                    if cls and sub[kh][rb] != "full":
                        c = 32 * kh + (v & 3) + 8 * (v >> 2)
                        if cls == 1:
                            p.v_sub_u32(self.tmp[0], vdj[rb], c)
                            p.v_cmp("lt_u32", self.tmp[0], self.v_wt)
                        else:
                            p.v_cmp("le_i32" if cls == 3 else "gt_i32", c, vdj[rb])
                        p.v_cndmask(x, 0, x)
                    p.v_sub_f32(y, y, nd[rb])
                    p.v_mul_f32(y, x, y)
                for s in range(2):
                    for q4 in range(4):
                        d = self.DPACC[kh][rb]
                        p.v_cvt_pk(dt, d[4 * s + q4], d[8 * s + 2 * q4], d[8 * s + 2 * q4 + 1])
            for s in range(2):
                for db in range(self.DB):
                    f = self.pool()
                    off = 8192 * kh + 512 * db
                    p.ds_read_b64_tr_b16(f[0:2], tr0, off + 2048 * (2 * s), mem=("stage_r",))
                    p.ds_read_b64_tr_b16(f[2:4], tr1, off + 2048 * (2 * s + 1), mem=("stage_r",))
                    for rb in range(2):
                        if sub[kh][rb] == "dead":
                            continue
                        zero = first and not started[rb]
                        p.mfma(dt, self.DQ[rb][db], f, self.DPACC[kh][rb][4 * s:4 * s + 4], 0 if zero else self.DQ[rb][db], tag="dQ")
                for rb in range(2):
                    if sub[kh][rb] != "dead":
                        started[rb] = True

    def emit_epilogue(self, p: Prog):
        """the finished item: dQ[row, d] = scale dQ^T[d, row], stores; then the offsets move on by one item"""
        dt = self.dtype
        npair = 0
        for rb in range(2):
            for db in range(self.DB):
                for gp in range(2):
                    if 32 * db + 16 * gp >= self.D:
                        continue
                    X, Y = self.ep[(2 * npair) % 4], self.ep[(2 * npair + 1) % 4]
                    npair += 1
                    for e in range(4):
                        p.v_accvgpr_read(X[e], self.DQ[rb][db][8 * gp + e])
                        p.v_accvgpr_read(Y[e], self.DQ[rb][db][8 * gp + 4 + e])
                    for e in range(4):
                        p.v_mul_f32(X[e], P("scale"), X[e])
                        p.v_mul_f32(Y[e], P("scale"), Y[e])
                    p.v_cvt_pk(dt, X[0], X[0], X[1])
                    p.v_cvt_pk(dt, X[1], X[2], X[3])
                    p.v_cvt_pk(dt, X[2], Y[0], Y[1])
                    p.v_cvt_pk(dt, X[3], Y[2], Y[3])
                    p.v_permlane32_swap(X[0], X[2])
                    p.v_permlane32_swap(X[1], X[3])
                    ins = p.buffer_store(X[0:4], self.vo[rb], self.d_x, 0, offset=64 * db + 32 * gp)
                    if self.st_from is not None:
                        self.gate(ins, self.st_from + self.st_step * (npair - 1))
        p.s_lshl_b32(self.s_tmp[4], P("dq_sn"), 6)
        for rb in range(2):
            p.v_add_u32(self.vo[rb], self.s_tmp[4], self.vo[rb])

    def n_store(self):
        return sum(1 for rb in range(2) for db in range(self.DB) for gp in range(2) if 32 * db + 16 * gp < self.D)

    # ------------------------------------------------------------------ blocks
    def prologue(self) -> Prog:
        p = Prog()
        t0, t1, t2, t3 = self.tmp
        st = self.s_tmp
        lane, wv = self.lane, self.s_wave
        p.v_and(lane, 63, PV("tid"))
        p.v_lshrrev(t0, 6, PV("tid"))
        p.v_readfirstlane(wv, t0)
        p.v_and(self.lane31, 31, lane)
        p.v_mov(self.v_oob, imm(0x7FFFF000))
        p.s_mov(self.s_hh, wv)                                # hpw = 4: wave = head, one row group
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
                if self.l_dma1 is not None:
                    p.v_lshrrev(self.vt[1], 4, t3)
                    p.v_add_u32(self.l_dma1[e][col], 128, self.l_dma[e][col])
                    p.v_cmp("gt_u32", self.NCH - 8, self.vt[1])
                    p.v_cndmask(self.l_dma1[e][col], self.v_oob, self.l_dma1[e][col])
        p.s_lshl_b32(self.s_wofs, wv, 12)
        p.s_lshr_b32(st[0], P("q0"), 6)
        p.s_sub_i32(self.s_T, st[0], self.NT - 1)
        p.s_mov(self.s_s0, 0)
        p.s_mov(self.s_n, P("n_it"))
        self.emit_item_scalars(p)
        for j in range(self.NT):
            p.s_add_i32(st[3], self.s_T, j)
            self.emit_dma(p, st[3], self.s_slot[j])
        # rows of the lane, descriptors (head = wave), offsets of the first item's loads / stores
        p.v_add_u32(t0, P("q0"), self.lane31)
        p.v_mov(self.v_pos[0], t0)
        p.v_add_u32(self.v_pos[1], 32, self.v_pos[0])
        for nm, d in (("q", self.d_q), ("do", self.d_do), ("dq", self.d_x)):
            p.s_mul_i32(st[1], self.s_hh, P(nm + "_hs"))
            p.s_mul_hi_u32(st[2], self.s_hh, P(nm + "_hs"))
            p.s_add_u32(d[0], P(nm + "_lo"), st[1])
            p.s_addc_u32(d[1], P(nm + "_hi"), st[2])
            p.s_mov(d[2], P(nm + "_rng"))
            p.s_mov(d[3], 0x00020000)
        p.s_mul_i32(st[1], self.s_hh, P("ld_hs"))
        p.s_mul_hi_u32(st[2], self.s_hh, P("ld_hs"))
        for nm, d in (("lse", self.d_l), ("dl", self.d_d)):
            p.s_add_u32(d[0], P(nm + "_lo"), st[1])
            p.s_addc_u32(d[1], P(nm + "_hi"), st[2])
            p.s_lshl_b32(d[2], P("nrows"), 2)
            p.s_mov(d[3], 0x00020000)
        for i, nm in enumerate(("q", "do")):
            p.v_mul_lo_u32(t1, t0, P(nm + "_sn"))
            p.v_lshl_add_u32(self.vl[i][0], t2, 4, t1)
            p.s_lshl_b32(st[1], P(nm + "_sn"), 5)
            p.v_add_u32(self.vl[i][1], st[1], self.vl[i][0])
        p.v_mul_lo_u32(t1, t0, P("dq_sn"))
        p.v_lshl_add_u32(self.vo[0], t2, 4, t1)
        p.s_lshl_b32(st[1], P("dq_sn"), 5)
        p.v_add_u32(self.vo[1], st[1], self.vo[0])
        p.v_lshlrev(self.v_lc[0], 2, t0)
        p.v_add_u32(self.v_lc[1], 128, self.v_lc[0])
        self.emit_loads(p, 0)
        p.v_mov(self.v_wc, P("W"))
        p.s_waitcnt(vmcnt=0, note="first item: its tiles, fragments and row constants")
        p.s_barrier()
        for rb in range(2):
            p.v_mul_f32(self.lse2b[0][rb], P("nlog2e"), self.lse2b[0][rb])
        return p

    def item(self, buf: int, head: bool) -> Prog:
        """one item: [bookkeeping | requests of the next item | tile 0 with the previous item's epilogue in its gaps | tiles 1 ..]"""
        p = Prog()
        NT = self.NT
        self.pool_next = 0
        if not head:
            p.s_add_i32(self.s_T, self.s_T, 1)
            self.emit_wrap(p, self.s_s0, self.s_s0, STG_BYTES)
            p.s_sub_u32(self.s_n, self.s_n, 1)
            self.emit_item_scalars(p)
            for rb in range(2):
                p.v_add_u32(self.v_pos[rb], 64, self.v_pos[rb])
            for rb in range(2):          # (the row constants were requested as LSE: into the exp2 domain once per item)
                p.v_mul_f32(self.lse2b[buf][rb], P("nlog2e"), self.lse2b[buf][rb])
        # requests for the NEXT item (in front of the finished item's stores in the memory queue)
        self.emit_offsets_advance(p)
        self.emit_loads(p, buf ^ 1, spread_from=self.ld_from)
        p.s_add_i32(self.s_tmp[3], self.s_T, NT)
        self.emit_wrap(p, self.s_tmp[2], self.s_slot[NT - 1], STG_BYTES)
        self.emit_dma(p, self.s_tmp[3], self.s_tmp[2], spread_from=self.dma_from)
        if not head:
            self.emit_epilogue(p)
        for j in range(NT):
            self.emit_tile(p, buf, j, first=(j == 0))
        return p

    def tail(self) -> Prog:
        p = Prog()
        self.emit_epilogue(p)
        p.s_waitcnt(vmcnt=0)
        return p

    def build(self):
        items = finish_block(self.prologue().items)
        nst = self.n_store()

        def block(prog):
            b = prog.items
            if self.do_sched:
                b = schedule(b)
            return fix_hazards(insert_waits(b))

        def item_block(buf, head, steady=False):
            self.steady = steady
            out = block(self.item(buf, head))
            self.steady = False
            return out

        items += item_block(0, True)
        p = Prog()
        p.s_mov(self.s_par, 1)
        p.s_cmp("le_u32", self.s_n, 1)
        p.s_cbranch("scc1", "L_tail%=")
        p.s_waitcnt(vmcnt=0, note="second item's requests (the first item has left no stores behind)")
        p.s_branch("L_disp%=")
        p.label("L_next%=")
        p.s_cmp("le_u32", self.s_n, 1)
        p.s_cbranch("scc1", "L_tail%=")
        p.s_waitcnt(vmcnt=0, note="next item's fragments, constants and newest tile (own pieces)")
        p.label("L_disp%=")
        p.s_barrier()
        p.s_cmp("ge_i32", self.s_T, -1)
        p.s_cselect(self.s_tmp[0], 2, 0)
        p.s_sub_u32(self.s_tmp[1], P("W"), 64 * (self.NT - 1))
        p.s_cmp("le_u32", self.s_tmp[1], 1)
        p.s_cselect(self.s_tmp[0], self.s_tmp[0], 0)
        p.s_or_b32(self.s_tmp[0], self.s_tmp[0], self.s_par)
        for code in (3, 2, 1):
            p.s_cmp("eq_u32", self.s_tmp[0], code)
            p.s_cbranch("scc1", "L_it%d_%%=" % code)
        items += finish_block(p.items)
        for code in (0, 3, 2, 1):
            p = Prog()
            p.label("L_it%d_%%=" % code)
            p.s_mov(self.s_par, (code & 1) ^ 1)
            items += finish_block(p.items)
            items += item_block(code & 1, False, steady=bool(code & 2))
            items.append(Instr("s_branch", mods={"label": "L_next%="}, kind="branch"))
        p = Prog()
        p.label("L_tail%=")
        items += finish_block(p.items)
        items += block(self.tail())
        return items
