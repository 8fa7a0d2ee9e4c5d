"""Generator of the SKEWED dK/dV body: short windows without sink keys, head dims 64 / 80 / 96 (gfx950, bf16 / f16).

Same work decomposition and maths as dkdv.py (workgroup = 4 waves = 256 keys of one (batch, KV head), wave w keeps dK^T / dV^T
of keys [64 w, 64 w + 64) in accumulator registers, trips of 32 query rows), other sweep.  With a window of W keys a wave's 64
keys are seen by the T = floor((62 + W) / 32) + 1 slices [2 w, 2 w + T) behind the block's first row, the block as a whole by
T + 6: in lock step over the block's slices every wave idles through 6 of T + 6 trips (W = 128: 6 of 12).  Here a q head is
swept in T trips; in trip r wave w takes
        slice r of head h                 if r >= 2 w          ("A")
        slice T + r of head h - 1         if r <  2 w          ("B": the tail the wave still owes the previous head)
so that every slice is consumed in ONE trip by all the waves that need it, a trip consumes exactly two slices (one for r >= 6)
and the ring stays 4 stages deep: stage = [A image | B image], 32 KB.  Trips per workgroup T g + 6 instead of (T + 6) g
(W = 128, g = 8: 54 instead of 96).  The slices of head -1 and head g do not exist: zero-record descriptors, the waves compute
zeros there.  Head dims below 128 leave accumulator registers free: the V fragments of the wave's keys are pinned there like the
K fragments (no V image in LDS, no V reads in the trip).

LDS map (bytes):   [0, 2048)            row constants, 4 stages x (A: 32 x -LSE/scale | 32 x -Delta ; B: the same) f32
                   [2048, 133120)       4 stages x (A: Q slice image 8 KB | dO slice image 8 KB ; B: the same)
Slice images as in dkdv.py.
"""
from __future__ import annotations

from .core import A, Imm, Instr, M0, P, PV, Prog, Reg, S, V, VCC, fimm, imm
from .dkdv import DkdvGen
from .sched import finish_block, fix_hazards, insert_waits, schedule

CST_BASE = 0
CST_STAGE = 512
STG_BASE = 2048
STG_BYTES = 32768
B_IMG = 16384                               # the B image inside a stage; its constants sit 256 bytes behind A's
LDS_BYTES = STG_BASE + 4 * STG_BYTES        # 133120
SKEW = 6                                    # slices between the first rows of wave 0's and wave 3's keys

# scalar inputs: dkdv.PARAMS without the sink / split / N_q < N_kv fields.  nq = T = trips per q head (>= SKEW: the shell
# rounds short windows up); q_row0 = first row of the block's slice 0 (= its first key: N_q = N_kv here)
PARAMS = [
    "q_lo", "q_hi", "do_lo", "do_hi", "c_lo", "c_hi", "k_lo", "k_hi", "v_lo", "v_hi", "dk_lo", "dk_hi", "dv_lo", "dv_hi",
    "q_rng", "do_rng", "c_rng", "k_rng", "v_rng", "dk_rng", "dv_rng",
    "q_sn", "do_sn", "k_sn", "v_sn", "dk_sn", "dv_sn",
    "q_hs", "do_hs", "c_hs",
    "nq", "g", "q_row0", "kb0", "W", "nrows", "cdelta", "c_log2", "scale",
]


class DkdvSkewGen(DkdvGen):
    def __init__(self, dtype="bf16", D=80, sched=True, npool=12, dma_t0=40, dma_dt=None, ablate=(), halves=True):
        assert D in (64, 80, 96)
        self.halves = halves            # trip bodies for slices that touch only one of the wave's two 32-key blocks
        if dma_dt is None:          # ten pieces per trip above head dim 64: spread over the whole trip (C4: -4 % against 100)
            dma_dt = 140 if D > 64 else 100
        super().__init__(dtype, sched=sched, sfirst=48, npool=npool, dma_t0=dma_t0, dma_dt=dma_dt, D=D, ablate=ablate, dead=False, half_edges=False)
        self.partials = False
        va, sa = self.va, self.sa
        # V fragments of the wave's keys in the accumulator registers the narrower dK^T / dV^T tiles leave free
        free = [32 * self.DB, 128 + 32 * self.DB]
        assert 4 * self.DK <= 128 - 32 * self.DB
        self.VF = [[A(free[kbi] + 4 * ks, 4) for ks in range(self.DK)] for kbi in range(2)]
        # B stream: source offsets of the wave's pieces (A's + T slices, previous head through a scratch descriptor)
        self.vo_qb = [self.a_v_e, self.a_v_o]           # (the V image address registers of the base class are idle here)
        self.vo_db = [va("vo_db0"), va("vo_db1")]
        self.vo_cb = va("vo_cb")
        self.s_w2 = sa("s_w2")                          # 2 * wave: the wave takes the B slice in trips r < 2 wave
        self.s_stw, self.s_stnw = sa("s_stw"), sa("s_stnw")          # image base of this / the next trip, A or B chosen
        self.s_cstw, self.s_cstnw = sa("s_cstw"), sa("s_cstnw")
        self.s_bmask = sa("s_bmask")                    # all ones while the B head (s_ldh - 1) exists
        self.s_t32 = self.s_allsink                     # 32 T (no sink keys here: the base class's flag register is free)
        self.NP = 2 * (2 * self.HALVES + 1)             # LDS-DMA pieces per wave and trip

    def params(self):
        return list(PARAMS)

    # ------------------------------------------------------------------ slice stream
    def emit_dma_issue(self, p: Prog, spread: bool = False):
        dl = (lambda i: {"alap": self.dma_t0 + self.dma_dt * i}) if spread else (lambda i: {})
        n = 0
        # ---- B: slice T + ldq of head ldh - 1, through the scratch descriptor (the head's base one head stride back); no wave
        #      takes a B slice in trips r >= SKEW (windows above 128: T > SKEW): zero records there as well
        bm = self.s_tmp[2]
        p.s_cmp("lt_u32", self.s_ldq, SKEW)
        p.s_cselect(bm, self.s_bmask, 0)
        for nm, d, vo, img in (("q", self.d_q, self.vo_qb, B_IMG), ("do", self.d_do, self.vo_db, B_IMG + 8192)):
            p.s_sub_u32(self.d_x[0], d[0], P(nm + "_hs"))
            p.s_subb_u32(self.d_x[1], d[1], 0)
            p.s_and_b32(self.d_x[2], P(nm + "_rng"), bm)
            p.s_add_m0(self.s_std, self.s_wofs)
            if img:
                p.s_add_m0(M0, img)
            p.buffer_load_lds(16, vo[0], self.d_x, 0, mem=("dma_stage",)).mods.update(dl(n)); n += 1
            if self.HALVES == 2:
                p.s_add_m0(M0, 1024)
                p.buffer_load_lds(16, vo[1], self.d_x, 0, mem=("dma_stage",)).mods.update(dl(n)); n += 1
        p.s_sub_u32(self.d_x[0], self.d_c[0], P("c_hs"))
        p.s_subb_u32(self.d_x[1], self.d_c[1], 0)
        p.s_and_b32(self.d_x[2], P("c_rng"), bm)
        p.s_sub_i32(self.s_tmp[0], P("nrows"), self.s_ldrow, note="rows left in the sequence behind A's first row")
        p.s_sub_i32(self.s_tmp[1], self.s_tmp[0], self.s_t32)
        p.v_cmp("gt_i32", self.s_tmp[1], self.lane31)
        p.v_cndmask(self.vo_ce, self.v_oob, self.vo_cb)
        p.s_add_m0(self.s_cstd, 256)
        p.buffer_load_lds(4, self.vo_ce, self.d_x, 0, mem=("dma_stage",)).mods.update(dl(n)); n += 1
        # ---- A: slice ldq of head ldh
        p.s_add_m0(self.s_std, self.s_wofs, note="Q piece 0 of this wave")
        p.buffer_load_lds(16, self.vo_q[0], self.d_q, 0, mem=("dma_stage",)).mods.update(dl(n)); n += 1
        if self.HALVES == 2:
            p.s_add_m0(M0, 1024)
            p.buffer_load_lds(16, self.vo_q[1], self.d_q, 0, mem=("dma_stage",)).mods.update(dl(n)); n += 1
        p.s_add_m0(M0, 8192 - 1024 * (self.HALVES - 1), note="dO piece 0")
        p.buffer_load_lds(16, self.vo_d[0], self.d_do, 0, mem=("dma_stage",)).mods.update(dl(n)); n += 1
        if self.HALVES == 2:
            p.s_add_m0(M0, 1024)
            p.buffer_load_lds(16, self.vo_d[1], self.d_do, 0, mem=("dma_stage",)).mods.update(dl(n)); n += 1
        p.v_cmp("gt_i32", self.s_tmp[0], self.lane31)
        p.v_cndmask(self.vo_ce, self.v_oob, self.vo_c)
        p.s_mov_m0(self.s_cstd)
        p.buffer_load_lds(4, self.vo_ce, self.d_c, 0, mem=("dma_stage",)).mods.update(dl(n)); n += 1
        assert n == self.NP

    def emit_dma_step(self, p: Prog):
        super().emit_dma_step(p)
        for h in range(self.HALVES):
            p.v_add_u32(self.vo_qb[h], self.s_stepq, self.vo_qb[h])
            p.v_add_u32(self.vo_db[h], self.s_stepd, self.vo_db[h])
        p.v_add_u32(self.vo_cb, 128, self.vo_cb)

    def emit_dma_headchange(self, p: Prog, lbl_done: str):
        """the next pair of slices starts a new A head (B = the head just left); behind head g - 1 A is empty, behind head g
        both are (the fetches that the pipeline still issues then touch no memory)"""
        p.s_mov(self.s_ldq, 0)
        p.s_mov(self.s_ldrow, P("q_row0"))
        p.s_add_u32(self.s_ldh, self.s_ldh, 1)
        for d, hs in ((self.d_q, "q_hs"), (self.d_do, "do_hs"), (self.d_c, "c_hs")):
            p.s_add_u32(d[0], d[0], P(hs))
            p.s_addc_u32(d[1], d[1], 0)
        for h in range(self.HALVES):
            p.v_sub_u32(self.vo_q[h], self.vo_q[h], self.s_spanq)
            p.v_sub_u32(self.vo_d[h], self.vo_d[h], self.s_spand)
            p.v_sub_u32(self.vo_qb[h], self.vo_qb[h], self.s_spanq)
            p.v_sub_u32(self.vo_db[h], self.vo_db[h], self.s_spand)
        p.v_sub_u32(self.vo_c, self.vo_c, self.s_spanc)
        p.v_sub_u32(self.vo_cb, self.vo_cb, self.s_spanc)
        p.s_cmp("le_u32", self.s_ldh, P("g"))
        p.s_cselect(self.s_bmask, -1, 0)
        p.s_cmp("lt_u32", self.s_ldh, P("g"))
        p.s_cbranch("scc1", lbl_done)
        p.s_mov(self.d_q[2], 0)
        p.s_mov(self.d_do[2], 0)
        p.s_mov(self.d_c[2], 0)

    def emit_class(self, p: Prog, q0p=None):
        """s_full = class of (the wave's 64 keys) x (the 32 rows at q0p): 1 <=> every key is causal for and inside the window
        of every row; 3 / 4 <=> no row sees any key of the SECOND / FIRST 32-key block (the other one is masked element by
        element): a wave's first slice touches only its first key block, its last slice only the second - two of its T
        trips per head; else 0 (both blocks masked element by element)"""
        q0p = self.s_q0p if q0p is None else q0p
        t0, t1, t2 = self.s_tmp[1], self.s_tmp[2], self.s_tmp[0]
        p.s_cmp("le_i32", self.s_kw63, q0p)
        p.s_cselect(t0, 1, 0)
        p.s_add_u32(t1, q0p, 31)
        p.s_cmp("gt_i32", self.s_kww, t1)
        p.s_cselect(t1, 1, 0)
        p.s_and_b32(self.s_full, t0, t1)
        if self.halves:
            # second block (keys kw63 - 31 .. kw63) dead: behind every row (kw63 - 31 > q0p + 31) or out of every window
            # (q0p - kw63 >= W: q0p >= kww + 63); first block (kw63 - 63 .. kw63 - 32): kw63 - 63 > q0p + 31, q0p >= kww + 31
            p.s_add_u32(t1, q0p, 62)
            p.s_cmp("gt_i32", self.s_kw63, t1)
            p.s_cselect(t0, 1, 0)
            p.s_add_u32(t1, self.s_kww, 63)
            p.s_cmp("ge_i32", q0p, t1)
            p.s_cselect(t1, 1, 0)
            p.s_or_b32(t0, t0, t1)                             # second block dead
            p.s_add_u32(t1, q0p, 94)
            p.s_cmp("gt_i32", self.s_kw63, t1)
            p.s_cselect(t2, 1, 0)
            p.s_add_u32(t1, self.s_kww, 31)
            p.s_cmp("ge_i32", q0p, t1)
            p.s_cselect(t1, 1, 0)
            p.s_or_b32(t2, t2, t1)                             # first block dead
            p.s_cmp("lg_u32", t0, t2)                          # exactly one of them (both: the masks make zeros of it)
            p.s_cselect(t1, 1, 0)
            p.s_and_b32(t0, t0, t1)
            p.s_and_b32(t2, t2, t1)
            p.s_cmp("lg_u32", t0, 0)
            p.s_cselect(self.s_full, 3, self.s_full)
            p.s_cmp("lg_u32", t2, 0)
            p.s_cselect(self.s_full, 4, self.s_full)

    # ------------------------------------------------------------------ prologue
    def prologue(self) -> Prog:
        p = Prog()
        t0, t1, t2, t3 = self.tmp
        lane, wv = self.lane, self.s_wave
        p.v_and(lane, 63, PV("tid"))
        p.v_lshrrev(t0, 6, PV("tid"))
        p.v_readfirstlane(wv, t0)
        p.s_lshl_b32(self.s_w2, wv, 1)
        p.s_lshl_b32(self.s_t32, P("nq"), 5)
        p.v_and(self.lane31, 31, lane)
        p.v_mov(self.v_oob, imm(0x7FFFFFF0))
        # r = lane & 31, h = lane >> 5 ; l_row_e = 2048 (r >> 3) + 64 (r & 7) + 16 (h ^ ((r >> 2) & 3))
        p.v_lshrrev(t0, 3, self.lane31)
        p.v_lshlrev(t0, 11, t0)
        p.v_and(t1, 7, lane)
        p.v_lshl_add_u32(t0, t1, 6, t0)
        p.v_bfe_u32(t1, lane, 2, 2)
        p.v_lshrrev(t2, 5, lane)
        p.v_xor(t1, t1, t2)
        p.v_lshl_add_u32(self.l_row_e, t1, 4, t0)
        p.v_lshlrev(self.l_c, 4, t2)
        # l_tr0 = 64 (4 h + q4) + 16 ((2 g1 + (p4 >> 1)) ^ h) + 8 (p4 & 1)
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
        p.s_lshl_b32(self.s_wofs, wv, 11)

        for d, nm in ((self.d_q, "q"), (self.d_do, "do"), (self.d_c, "c")):
            p.s_mov(d[0], P(nm + "_lo"))
            p.s_mov(d[1], P(nm + "_hi"))
            p.s_mov(d[2], P(nm + "_rng"))
            p.s_mov(d[3], 0x00020000)

        # ---- K and V fragments: lane (r, h) of key block kbi holds X[key][16 ks + 8 h ..+8)
        p.s_lshl_b32(self.s_tmp[0], wv, 6)
        p.s_add_u32(self.s_tmp[0], self.s_tmp[0], P("kb0"))       # first key of the wave
        p.s_mov(self.d_x[3], 0x00020000)
        for nm, frags in (("k", self.KF), ("v", self.VF)):
            p.s_mov(self.d_x[0], P(nm + "_lo"))
            p.s_mov(self.d_x[1], P(nm + "_hi"))
            p.s_mov(self.d_x[2], P(nm + "_rng"))
            p.s_mul_i32(self.s_tmp[1], self.s_tmp[0], P(nm + "_sn"))
            p.v_mul_lo_u32(t2, self.lane31, P(nm + "_sn"))
            p.v_add_u32(t2, self.s_tmp[1], t2)
            p.v_lshrrev(t3, 5, lane)
            p.v_lshl_add_u32(self.vo_k[0], t3, 4, t2)
            p.s_lshl_b32(self.s_tmp[1], P(nm + "_sn"), 5)
            p.v_add_u32(self.vo_k[1], self.s_tmp[1], self.vo_k[0])
            for kbi in range(2):
                for ks in range(self.DK):
                    p.buffer_load(frags[kbi][ks], self.vo_k[kbi], self.d_x, 0, offset=32 * ks)

        # ---- mask constants: key0 = wave key0 + r ; v_kh = key0 - 4 h ; weff = W (no sink keys)
        p.v_add_u32(t2, self.s_tmp[0], self.lane31)
        p.v_lshrrev(t3, 5, lane)
        p.v_lshlrev(t3, 2, t3)
        p.v_sub_u32(self.v_kh, t2, t3)
        p.v_mov(self.v_weff[0], P("W"))
        p.v_mov(self.v_weff[1], P("W"))
        p.s_add_u32(self.s_kw63, self.s_tmp[0], 63)
        p.s_add_u32(self.s_kww, self.s_tmp[0], P("W"))

        # ---- slice streams: source offsets of this wave's pieces.  piece (2 wave + e): rows 8 wave + rr, 16-byte chunk
        #      4 (2 e + cbl) + (slot ^ ((2 wave + (rr >> 2)) & 3))
        rr, slot = t0, t1
        p.v_bfe_u32(rr, lane, 2, 3)
        p.v_and(slot, 3, lane)
        p.v_lshrrev(t3, 2, rr)
        p.s_lshl_b32(self.s_tmp[1], wv, 1)
        p.v_add_u32(t3, self.s_tmp[1], t3)
        p.v_and(t3, 3, t3)
        p.v_xor(t3, t3, slot)
        p.v_lshrrev(t2, 5, lane)
        p.v_lshl_add_u32(t3, t2, 2, t3)
        p.v_lshlrev(t3, 4, t3)
        p.s_lshl_b32(self.s_tmp[1], wv, 3)
        p.s_add_u32(self.s_tmp[1], self.s_tmp[1], P("q_row0"))
        p.v_add_u32(t2, self.s_tmp[1], rr)
        for vo, sn in ((self.vo_q, "q_sn"), (self.vo_d, "do_sn")):
            p.v_mul_lo_u32(vo[0], t2, P(sn))
            p.v_add_u32(vo[0], vo[0], t3)
            if self.HALVES == 2:
                p.v_lshrrev(rr, 4, t3)
                self.emit_half1(p, vo[1], vo[0], rr, 0, slot)
        p.v_add_u32(t2, P("q_row0"), self.lane31)
        p.v_lshlrev(t2, 2, t2)
        p.v_lshrrev(t3, 5, lane)
        p.v_mul_lo_u32(t3, t3, P("cdelta"))
        p.v_add_u32(self.vo_c, t2, t3)
        p.s_lshl_b32(self.s_stepq, P("q_sn"), 5)
        p.s_lshl_b32(self.s_stepd, P("do_sn"), 5)
        p.s_mul_i32(self.s_spanq, self.s_stepq, P("nq"))
        p.s_mul_i32(self.s_spand, self.s_stepd, P("nq"))
        p.s_lshl_b32(self.s_spanc, P("nq"), 7)
        # B = A + T slices (out-of-range chunks of the second half stay out of range: the offset is far above any range)
        for h in range(self.HALVES):
            p.v_add_u32(self.vo_qb[h], self.s_spanq, self.vo_q[h])
            p.v_add_u32(self.vo_db[h], self.s_spand, self.vo_d[h])
        p.v_add_u32(self.vo_cb, self.s_spanc, self.vo_c)
        p.s_mov(self.s_ldq, 0)
        p.s_mov(self.s_ldh, 0)
        p.s_mov(self.s_bmask, 0)
        p.s_mov(self.s_ldrow, P("q_row0"))
        # trips: T per head, and SKEW more for the tail the waves still owe the last head (B slices exist for r < SKEW only)
        p.s_mul_i32(self.s_n, P("g"), P("nq"))
        p.s_add_u32(self.s_n, self.s_n, SKEW)

        for acc in (self.DV, self.DKA):
            for db in range(self.DB):
                for kbi in range(2):
                    for i in range(16):
                        p.v_accvgpr_write(acc[db][kbi][i], 0)

        # ---- first three stages
        for j in range(3):
            p.s_mov(self.s_std, STG_BASE + j * STG_BYTES)
            p.s_mov(self.s_cstd, CST_BASE + j * CST_STAGE)
            self.emit_dma_headcheck(p, "pro%d" % j)
            self.emit_dma_issue(p)
            self.emit_dma_step(p)
        p.s_waitcnt(vmcnt=2 * self.NP, note="K / V fragments, stage 0 landed (stages 1, 2 in flight)")
        p.s_barrier()
        # ---- state of trip 0 (r = 0: wave 0 on A, the others on B) and of trip 1
        p.s_mov(self.s_t, 0)
        p.s_mov(self.s_cq, 0)
        p.s_mov(self.s_st, STG_BASE)
        p.s_mov(self.s_stn, STG_BASE + STG_BYTES)
        p.s_mov(self.s_std, STG_BASE + 3 * STG_BYTES)
        p.s_mov(self.s_cst, CST_BASE)
        p.s_mov(self.s_cstn, CST_BASE + CST_STAGE)
        p.s_mov(self.s_cstd, CST_BASE + 3 * CST_STAGE)
        self.emit_wave_state(p, 0, self.s_st, self.s_cst, self.s_stw, self.s_cstw, self.s_q0p)
        self.emit_class(p)
        p.s_mov(self.s_tmp[0], 1)
        self.emit_wave_state(p, self.s_tmp[0], self.s_stn, self.s_cstn, self.s_stnw, self.s_cstnw, None)
        p.v_add_u32(self.a_rown_e, self.s_stw, self.l_row_e)
        p.v_xor(self.a_rown_o, 32, self.a_rown_e)
        p.v_add_u32(self.a_cn, self.s_cstw, self.l_c)
        self.emit_next_prefetch(p)
        return p

    def emit_wave_state(self, p: Prog, r, st, cst, stw, cstw, q0p):
        """stw / cstw = image / constants base of the slice this wave takes in trip r of a head (stage bases st, cst);
        q0p = position of that slice's first row (None: not wanted).  Uses s_tmp[3]."""
        t = self.s_tmp[3]
        p.s_cmp("lt_u32", r, self.s_w2)
        p.s_cselect(stw, B_IMG, 0)
        p.s_cselect(cstw, 256, 0)
        if q0p is not None:
            p.s_cselect(t, self.s_t32, 0)
        p.s_add_u32(stw, stw, st)
        p.s_add_u32(cstw, cstw, cst)
        if q0p is not None:
            p.s_lshl_b32(q0p, r, 5)
            p.s_add_u32(q0p, q0p, t)
            p.s_add_u32(q0p, q0p, P("q_row0"))

    # ------------------------------------------------------------------ loop head
    def loop_top(self) -> Prog:
        p = Prog()
        p.label("L_top%=")
        p.s_cmp("ge_u32", self.s_t, self.s_n)
        p.s_cbranch("scc1", "L_done%=")
        p.s_cmp("lt_u32", self.s_ldq, P("nq"))
        p.s_cbranch("scc0", "L_dmahead%=")
        p.label("L_top_a%=")
        p.s_cmp("lg_u32", self.s_full, 0)
        p.s_waitcnt(vmcnt=self.NP, note="stage t+1 landed (own pieces); stage t+2 may be in flight")
        p.s_barrier()
        p.s_waitcnt(lgkmcnt=0, note="S-chain operands of this trip (fetched at the end of the last one)")
        p.s_cbranch("scc0", "L_edge%=")
        if self.halves:
            p.s_cmp("eq_u32", self.s_full, 3)
            p.s_cbranch("scc1", "L_e0%=")
            p.s_cmp("eq_u32", self.s_full, 4)
            p.s_cbranch("scc1", "L_e1%=")
        return p

    def trip_bodies(self):
        b = [(None, self.trip_body(False)), ("L_edge%=", self.trip_body(True))]
        if self.halves:
            b += [("L_e0%=", self.trip_body(True, live=(0,))), ("L_e1%=", self.trip_body(True, live=(1,)))]
        return b

    def out_of_line(self) -> Prog:
        p = Prog()
        p.label("L_dmahead%=")
        self.emit_dma_headchange(p, "L_top_a%=")
        p.s_branch("L_top_a%=")
        return p

    # ------------------------------------------------------------------ one trip
    def trip_body(self, edge: bool, live=(0, 1)) -> Prog:
        """live: the wave's 32-key blocks that take part (the other one is seen by no row of the slice)"""
        p = Prog()
        dt = self.dtype
        self.pool_next = 0
        p.v_add_u32(self.a_row_e, self.s_stw, self.l_row_e)
        p.v_xor(self.a_row_o, 32, self.a_row_e)
        p.v_add_u32(self.a_tr0, self.s_stw, self.l_tr0)
        p.v_xor(self.a_tr1, 32, self.a_tr0)
        p.v_add_u32(self.a_c, self.s_cstw, self.l_c)
        p.v_add_u32(self.a_rown_e, self.s_stnw, self.l_row_e)
        p.v_xor(self.a_rown_o, 32, self.a_rown_e)
        p.v_add_u32(self.a_cn, self.s_cstnw, self.l_c)
        for kbi in live:
            for g4 in range(4):
                p.ds_read_b128(self.DPACC[kbi][4 * g4:4 * g4 + 4], self.a_c, 128 + 32 * g4, mem=("stage_r",))
        # fetch stage t + 3
        self.emit_dma_issue(p, spread=True)
        self.emit_dma_step(p)
        # ---- [A] S' = Q K^T - LSE/scale
        for kbi in live:
            for ks in range(self.DK):
                p.mfma(dt, self.SACC[kbi], self.QROW[ks], self.KF[kbi][ks], self.SACC[kbi], tag="S")
        if edge:
            p.v_sub_u32(self.v_d[0], self.s_q0p, self.v_kh, note="(q0 + 4 h) - key")
            p.v_sub_u32(self.v_d[1], self.v_d[0], 32)
        for kbi in live:
            # (v_pk_mul_f32 for the two multiplies of this body - half the issue slots - measured no faster: C4 dK/dV 84.0 vs
            # 84.2 us, C4 x 4 batches 339 vs 329 us, same box, profiles/r03_skew_pk.log)
            for v in range(16):
                x = self.SACC[kbi][v]
                p.v_mul_f32(x, P("c_log2"), x)
                p.v_exp_f32(x, x)
                if edge:
                    o = (v & 3) + 8 * (v >> 2)
                    p.v_add_u32(self.tmp[0], o, self.v_d[kbi])
                    p.v_cmp("lt_u32", self.tmp[0], self.v_weff[kbi])
                    p.v_cndmask(x, 0, x)
            for s in range(2):
                for j in range(4):
                    p.v_cvt_pk(dt, self.PPK[kbi][s][j], self.SACC[kbi][8 * s + 2 * j], self.SACC[kbi][8 * s + 2 * j + 1])
        # ---- [B] dP' = dO V^T - Delta (V fragments pinned)
        for ks in range(self.DK):
            base = self.a_row_o if ks & 1 else self.a_row_e
            fa = self.pool()
            p.ds_read_b128(fa, base, 8192 + 512 * (ks >> 1), mem=("stage_r",), note="dO rows, k-step %d" % ks)
            for kbi in live:
                p.mfma(dt, self.DPACC[kbi], fa, self.VF[kbi][ks], self.DPACC[kbi], tag="dP")
        for kbi in live:
            for v in range(16):
                p.v_mul_f32(self.DPACC[kbi][v], self.SACC[kbi][v], self.DPACC[kbi][v])
            for s in range(2):
                for j in range(4):
                    p.v_cvt_pk(dt, self.DPACC[kbi][4 * s + j], self.DPACC[kbi][8 * s + 2 * j], self.DPACC[kbi][8 * s + 2 * j + 1])
        # ---- [C] dV^T += dO^T P ; [D] dK^T += Q^T dS
        for which in ("dV", "dK"):
            img = 8192 if which == "dV" else 0
            for db in range(self.DB):
                for s in range(2):
                    f = self.pool()
                    p.ds_read_b64_tr_b16(f[0:2], self.a_tr0, img + 2048 * (2 * s) + 512 * db, mem=("stage_r",))
                    p.ds_read_b64_tr_b16(f[2:4], self.a_tr1, img + 2048 * (2 * s + 1) + 512 * db, mem=("stage_r",))
                    for kbi in live:
                        if which == "dV":
                            p.mfma(dt, self.DV[db][kbi], f, self.PPK[kbi][s], self.DV[db][kbi], tag="dV")
                        else:
                            p.mfma(dt, self.DKA[db][kbi], f, self.DPACC[kbi][4 * s:4 * s + 4], self.DKA[db][kbi], tag="dK")
        n_mfma = (4 * self.DK + 8 * self.DB) * len(live) // 2
        self.emit_next_prefetch(p, deadline=max(200, n_mfma * 32 - 900))
        # ---- scalar state of the next trip: r' = r + 1 (mod T), the trip after it r'' (its stage follows s_stn in the ring)
        t0 = self.s_tmp[0]          # (free again behind the DMA issue, which the register dependences order in front)
        p.s_add_u32(self.s_t, self.s_t, 1)
        p.s_add_u32(self.s_cq, self.s_cq, 1)
        p.s_cmp("ge_u32", self.s_cq, P("nq"))
        p.s_cselect(self.s_cq, 0, self.s_cq)
        p.s_mov(self.s_stw, self.s_stnw)
        p.s_mov(self.s_cstw, self.s_cstnw)
        # q0p / class of trip r'
        t3 = self.s_tmp[3]
        p.s_cmp("lt_u32", self.s_cq, self.s_w2)
        p.s_cselect(t3, self.s_t32, 0)
        p.s_lshl_b32(self.s_q0p, self.s_cq, 5)
        p.s_add_u32(self.s_q0p, self.s_q0p, t3)
        p.s_add_u32(self.s_q0p, self.s_q0p, P("q_row0"))
        self.emit_class(p)
        # ring
        p.s_mov(self.s_st, self.s_stn)
        p.s_add_u32(t3, self.s_stn, STG_BYTES - STG_BASE)
        p.s_and_b32(t3, t3, 4 * STG_BYTES - 1)
        p.s_add_u32(self.s_stn, t3, STG_BASE)
        p.s_add_u32(t3, self.s_std, STG_BYTES - STG_BASE)
        p.s_and_b32(t3, t3, 4 * STG_BYTES - 1)
        p.s_add_u32(self.s_std, t3, STG_BASE)
        p.s_mov(self.s_cst, self.s_cstn)
        p.s_add_u32(t3, self.s_cstn, CST_STAGE)
        p.s_and_b32(self.s_cstn, t3, 4 * CST_STAGE - 1)
        p.s_add_u32(t3, self.s_cstd, CST_STAGE)
        p.s_and_b32(self.s_cstd, t3, 4 * CST_STAGE - 1)
        # selection of trip r'' in the new s_stn stage
        p.s_add_u32(t0, self.s_cq, 1)
        p.s_cmp("ge_u32", t0, P("nq"))
        p.s_cselect(t0, 0, t0)
        self.emit_wave_state(p, t0, self.s_stn, self.s_cstn, self.s_stnw, self.s_cstnw, None)
        if "dma_b" in self.ablate:        # (timing only: without the B pieces)
            p.items = [it for it in p.items if not (it.kind == "dma" and it.src[1] == self.d_x)]
        self.apply_ablate(p)
        return p
