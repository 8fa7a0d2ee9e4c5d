"""Generator of the hand-placed dQ backward kernel body (gfx950, head dim 128, bf16 / f16).

Replaces _sink_flash_attn_bwd_dq_kernel of the reference (sink_attention/sink_flash_attention.py:371-484); same maths
as csrc/sfa_bwd_mfma.hip's bwd_dq_mfma_kernel (Q-stationary, S and dP recomputed), different machine mapping:

  workgroup = 4 waves; wave w = 64 query rows (two 32-row blocks rb) of ONE q head: HPW = gcd(group, 4) heads of the
  GQA group x 4 / HPW row groups share every 64-key K / V tile through LDS.  One wave per SIMD, 512 registers:
      Q and dO fragments of the wave's rows (B operands) live in 128 accumulator registers for the whole workgroup,
      dQ^T [d, row] accumulates in the other 128.
  Per tile (64 keys = two 32-key halves kh) and wave, 96 MFMAs:
      S^T[key,row]  = K Q^T          4 chains x 8   (A: K row fragments from LDS, 8 fragments feed both row blocks)
      dP^T[key,row] = V dO^T         4 chains x 8   (A: V row fragments from LDS)
      P = exp2(c S - LSE log2e), dS = P (dP - Delta)   VALU on 16 elements per lane and chain, packed in place
      dQ^T[d,row]  += K^T dS^T       32             (A: transposed LDS reads of the SAME K image, B: packed dS)
  Spine order S00 S01 dP00 dP01 S10 S11 dP10 dP11 dQ(kh0) dQ(kh1): the VALU work of a chain pair runs in the gaps of the
  next 16 MFMAs.  K / V tiles arrive by LDS-DMA three tiles ahead into a 4-deep ring (8 pieces per wave and tile); one
  s_barrier per tile.

LDS map: 4 stages x (K image 16 KB | V image 16 KB) from byte 0; images are the dual-use 8-row x 32-column subtile
layout of dkdv.py.
"""
from __future__ import annotations

from .core import A, Imm, Instr, M0, P, PV, Prog, Reg, S, V, VCC, imm
from .dkdv import Alloc
from .sched import finish_block, fix_hazards, insert_waits, schedule
from .worklist import WorkList

STG_BYTES = 32768
NSTAGE = 4
LDS_BYTES = NSTAGE * STG_BYTES

# ---- persistent (work-list) form: the workgroup walks a list of work items; the HIP shell writes one 128-byte descriptor
# per item into LDS behind the tile ring, the body reads the fields it needs with uniform ds_reads + v_readfirstlane.
DESC_BASE = LDS_BYTES          # byte 131072 (tools/asmgen/worklist.py: descriptor size, items per invocation)
# descriptor dwords; the K / V stream part (16 ..) has the same layout in the forward kernel's descriptor
DESC = {"q_lo": 0, "q_hi": 1, "do_lo": 2, "do_hi": 3, "lse_lo": 4, "lse_hi": 5, "dl_lo": 6, "dl_hi": 7,
        "dq_lo": 8, "dq_hi": 9, "q0": 10, "nrows": 11, "q_rng": 12, "do_rng": 13, "dq_rng": 14,
        "k_lo": 16, "k_hi": 17, "v_lo": 18, "v_hi": 19, "k_rng": 20, "v_rng": 21, "nt": 22, "ts_hi": 23, "tw_off": 24}
PARAMS_PK = ["q_hs", "q_sn", "do_hs", "do_sn", "dq_hs", "dq_sn", "k_sn", "v_sn", "ld_hs", "pos0", "W", "ns",
             "hpw_log2", "c_log2", "nlog2e", "scale", "n_items"]

PARAMS = [
    "q_lo", "q_hi", "q_hs", "q_sn", "q_rng",
    "do_lo", "do_hi", "do_hs", "do_sn", "do_rng",
    "dq_lo", "dq_hi", "dq_hs", "dq_sn", "dq_rng",
    "k_lo", "k_hi", "k_sn", "k_rng", "v_lo", "v_hi", "v_sn", "v_rng",
    "lse_lo", "lse_hi", "dl_lo", "dl_hi", "ld_hs",          # LSE / Delta [.., head, row] f32; head stride in bytes
    "q0", "nrows", "pos0", "W", "ns", "nt", "ts_hi", "tw_off",     # tile it -> it < ts_hi ? it : it + tw_off
    "hpw_log2", "c_log2", "nlog2e", "scale",
]
# q_*/do_*/dq_*/lse/dl bases point at (first head of the workgroup's head set, the sequence's first row); q0 = first row
# of the workgroup; nlog2e = -log2(e) (f32 bits).


class DqGen(WorkList):
    DESC, DESC_BASE = DESC, DESC_BASE

    def __init__(self, dtype="bf16", sched=True, vfirst=4, sfirst=None, npool=12, D=128, ablate=(), dma_t0=40, dma_dt=120, persist=True, stamps=False, dead=True, wide64=False, sinkfar=True, edge_subs=True):
        assert dtype in ("bf16", "f16") and D in (64, 80, 96, 128)
        self.dtype, self.do_sched = dtype, sched
        self.persist = persist
        if sfirst is None:
            sfirst = (44 if stamps else 48) if persist else 56
        self.ablate = set(ablate)         # timing-only knock-out builds (wrong results)
        self.dead = dead                  # tile class 3 and its body
        self.sinkfar = sinkfar and persist   # class 4: a sink tile whose second key half nobody sees (set at item init)
        self.edge_subs = edge_subs        # edge tiles: a 32-key x 32-row sub-block that no row sees is left out (classes 5, 6)
        # the bodies compute "full or not" only (one compare), the selector the rest: measured no faster (C3 dQ 1.861 vs 1.857,
        # 1.845 vs 1.852 ms, same box), off
        self.range_cls = False
        # row stores of 64 contiguous bytes (4 lanes per row, 16 rows per instruction): an experiment that did not pay - C3 dQ
        # 1.9181 vs 1.9195 ms, W = 512 0.4831 vs 0.4858, bitwise equal (profiles/r03_ab_dq_wide64.log): the stores of an item
        # transition are not bound by the cache lines an instruction touches
        self.wide64 = wide64 and persist
        self.dma_t0, self.dma_dt = dma_t0, dma_dt   # deadlines of the eight LDS-DMA pieces inside a trip (cycles of the model)
        # head dim: DK k-steps of 16, DB 32-wide output blocks, NCH valid 16-byte chunks per row (LDS rows stay 256 bytes:
        # chunks beyond the head dim are fetched as zeros or, when a whole 128-byte half is padding, not at all)
        self.D, self.DK, self.DB, self.NCH = D, D // 16, (D + 31) // 32, D // 8
        self.HALVES = 2 if D > 64 else 1
        self.vfirst, self.sfirst = vfirst, sfirst
        va = self.va = Alloc("v", vfirst, 255)
        sa = self.sa = Alloc("s", sfirst, 99)
        # ---------------- VGPRs
        self.SACC = [[va("sacc%d%d" % (kh, rb), 16, 4) for rb in range(2)] for kh in range(2)]
        self.DPACC = [[va("dpacc%d%d" % (kh, rb), 16, 4) for rb in range(2)] for kh in range(2)]
        self.POOL = [va("pool%d" % i, 4, 4) for i in range(npool)]
        self.lse2 = [va("lse2_%d" % rb) for rb in range(2)]       # -LSE * log2(e) of the lane's row
        self.nd = [va("nd_%d" % rb) for rb in range(2)]           # Delta of the lane's row
        self.lane, self.lane31 = va("lane"), va("lane31")
        self.l_row_e, self.l_tr0 = va("l_row_e"), va("l_tr0")
        self.a_k_e, self.a_k_o, self.a_v_e, self.a_v_o = va("a_k_e"), va("a_k_o"), va("a_v_e"), va("a_v_o")
        self.a_tr0, self.a_tr1 = va("a_tr0"), va("a_tr1")
        self.a_kn_e, self.a_kn_o = va("a_kn_e"), va("a_kn_o")
        self.l_dma = [[va("l_dma%d%s" % (e, t)) for t in "kv"] for e in range(2)]
        # second 128-byte half of the rows: an out-of-range offset for lanes whose chunk lies beyond the head dim
        self.l_dma1 = [[va("l_dma1_%d%s" % (e, t)) for t in "kv"] for e in range(2)] if 64 < D < 128 else None
        self.vt = [va("vt%d" % i) for i in range(2)]
        self.v_oob = va("v_oob")               # DMA source offsets of the piece being issued
        self.v_pos = [va("v_pos%d" % rb) for rb in range(2)]       # key position of the lane's row
        self.v_d = [va("v_d%d" % rb) for rb in range(2)]
        self.v_w, self.v_2e31, self.v_nsh, self.v_weff = va("v_w"), va("v_2e31"), va("v_nsh"), va("v_weff")
        self.tmp = [va("tmp%d" % i) for i in range(4)]
        self.vo = [va("vo%d" % rb) for rb in range(2)]             # row offsets (prologue loads, epilogue stores)
        # ---------------- AGPRs
        self.QF = [[A((rb * 8 + ks) * 4, 4) for ks in range(self.DK)] for rb in range(2)]
        self.DOF = [[A(64 + (rb * 8 + ks) * 4, 4) for ks in range(self.DK)] for rb in range(2)]
        self.DQ = [[A(128 + (rb * 4 + db) * 16, 16) for db in range(self.DB)] for rb in range(2)]
        # ---------------- SGPRs
        self.d_k, self.d_v, self.d_x = sa("d_k", 4, 4), sa("d_v", 4, 4), sa("d_x", 4, 4)
        self.s_wave, self.s_hh, self.s_rgi = sa("s_wave"), sa("s_hh"), sa("s_rgi")
        self.s_pw0, self.s_pwhi = sa("s_pw0"), sa("s_pwhi")
        self.s_flo, self.s_frng = sa("s_flo"), sa("s_frng")        # tiles starting in [flo, flo + frng) are "full" for this wave
        self.s_it, self.s_k0 = sa("s_it"), sa("s_k0")
        self.s_st, self.s_stn, self.s_std = sa("s_st"), sa("s_stn"), sa("s_std")
        self.s_koff, self.s_voff = sa("s_koff"), sa("s_voff")
        self.s_wofs = sa("s_wofs")
        self.s_cls = sa("s_cls")
        # offsets of the 64-byte row stores: rows l and l + 16 of the two row blocks (the mask scratch of the tile bodies is
        # idle at a transition)
        self.vo64, self.vo64b = self.v_d, [self.v_nsh, self.v_weff]
        self.s_tmp = [sa("s_tmp%d" % i) for i in range(5)]
        if persist:
            self.wl_alloc(sa)
        self.stamps = stamps and persist          # diagnostic build (tools/stamps_wl.py --phases): where an item transition goes
        if self.stamps:
            self.s_tt = sa("s_tt", 2, 2)
            self.s_T0 = sa("s_T0")
            self.s_acc = [sa("s_acc%d" % i) for i in range(3)]
        # per-item tile-list scalars: SGPRs of the work-list form, asm inputs otherwise
        self.r_nt = self.s_nt if persist else P("nt")
        self.r_ts_hi = self.s_ts_hi if persist else P("ts_hi")
        self.r_tw_off = self.s_tw_off if persist else P("tw_off")
        self.pool_next = 0

    def params(self):
        return list(PARAMS_PK if self.persist else PARAMS) + (["dbg_lo", "dbg_hi", "bid"] if self.stamps else [])

    def emit_stamp(self, p: Prog, k: int):
        """k = -1: the reference time (item's loop ended); k >= 0: add the time since then to accumulator k"""
        if not self.stamps:
            return
        p.add(Instr("s_memtime", [self.s_tt], [], kind="fence"))
        p.s_waitcnt(lgkmcnt=0)
        if k < 0:
            p.s_mov(self.s_T0, self.s_tt[0])
        else:
            p.s_sub_u32(self.s_tmp[0], self.s_tt[0], self.s_T0)
            p.s_add_u32(self.s_acc[k], self.s_acc[k], self.s_tmp[0])

    def pool(self):
        r = self.POOL[self.pool_next % len(self.POOL)]
        self.pool_next += 1
        return r

    # ------------------------------------------------------------------ pieces
    def emit_tile_of(self, p: Prog, dst, it):
        """dst = key-tile index of iteration `it` (sink tiles first, then the window tiles)"""
        p.s_add_u32(dst, it, self.r_tw_off)
        p.s_cmp("lt_u32", it, self.r_ts_hi)
        p.s_cselect(dst, it, dst)

    def emit_dma_tile(self, p: Prog, it_reg, spread=False):
        """LDS-DMA of the K and V images of iteration it_reg's tile into stage s_std: this wave's 4 + 4 pieces (row
        groups 2 wave, 2 wave + 1; two 128-byte halves each).  Iterations past the last tile fetch through zero-record
        descriptors (nothing is read)."""
        t = self.s_tmp
        self.emit_tile_of(p, t[0], it_reg)
        p.s_lshl_b32(t[0], t[0], 6)                        # first key of the tile
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

    def emit_k_prefetch(self, p: Prog, e, o, deadline=None):
        """first four K row fragments (key half 0, k-steps 0..3) of the next tile, into pool slots 0..3"""
        for ks in range(4):      # (DK >= 4 for every supported head dim)
            base = o if ks & 1 else e
            ins = p.ds_read_b128(self.POOL[ks], base, 512 * (ks >> 1), mem=("stage_r",))
            if deadline is not None:
                ins.mods["alap"] = deadline + 6 * ks

    def emit_class(self, p: Prog):
        """s_cls: 0 full, 1 edge (no sink key in the tile), 2 edge with sink keys.  full <=> k0 + 63 <= pw0 and
        (k0 + 63 < ns or k0 >= pw_hi - W + 1)"""
        t = self.s_tmp
        if self.range_cls:
            # full <=> pwhi - W + 1 <= k0 <= pw0 - 63 (every key causal for and inside the window of every row): one unsigned
            # compare against two per-item constants; the selector behind the loop head sorts out the tiles that are not full
            p.s_sub_u32(t[0], self.s_k0, self.s_flo)
            p.s_cmp("lt_u32", t[0], self.s_frng)
            p.s_cselect(self.s_cls, 0, 1)
            return
        p.s_add_u32(t[0], self.s_k0, 63)
        p.s_cmp("le_i32", t[0], self.s_pw0)
        p.s_cselect(t[1], 1, 0)
        p.s_cmp("lt_i32", t[0], P("ns"))
        p.s_cselect(t[2], 1, 0)
        p.s_sub_i32(t[0], self.s_pwhi, P("W"))
        p.s_add_i32(t[0], t[0], 1)
        p.s_cmp("ge_i32", self.s_k0, t[0])
        p.s_cselect(t[0], 1, 0)
        p.s_or_b32(t[0], t[0], t[2])
        p.s_and_b32(t[0], t[0], t[1])                      # full
        p.s_cmp("lt_i32", self.s_k0, P("ns"))
        p.s_cselect(t[1], 2, 1)
        p.s_cmp("lg_u32", t[0], 0)
        p.s_cselect(self.s_cls, 0, t[1])
        if self.dead and not self.edge_subs:       # (with edge_subs the selector behind the loop head finds the dead tiles)
            # 3: no row of the wave sees any key of the tile - every key lies behind every row (k0 > pwhi), or the tile holds
            # no sink key and every key has left every row's window (k0 + 63 <= pw0 - W): waves that own different ROWS
            # (MHA, groups of 2) walk tiles that only the other waves' rows can see (fwd.py, class 4)
            p.s_cmp("gt_i32", self.s_k0, self.s_pwhi)
            p.s_cselect(t[0], 1, 0)
            p.s_sub_i32(t[1], self.s_pw0, P("W"))
            p.s_add_u32(t[2], self.s_k0, 63)
            p.s_cmp("le_i32", t[2], t[1])
            p.s_cselect(t[1], 1, 0)
            p.s_cmp("ge_i32", self.s_k0, P("ns"))
            p.s_cselect(t[2], 1, 0)
            p.s_and_b32(t[1], t[1], t[2])
            p.s_or_b32(t[0], t[0], t[1])
            p.s_cmp("lg_u32", t[0], 0)
            p.s_cselect(self.s_cls, 3, self.s_cls)

    # ------------------------------------------------------------------ prologue
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
        # wave -> head hh = wave & (HPW - 1), row group rgi = wave >> log2(HPW)
        p.s_lshl_b32(st[0], 1, P("hpw_log2"))
        p.s_sub_u32(st[0], st[0], 1)
        p.s_and_b32(self.s_hh, wv, st[0])
        p.s_lshr_b32(self.s_rgi, wv, P("hpw_log2"))
        # lane parts of the LDS addresses (see dkdv.py)
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
        # ---- K / V streams: descriptors and lane parts of the source offsets.  piece of row group 2 wave + e: rows
        #      16 wave + 8 e + rr, chunk 4 cbl + (slot ^ ((2 e + (rr >> 2)) & 3))   (+ 8 chunks for the second half)
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
            p.v_lshl_add_u32(t3, t2, 2, t3)                   # + 4 cbl   (t2 = lane >> 5)
            p.v_lshlrev(t3, 4, t3)                            # bytes inside the row
            p.s_add_u32(st[1], st[0], 8 * e)
            p.v_add_u32(self.vt[0], st[1], rr)                # row inside the tile
            for col, nm in ((0, "k"), (1, "v")):
                p.v_mul_lo_u32(self.l_dma[e][col], self.vt[0], P(nm + "_sn"))
                p.v_add_u32(self.l_dma[e][col], self.l_dma[e][col], t3)
                if self.l_dma1 is not None:      # chunk 8 + (t3 >> 4) of the row must be < NCH
                    p.v_lshrrev(self.vt[1], 4, t3)
                    p.v_add_u32(self.l_dma1[e][col], 128, self.l_dma[e][col])
                    p.v_cmp("gt_u32", self.NCH - 8, self.vt[1])
                    p.v_cndmask(self.l_dma1[e][col], self.v_oob, self.l_dma1[e][col])
        p.s_lshl_b32(self.s_wofs, wv, 12)                     # 4096 wave: the wave's four pieces inside an image
        # requests in the order their data is needed: tile 0, the Q / dO fragments and row constants, tiles 1 and 2; the
        # accumulators and constants are set up while they are in flight
        p.s_mov(self.s_std, 0)
        p.s_mov(st[3], 0)
        self.emit_dma_tile(p, st[3])
        # row of the lane in block rb: qw0 + 32 rb + r ; qw0 = q0 + 64 rgi
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
        # ---- Q and dO fragments of the lane's rows: B operands, chunk 2 ks + h of the row -> accumulator registers
        for nm, frags in (("q", self.QF), ("do", self.DOF)):
            p.s_mul_i32(st[1], self.s_hh, P(nm + "_hs"))
            p.s_mul_hi_u32(st[2], self.s_hh, P(nm + "_hs"))
            p.s_add_u32(self.d_x[0], P(nm + "_lo"), st[1])
            p.s_addc_u32(self.d_x[1], P(nm + "_hi"), st[2])
            p.s_mov(self.d_x[2], P(nm + "_rng"))
            p.s_mov(self.d_x[3], 0x00020000)
            p.v_mul_lo_u32(t1, t0, P(nm + "_sn"))
            p.v_lshl_add_u32(self.vo[0], t2, 4, t1)           # + 16 h
            p.s_lshl_b32(st[1], P(nm + "_sn"), 5)
            p.v_add_u32(self.vo[1], st[1], self.vo[0])
            for rb in range(2):
                for ks in range(self.DK):
                    p.buffer_load(frags[rb][ks], self.vo[rb], self.d_x, 0, offset=32 * ks)
        # ---- row constants: -LSE log2(e) and Delta (rows >= nrows read 0)
        p.s_mul_i32(st[1], self.s_hh, P("ld_hs"))
        p.s_mul_hi_u32(st[2], self.s_hh, P("ld_hs"))
        p.v_lshlrev(t1, 2, t0)                                # row * 4
        p.v_add_u32(t3, 128, t1)
        for nm, dst in (("lse", self.lse2), ("dl", self.nd)):
            p.s_add_u32(self.d_x[0], P(nm + "_lo"), st[1])
            p.s_addc_u32(self.d_x[1], P(nm + "_hi"), st[2])
            p.s_lshl_b32(self.d_x[2], P("nrows"), 2)
            p.s_mov(self.d_x[3], 0x00020000)
            p.buffer_load(dst[0], t1, self.d_x, 0)
            p.buffer_load(dst[1], t3, self.d_x, 0)
        for j in (1, 2):
            p.s_mov(self.s_std, j * STG_BYTES)
            p.s_mov(st[3], j)
            self.emit_dma_tile(p, st[3])
        # ---- mask constants
        p.v_mov(self.v_w, P("W"))
        p.v_mov(self.v_2e31, imm(0x80000000))
        # ---- accumulators
        for rb in range(2):
            for db in range(self.DB):
                for i in range(16):
                    p.v_accvgpr_write(self.DQ[rb][db][i], 0)
        p.s_waitcnt(vmcnt=2 * 4 * self.HALVES, note="Q / dO fragments, row constants, tile 0 landed (tiles 1, 2 in flight)")
        p.s_barrier()
        for rb in range(2):
            p.v_mul_f32(self.lse2[rb], P("nlog2e"), self.lse2[rb])
        p.v_mov(self.a_kn_e, self.l_row_e)
        p.v_xor(self.a_kn_o, 32, self.a_kn_e)
        self.emit_k_prefetch(p, self.a_kn_e, self.a_kn_o)
        p.s_mov(self.s_it, 0)
        self.emit_tile_state(p, self.s_it)
        p.s_mov(self.s_st, 0)
        p.s_mov(self.s_stn, STG_BYTES)
        p.s_mov(self.s_std, 3 * STG_BYTES)
        return p

    # ------------------------------------------------------------------ loop head
    def emit_tile_state(self, p: Prog, it):
        """first key and mask class of the tile of iteration `it`"""
        t = self.s_tmp
        self.emit_tile_of(p, t[3], it)
        p.s_lshl_b32(self.s_k0, t[3], 6)
        self.emit_class(p)

    def loop_top(self) -> Prog:
        p = Prog()
        p.label("L_top%=")
        if self.persist:
            self.emit_stream_advance(p)
        p.s_cmp("ge_u32", self.s_it, self.r_nt)
        p.s_cbranch("scc1", "L_done%=")
        p.s_waitcnt(vmcnt=4 * self.HALVES, note="tile it+1 landed (own pieces); tile it+2 may be in flight")
        p.label("L_top_b%=")
        p.s_barrier()
        p.s_waitcnt(lgkmcnt=0, note="the K fragments fetched at the end of the last trip")
        p.s_cmp("eq_u32", self.s_cls, 0)
        p.s_cbranch("scc1", "L_full%=")
        p.s_cmp("eq_u32", self.s_cls, 1)
        p.s_cbranch("scc1", "L_edgesel%=" if self.edge_subs else "L_edge%=")
        if self.dead and not self.edge_subs:
            p.s_cmp("eq_u32", self.s_cls, 3)
            p.s_cbranch("scc1", "L_dead%=")
        if self.sinkfar:
            p.s_cmp("eq_u32", self.s_cls, 4)
            p.s_cbranch("scc1", "L_sinkfar%=")
        p.s_branch("L_sink%=")
        return p

    # ------------------------------------------------------------------ one tile
    def tile_body(self, cls: int) -> Prog:
        """cls: 0 full, 1 edge, 2 edge with sink keys, 3 dead (the wave keeps the K / V stream and its state going, computes
        nothing)"""
        p = Prog()
        dt = self.dtype
        self.pool_next = 0
        st = self.s_tmp
        if cls == 3:
            p.v_add_u32(self.a_kn_e, self.s_stn, self.l_row_e)
            p.v_xor(self.a_kn_o, 32, self.a_kn_e)
            if self.persist:
                self.emit_dma_stream_tile(p, spread=False)
            else:
                p.s_add_u32(st[4], self.s_it, 3)
                self.emit_dma_tile(p, st[4], spread=False)
            self.emit_k_prefetch(p, self.a_kn_e, self.a_kn_o)
            self.emit_tile_advance(p)
            return p
        halves = (0,) if cls == 4 else (0, 1)      # class 4: the sink tile without its second key half
        if cls == 4:
            cls = 2
        skip = set()                               # (key half, row block) sub-blocks that no row sees: classes 5 / 6 of edge tiles
        if cls == 5:
            cls, skip = 1, {(1, 0)}                # the diagonal tile: rows 0..31 never see keys 32..63
        elif cls == 6:
            cls, skip = 1, {(0, 1)}                # the window's first tile: rows 32..63 have left keys 0..31 behind
        p.v_add_u32(self.a_k_e, self.s_st, self.l_row_e)
        p.v_xor(self.a_k_o, 32, self.a_k_e)
        p.v_add_u32(self.a_v_e, 16384, self.a_k_e)
        p.v_xor(self.a_v_o, 32, self.a_v_e)
        p.v_add_u32(self.a_tr0, self.s_st, self.l_tr0)
        p.v_xor(self.a_tr1, 32, self.a_tr0)
        p.v_add_u32(self.a_kn_e, self.s_stn, self.l_row_e)
        p.v_xor(self.a_kn_o, 32, self.a_kn_e)
        # fetch tile it + 3 (work-list form: the next tile of the K / V stream, which may belong to the next item)
        if self.persist:
            self.emit_dma_stream_tile(p, spread=True)
        else:
            p.s_add_u32(st[4], self.s_it, 3)
            self.emit_dma_tile(p, st[4], spread=True)
        if cls:
            for rb in range(2):      # (pos - k0 - 4 h): u = this - (32 kh + o_v) = pos - key
                p.v_lshrrev(self.tmp[1], 5, self.lane)
                p.v_lshlrev(self.tmp[1], 2, self.tmp[1])
                p.v_add_u32(self.tmp[1], self.s_k0, self.tmp[1])
                p.v_sub_u32(self.v_d[rb], self.v_pos[rb], self.tmp[1])
            if cls == 2:             # sink key <=> 32 kh + o_v < ns - k0 - 4 h
                p.v_sub_u32(self.v_nsh, P("ns"), self.tmp[1])
        for kh in halves:
            # ---- S^T = K Q^T: the eight K row fragments of this key half feed both row blocks
            kf = []
            for ks in range(self.DK):
                f = self.pool()
                if not (kh == 0 and ks < 4):          # the first four were fetched at the end of the previous trip
                    base = self.a_k_o if ks & 1 else self.a_k_e
                    p.ds_read_b128(f, base, 8192 * kh + 512 * (ks >> 1), mem=("stage_r",), note="K rows")
                kf.append(f)
            for rb in range(2):
                if (kh, rb) in skip:
                    continue
                for ks in range(self.DK):
                    p.mfma(dt, self.SACC[kh][rb], kf[ks], self.QF[rb][ks], self.SACC[kh][rb] if ks else 0, tag="S")
            # ---- dP^T = V dO^T
            vf = []
            for ks in range(self.DK):
                f = self.pool()
                base = self.a_v_o if ks & 1 else self.a_v_e
                p.ds_read_b128(f, base, 8192 * kh + 512 * (ks >> 1), mem=("stage_r",), note="V rows")
                vf.append(f)
            for rb in range(2):
                if (kh, rb) in skip:
                    continue
                for ks in range(self.DK):
                    p.mfma(dt, self.DPACC[kh][rb], vf[ks], self.DOF[rb][ks], self.DPACC[kh][rb] if ks else 0, tag="dP")
            # ---- P, dS, packed in place
            for rb in range(2):
                if (kh, rb) in skip:
                    continue
                for v in range(16):
                    x, y = self.SACC[kh][rb][v], self.DPACC[kh][rb][v]
                    if "fma" not in self.ablate:
                        p.v_fma_f32(x, x, P("c_log2"), self.lse2[rb])
                    p.v_exp_f32(x, x)
                    if cls:
                        c = 32 * kh + (v & 3) + 8 * (v >> 2)
                        p.v_sub_u32(self.tmp[0], self.v_d[rb], c)
                        if cls == 2:
                            p.v_cmp("lt_i32", c, self.v_nsh)
                            p.v_cndmask(self.v_weff, self.v_w, self.v_2e31)
                            p.v_cmp("lt_u32", self.tmp[0], self.v_weff)
                        else:
                            p.v_cmp("lt_u32", self.tmp[0], self.v_w)
                        p.v_cndmask(x, 0, x)
                    p.v_sub_f32(y, y, self.nd[rb])
                    p.v_mul_f32(y, x, y)
                for s in range(2):
                    for j in range(4):
                        d = self.DPACC[kh][rb]
                        p.v_cvt_pk(dt, d[4 * s + j], d[8 * s + 2 * j], d[8 * s + 2 * j + 1])
        # ---- dQ^T += K^T dS^T (K^T fragments: transposed reads of the K image, rows = keys)
        for kh in halves:
            for s in range(2):
                for db in range(self.DB):
                    f = self.pool()
                    off = 8192 * kh + 512 * db
                    p.ds_read_b64_tr_b16(f[0:2], self.a_tr0, off + 2048 * (2 * s), mem=("stage_r",))
                    p.ds_read_b64_tr_b16(f[2:4], self.a_tr1, off + 2048 * (2 * s + 1), mem=("stage_r",))
                    for rb in range(2):
                        if (kh, rb) not in skip:
                            p.mfma(dt, self.DQ[rb][db], f, self.DPACC[kh][rb][4 * s:4 * s + 4], self.DQ[rb][db], tag="dQ")
        # first K fragments of the next tile (landed before this trip's barrier)
        self.emit_k_prefetch(p, self.a_kn_e, self.a_kn_o, deadline=max(200, (8 * self.DK + 8 * self.DB) * 16 * len(halves) - 300))
        self.emit_tile_advance(p)
        return p

    def edge_selector(self) -> Prog:
        """behind the loop head, edge tiles only: is a 32-key x 32-row sub-block out of reach?  (kh 1, rb 0): every key of the
        second half lies behind rows pw0 .. pw0 + 31 (k0 >= pw0: the diagonal tile); (kh 0, rb 1): every key of the first half
        has left the window of rows pw0 + 32 .. (k0 + 31 <= pw0 + 32 - W: the window's first tile; class 1 tiles hold no sink
        key).  Both at once (windows below 64): the generic edge body."""
        p = Prog()
        t = self.s_tmp
        p.label("L_edgesel%=")
        if self.range_cls:
            # a tile with sink keys: all of them sinks and causal for every row - full after all; else the sink-edge body
            p.s_cmp("lt_i32", self.s_k0, P("ns"))
            p.s_cbranch("scc0", "L_selns%=")
            p.s_add_u32(t[2], self.s_k0, 63)
            p.s_cmp("lt_i32", t[2], P("ns"))
            p.s_cbranch("scc0", "L_sink%=")
            p.s_cmp("le_i32", t[2], self.s_pw0)
            p.s_cbranch("scc1", "L_full%=")
            p.s_branch("L_sink%=")
            p.label("L_selns%=")
        if self.dead:
            # no row of the wave sees any key of the tile: every key lies behind every row (k0 > pwhi), or every key has left
            # every row's window (k0 + 63 <= pw0 - W; class 1 tiles hold no sink key) - waves that own different ROWS (MHA,
            # groups of 2) walk tiles that only the other waves' rows can see
            p.s_cmp("gt_i32", self.s_k0, self.s_pwhi)
            p.s_cbranch("scc1", "L_dead%=")
            p.s_sub_i32(t[1], self.s_pw0, P("W"))
            p.s_add_u32(t[2], self.s_k0, 63)
            p.s_cmp("le_i32", t[2], t[1])
            p.s_cbranch("scc1", "L_dead%=")
        p.s_cmp("ge_i32", self.s_k0, self.s_pw0)
        p.s_cselect(t[0], 1, 0)
        p.s_sub_i32(t[1], self.s_pw0, P("W"))
        p.s_add_i32(t[1], t[1], 1)
        p.s_cmp("le_i32", self.s_k0, t[1])
        p.s_cselect(t[1], 1, 0)
        p.s_cmp("lg_u32", t[0], t[1])                  # exactly one of the two
        p.s_cbranch("scc0", "L_edge%=")
        p.s_cmp("lg_u32", t[0], 0)
        p.s_cbranch("scc1", "L_edge_d%=")
        p.s_branch("L_edge_w%=")
        return p

    def emit_tile_advance(self, p: Prog):
        """next trip: ring offsets, first key and class of its tile (the masks of the current tile read k0 first)"""
        st = self.s_tmp
        p.s_add_u32(self.s_it, self.s_it, 1)
        t0 = st[3]
        p.s_mov(self.s_st, self.s_stn)
        p.s_add_u32(t0, self.s_stn, STG_BYTES)
        p.s_and_b32(self.s_stn, t0, LDS_BYTES - 1)
        p.s_add_u32(t0, self.s_std, STG_BYTES)
        p.s_and_b32(self.s_std, t0, LDS_BYTES - 1)
        self.emit_tile_state(p, self.s_it)

    # ------------------------------------------------------------------ epilogue
    def epilogue(self) -> Prog:
        p = Prog()
        dt = self.dtype
        t0, t1, t2, t3 = self.tmp
        st = self.s_tmp
        p.label("L_done%=")
        p.s_waitcnt(vmcnt=0, lgkmcnt=0)
        # dQ[row, d] = scale * dQ^T[d, row]: lane = row, registers 4 g4 + e <-> d = 32 db + 8 g4 + 4 h + e
        p.s_mul_i32(st[1], self.s_hh, P("dq_hs"))
        p.s_mul_hi_u32(st[2], self.s_hh, P("dq_hs"))
        p.s_add_u32(self.d_x[0], P("dq_lo"), st[1])
        p.s_addc_u32(self.d_x[1], P("dq_hi"), st[2])
        p.s_mov(self.d_x[2], P("dq_rng"))
        p.s_mov(self.d_x[3], 0x00020000)
        p.v_sub_u32(t0, self.v_pos[0], P("pos0"))             # row, rb = 0
        p.v_mul_lo_u32(t1, t0, P("dq_sn"))
        p.v_lshrrev(t2, 5, self.lane)
        p.v_lshl_add_u32(self.vo[0], t2, 4, t1)               # + 16 h bytes
        p.s_lshl_b32(st[1], P("dq_sn"), 5)
        p.v_add_u32(self.vo[1], st[1], self.vo[0])
        # groups k / k+1 exchanged between the half-waves: 16 contiguous bytes per lane, one dwordx4 store per pair (as
        # the forward's epilogue)
        npair = 0
        for rb in range(2):
            for db in range(self.DB):
                for gp in range(2):
                    if 32 * db + 16 * gp >= self.D:
                        continue                              # padding columns of the last block
                    X, Y = self.POOL[(2 * npair) % 8], self.POOL[(2 * npair + 1) % 8]
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
                    p.buffer_store(X[0:4], self.vo[rb], self.d_x, 0, offset=64 * db + 32 * gp)
        p.s_waitcnt(vmcnt=0)
        return p


    # ================================================================== work-list (persistent) form
    def setup_pk(self) -> Prog:
        """once per workgroup: lane constants, wave -> (head, row group), K / V stream offsets"""
        p = Prog()
        t0, t1, t2, t3 = self.tmp
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
        p.v_mov(self.v_w, P("W"))
        p.v_mov(self.v_2e31, imm(0x80000000))
        return p

    def item_land(self):
        """scratch VGPR quads the descriptor groups of the item being opened land in (the S^T / dP^T tiles: dead between
        two items)"""
        regs = [self.SACC[kh][rb][4 * i:4 * i + 4] for kh in range(2) for rb in range(2) for i in range(4)]
        return {g: regs[g] for g in range(8)}

    def emit_item_begin(self, p: Prog):
        """open item s_item: tile-list scalars, row positions, and the requests for its Q / dO fragments and row
        constants (into registers that are dead once the previous item's last tile is done)"""
        land = self.item_land()
        t0, t1, t2, t3 = self.tmp
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
        vl = [self.a_k_e, self.a_k_o]                         # load offsets (the bodies recompute these registers)
        for nm, frags in (("q", self.QF), ("do", self.DOF)):
            self.desc_get(p, self.d_y[0], land, nm + "_lo")
            self.desc_get(p, self.d_y[1], land, nm + "_hi")
            p.s_mul_i32(st[1], self.s_hh, P(nm + "_hs"))
            p.s_mul_hi_u32(st[2], self.s_hh, P(nm + "_hs"))
            p.s_add_u32(self.d_y[0], self.d_y[0], st[1])
            p.s_addc_u32(self.d_y[1], self.d_y[1], st[2])
            self.desc_get(p, self.d_y[2], land, nm + "_rng")
            p.s_mov(self.d_y[3], 0x00020000)
            p.v_mul_lo_u32(t1, t0, P(nm + "_sn"))
            p.v_lshl_add_u32(vl[0], t2, 4, t1)                # + 16 h
            p.s_lshl_b32(st[1], P(nm + "_sn"), 5)
            p.v_add_u32(vl[1], st[1], vl[0])
            for rb in range(2):
                for ks in range(self.DK):
                    p.buffer_load(frags[rb][ks], vl[rb], self.d_y, 0, offset=32 * ks)
        # row constants: Delta, then LSE (the LAST loads: waiting for them covers every request of the item)
        p.s_mul_i32(st[1], self.s_hh, P("ld_hs"))
        p.s_mul_hi_u32(st[2], self.s_hh, P("ld_hs"))
        p.v_lshlrev(t1, 2, t0)                                # row * 4
        p.v_add_u32(t3, 128, t1)
        for nm, dst in (("dl", self.nd), ("lse", self.lse2)):
            self.desc_get(p, self.d_y[0], land, nm + "_lo")
            self.desc_get(p, self.d_y[1], land, nm + "_hi")
            p.s_add_u32(self.d_y[0], self.d_y[0], st[1])
            p.s_addc_u32(self.d_y[1], self.d_y[1], st[2])
            p.s_lshl_b32(self.d_y[2], st[4], 2)
            p.s_mov(self.d_y[3], 0x00020000)
            p.buffer_load(dst[0], t1, self.d_y, 0)
            p.buffer_load(dst[1], t3, self.d_y, 0)

    def emit_item_init(self, p: Prog):
        """accumulators, row constants in the exp2 domain (waits for every request of emit_item_begin), first K fragments
        of the item's first tile (ring slot s_st: landed and barrier-visible), tile state"""
        for rb in range(2):
            for db in range(self.DB):
                for i in range(16):
                    p.v_accvgpr_write(self.DQ[rb][db][i], 0)
        for rb in range(2):
            p.v_mul_f32(self.lse2[rb], P("nlog2e"), self.lse2[rb])
        p.v_add_u32(self.a_kn_e, self.s_st, self.l_row_e)
        p.v_xor(self.a_kn_o, 32, self.a_kn_e)
        self.emit_k_prefetch(p, self.a_kn_e, self.a_kn_o)
        p.s_mov(self.s_it, 0)
        self.emit_tile_state(p, self.s_it)
        if self.sinkfar:
            # The item's FIRST tile, when it is a sink tile (class 2) whose second 32-key half holds no sink key and lies
            # outside every row's window (rows far behind the sinks: every item but the first few): class 4, the body
            # without that half.  Decided here, once per item - the loop's own classification never yields 4.
            t = self.s_tmp
            p.s_add_u32(t[0], self.s_k0, 32)
            p.s_cmp("ge_i32", t[0], P("ns"))                        # no sink key in keys k0 + 32 .. k0 + 63
            p.s_cselect(t[1], 1, 0)
            p.s_sub_i32(t[2], self.s_pw0, P("W"))
            p.s_add_u32(t[0], self.s_k0, 63)
            p.s_cmp("le_i32", t[0], t[2])                           # k0 + 63 <= pw0 - W
            p.s_cselect(t[2], 1, 0)
            p.s_and_b32(t[1], t[1], t[2])
            p.s_cmp("lt_i32", self.s_k0, P("ns"))                   # it is a sink tile
            p.s_cselect(t[2], 1, 0)
            p.s_and_b32(t[1], t[1], t[2])
            p.s_cmp("lg_u32", t[1], 0)
            p.s_cselect(self.s_cls, 4, self.s_cls)

    def prologue_pk(self) -> Prog:
        p = self.setup_pk()
        if self.stamps:
            for r in self.s_acc:
                p.s_mov(r, 0)
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
        p.s_waitcnt(vmcnt=2 * 4 * self.HALVES, note="fragments, row constants, tile 0 landed (two tiles in flight)")
        p.s_barrier()
        self.emit_item_init(p)
        return p

    def emit_store_setup(self, p: Prog):
        """the finished item's dQ descriptor (d_x) and row offsets (vo), before its row positions are replaced"""
        land = self.item_land()
        t0, t1, t2, t3 = self.tmp
        st = self.s_tmp
        self.desc_read(p, self.s_item, (2, 3), land, t0)
        self.desc_get(p, self.d_x[0], land, "dq_lo")
        self.desc_get(p, self.d_x[1], land, "dq_hi")
        p.s_mul_i32(st[1], self.s_hh, P("dq_hs"))
        p.s_mul_hi_u32(st[2], self.s_hh, P("dq_hs"))
        p.s_add_u32(self.d_x[0], self.d_x[0], st[1])
        p.s_addc_u32(self.d_x[1], self.d_x[1], st[2])
        self.desc_get(p, self.d_x[2], land, "dq_rng")
        p.s_mov(self.d_x[3], 0x00020000)
        p.v_sub_u32(t0, self.v_pos[0], P("pos0"))             # row, rb = 0
        p.v_mul_lo_u32(t1, t0, P("dq_sn"))
        p.v_lshrrev(t2, 5, self.lane)
        p.v_lshl_add_u32(self.vo[0], t2, 4, t1)               # + 16 h bytes
        p.s_lshl_b32(st[1], P("dq_sn"), 5)
        p.v_add_u32(self.vo[1], st[1], self.vo[0])
        if self.wide64:
            # 64-byte row pieces: lane = 16 g + l writes bytes 16 g .. of row (first row of the block) + l; the second store of
            # a pair the rows 16 further
            p.v_and(t2, 15, self.lane)
            p.v_sub_u32(t0, self.v_pos[0], P("pos0"))
            p.v_and(t3, 31, self.lane)
            p.v_sub_u32(t0, t0, t3)                               # first row of row block 0
            p.v_add_u32(t0, t0, t2)
            p.v_mul_lo_u32(t1, t0, P("dq_sn"))
            p.v_lshrrev(t2, 4, self.lane)
            p.v_lshl_add_u32(self.vo64[0], t2, 4, t1)
            p.v_add_u32(self.vo64[1], st[1], self.vo64[0])
            p.s_lshl_b32(st[1], P("dq_sn"), 4)
            p.v_add_u32(self.vo64b[0], st[1], self.vo64[0])
            p.v_add_u32(self.vo64b[1], st[1], self.vo64[1])

    def emit_stores(self, p: Prog):
        """dQ[row, d] = scale * dQ^T[d, row] (as the one-item epilogue).  wide64: the two 16-column groups of a 32-column block
        are exchanged once more between the 16-lane rows (v_permlane32_swap + v_permlane16_swap), so that four lanes hold 64
        contiguous bytes of one row and a store instruction touches 16 rows instead of 32 (the stores of an item transition
        are bound by the cache lines an instruction touches)"""
        dt = self.dtype
        npair = 0

        def quad(rb, db, gp, X, Y):
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

        for rb in range(2):
            for db in range(self.DB):
                if self.wide64 and 32 * db + 32 <= self.D:
                    X0, Y0, X1, Y1 = (self.POOL[(4 * npair + i) % 8] for i in range(4))
                    npair += 1
                    quad(rb, db, 0, X0, Y0)
                    quad(rb, db, 1, X1, Y1)
                    for e in range(4):
                        p.v_permlane32_swap(X0[e], X1[e])
                        p.v_permlane16_swap(X0[e], X1[e])
                    p.buffer_store(X0[0:4], self.vo64[rb], self.d_x, 0, offset=64 * db)
                    p.buffer_store(X1[0:4], self.vo64b[rb], self.d_x, 0, offset=64 * db)
                    continue
                for gp in range(2):
                    if 32 * db + 16 * gp >= self.D:
                        continue
                    X, Y = self.POOL[(2 * npair) % 8], self.POOL[(2 * npair + 1) % 8]
                    npair += 1
                    quad(rb, db, gp, X, Y)
                    p.buffer_store(X[0:4], self.vo[rb], self.d_x, 0, offset=64 * db + 32 * gp)

    def build_pk(self):
        items = []
        items += finish_block(self.prologue_pk().items)
        items += insert_waits(self.loop_top().items)
        for cls, lbl in ((0, "L_full%="), (1, "L_edge%="), (2, "L_sink%=")) + (((3, "L_dead%="),) if self.dead else ()) + (((4, "L_sinkfar%="),) if self.sinkfar else ()) + (((5, "L_edge_d%="), (6, "L_edge_w%=")) if self.edge_subs else ()):
            body = self.tile_body(cls).items
            items.append(Instr("label", mods={"label": lbl}, kind="label", cost=0))
            if self.do_sched:
                body = schedule(body)
            body = insert_waits(body)
            body = fix_hazards(body, loop=True)
            items += body
            items.append(Instr("s_branch", mods={"label": "L_top%="}, kind="branch"))
        if self.edge_subs:
            items += finish_block(self.edge_selector().items)
        # ---- item transition: the next item's requests go out BEFORE the finished item's stores are formed
        p = Prog()
        p.label("L_done%=")
        p.s_waitcnt(lgkmcnt=0, note="the K fragments fetched at the end of the last trip (re-read below)")
        self.emit_stamp(p, -1)
        self.emit_store_setup(p)
        p.s_add_u32(self.s_item, self.s_item, 1)
        p.s_cmp("ge_u32", self.s_item, P("n_items"))
        p.s_cbranch("scc1", "L_last%=")
        items += finish_block(p.items)
        p = Prog()
        self.emit_item_begin(p)
        self.emit_stamp(p, 0)                    # requests of the next item issued
        self.emit_stores(p)
        self.emit_stamp(p, 1)                    # stores of the finished item issued
        self.emit_item_init(p)
        self.emit_stamp(p, 2)                    # the next item's fragments / constants have landed, first K fragments requested
        blk = fix_hazards(insert_waits(p.items, strict_tail=True))
        items += blk
        # (one barrier between the finished item's last LDS reads and the first LDS-DMA of the new item's first body: the
        # loop head's, entered behind its vmcnt wait - the stores just issued need not have drained)
        items.append(Instr("s_branch", mods={"label": "L_top_b%="}, kind="branch"))
        p = Prog()
        p.label("L_last%=")
        self.emit_stores(p)
        p.s_waitcnt(vmcnt=0)
        if self.stamps:      # lane 0 of every wave: the three sums + the item count, 16 bytes at dbg[(4 bid + wave) * 16]
            st = self.s_tmp
            t0, t1 = self.tmp[0], self.tmp[1]
            p.s_mov(self.d_y[0], P("dbg_lo"))
            p.s_mov(self.d_y[1], P("dbg_hi"))
            p.s_mov(self.d_y[3], 0x00020000)
            p.s_lshl_b32(st[0], P("bid"), 2)
            p.s_add_u32(st[0], st[0], self.s_wave)
            p.s_lshl_b32(st[0], st[0], 4)
            p.s_add_u32(self.d_y[2], st[0], 16)
            p.v_mov(t0, st[0])
            p.v_cmp("eq_u32", 0, self.lane)
            p.v_cndmask(t0, self.v_oob, t0)
            for k2, src in enumerate(self.s_acc + [P("n_items")]):
                p.v_mov(t1, src)
                p.buffer_store(t1, t0, self.d_y, 0, offset=4 * k2)
            p.s_waitcnt(vmcnt=0)
        items += finish_block(p.items)
        return items

    def build(self):
        if self.persist:
            return self.build_pk()
        items = []
        items += finish_block(self.prologue().items)
        items += insert_waits(self.loop_top().items)
        for cls, lbl in ((0, "L_full%="), (1, "L_edge%="), (2, "L_sink%=")) + (((3, "L_dead%="),) if self.dead else ()) + (((4, "L_sinkfar%="),) if self.sinkfar else ()) + (((5, "L_edge_d%="), (6, "L_edge_w%=")) if self.edge_subs else ()):
            body = self.tile_body(cls).items
            items.append(Instr("label", mods={"label": lbl}, kind="label", cost=0))
            if self.do_sched:
                body = schedule(body)
            body = insert_waits(body)
            body = fix_hazards(body, loop=True)
            items += body
            items.append(Instr("s_branch", mods={"label": "L_top%="}, kind="branch"))
        if self.edge_subs:
            items += finish_block(self.edge_selector().items)
        items += finish_block(self.epilogue().items)
        return items

    def clobbers(self):
        c = ["v%d" % i for i in range(self.vfirst, 256)] + ["a%d" % i for i in range(256)]
        c += ["s%d" % i for i in range(self.sfirst, 100)] + ["vcc", "scc", "memory"]
        return c
