"""Generator of the hand-placed SHORT-WINDOW forward kernel body (gfx950, head dims 64 / 80 / 96, bf16 / f16).

The gpt-oss sliding layers (sink_attention/verl_patch.py:156-174 of the reference: sliding_window = 128, num_sink = 0,
s_aux) and BASELINE.json's C4: a query tile sees ~3 key tiles, so the shape is bound by HBM and by per-tile fixed costs,
not by MFMA.  Same maths and tile phases as fwd.py (A = S^T chains, M = mask / row maximum / reference point, E = exp2
+ pack, C = PV + row sums); what differs is the walk:

  workgroup = 4 waves x 64 query rows of 4 q heads of one GQA group (hpw = 4), and it owns a STRIP of consecutive
  64-row query tiles ("items") of that (batch, KV head, head set), walked in increasing order.
  * Consecutive items share all but one of their NT = ceil((W - 1) / 64) + 1 key tiles: the K / V tiles live in an LDS
    ring of NT + 1 slots that slides along the strip - ONE new tile is fetched per item (LDS-DMA, a whole item ahead), one
    barrier per item.
  * Head dims below 128 leave registers free: the Q fragments are DOUBLE BUFFERED (the next item's are requested while
    the current item computes), and the finished item's output is normalised, packed and stored in the gaps of the next
    item's first MFMAs (its accumulators are free again only when the next item's first PV MFMA - srcC = 0 - writes them:
    the scheduler sees that as an ordinary register dependence).
  * An item's tiles are unrolled (NT is a launch constant): per item one BIG block
        [E, C of the previous item's last tile | A(t0) M(t0) | previous item's epilogue | Q requests | tile DMA |
         A(t1) E(t0) C(t0) M(t1)]
    and NT - 2 MID blocks [A(t) E(t-1) C(t-1) M(t)], each placed around its MFMA spine by sched.py; the rare rescale of O
    and l (deferred reference point, as fwd.py) runs between blocks.  Two copies (Q buffer parity).
  * Every tile goes through the one-compare mask ((pos - key) <u W_tile) with a per-tile threshold: W, or 0 for tile
    indices below 0 (the first items of a sequence; their fetches fall outside the buffer range and read zeros), so
    there is no special case for the start of a sequence.
Restrictions (the HIP shell's rule): num_sink = 0, N_q = N_kv, GQA group a multiple of 4, NT <= 4 (W <= 192).
"""
from __future__ import annotations

from .core import A, Imm, Instr, M0, P, PV, Prog, Reg, S, V, VCC, fimm, imm
from .fwd import NEG_INF, STG_BYTES, FwdGen
from .sched import finish_block, fix_hazards, insert_waits, schedule

PARAMS = [
    "q_lo", "q_hi", "q_hs", "q_sn", "q_rng",
    "o_lo", "o_hi", "o_hs", "o_sn", "o_rng",
    "k_lo", "k_hi", "k_sn", "k_rng", "v_lo", "v_hi", "v_sn", "v_rng",
    "lse_lo", "lse_hi", "ld_hs",
    "m0_0", "m0_1", "m0_2", "m0_3", "l0",
    "q0", "n_it", "nrows", "W", "c_log2", "ln2",
]
# q / o / lse bases: (first head of the workgroup's head set, the sequence's first row); q0 = first row of the strip's
# first item; n_it = items of the strip (>= 1); k / v bases: the KV head, the sequence's first row.


class FwdStripGen(FwdGen):
    def __init__(self, dtype="bf16", D=80, NT=3, sched=True, vfirst=4, sfirst=48, npool=10, thr=8.0):
        assert D in (64, 80, 96) and 2 <= NT <= 4
        FwdGen.__init__(self, dtype, sched=sched, vfirst=vfirst, sfirst=sfirst, npool=npool, thr=thr, D=D, persist=False,
                        kpre=False)
        self.NT, self.R = NT, NT + 1
        self.RING = self.R * STG_BYTES
        self.inf_no_rescale = True
        va, sa = self.va, self.sa
        # ---------------- AGPRs: two Q buffers, row sums, ones, O^T
        DK, DB = self.DK, self.DB
        self.QFB = [[[A(((b * 2 + rb) * DK + ks) * 4, 4) for ks in range(DK)] for rb in range(2)] for b in range(2)]
        base = 2 * 2 * DK * 4
        self.LACC = [A(base + rb * 16, 16) for rb in range(2)]
        self.ONES = A(base + 32, 4)
        self.OACC = [[A(base + 36 + (rb * DB + db) * 16, 16) for db in range(DB)] for rb in range(2)]
        assert base + 36 + 2 * DB * 16 <= 256
        # ---------------- more VGPRs / SGPRs than the one-item body
        self.ak = [[va("ak%d%s" % (i, x)) for x in "eo"] for i in range(2)]       # K row-read addresses of two tiles in one block
        self.v_wt = va("v_wt")                                                     # mask threshold of the tile being masked
        self.v_wc = va("v_wc")                                                     # W (steady blocks)
        self.steady = False
        self.st_from, self.st_step, self.ld_step = None, 0, 1
        self.vl = [va("vl%d" % rb) for rb in range(2)]                             # Q load offsets (next item)
        self.v_ls = [va("v_ls%d" % rb) for rb in range(2)]                         # LSE store offsets (finished item)
        # registers of the finished item's epilogue (its own: it is spread over the gaps of the next item's first phases)
        self.ep = [va("ep%d" % i, 4, 4) for i in range(4)]
        self.ep_inv = [va("ep_inv%d" % rb) for rb in range(2)]
        self.ep_lg = [va("ep_lg%d" % rb) for rb in range(2)]
        self.ep_t = [va("ep_t%d" % i) for i in range(2)]
        # (scalars of the one-item body that this walk does not use are taken over: s_it, s_k0n, s_st, s_stn, s_std, s_cls,
        # s_rgi, s_pw0, s_pwhi)
        spare = [self.s_it, self.s_k0n, self.s_st, self.s_stn, self.s_std, self.s_cls, self.s_rgi, self.s_pw0, self.s_pwhi]
        take = lambda name: spare.pop() if spare else sa(name)
        self.s_n, self.s_T = take("s_n"), take("s_T")                              # items left, first tile of the current item
        self.s_s0 = take("s_s0")                                                   # ring offset of the current item's first tile
        self.s_slot = [take("s_slot%d" % j) for j in range(NT)]                    # ring offsets of the current item's tiles
        self.s_k0 = [take("s_k0_%d" % j) for j in range(NT)]                       # their first keys (signed)
        self.s_m0 = take("s_m0")
        self.s_par = sa("s_par")
        self.d_q = sa("d_q", 4, 4)
        self.d_x2 = sa("d_x2", 4, 4)
        self.uid = 0

    def params(self):
        return list(PARAMS)

    @staticmethod
    def gate(ins, k):
        """not before the k-th MFMA of the block, and right there (sched.py: after_mfma)"""
        ins.mods["after_mfma"] = int(k)
        ins.mods["alap"] = 32 * (int(k) + 1)

    def label(self, stem):
        self.uid += 1
        return "L_%s%d_%%=" % (stem, self.uid)

    # ------------------------------------------------------------------ small pieces
    def emit_wrap(self, p: Prog, dst, src, add):
        """dst = (src + add) mod RING (src < RING, add <= RING)"""
        p.s_add_u32(dst, src, add)
        p.s_cmp("ge_u32", dst, self.RING)
        p.s_cselect(self.s_tmp[4], self.RING, 0)
        p.s_sub_u32(dst, dst, self.s_tmp[4])

    def emit_item_scalars(self, p: Prog):
        """ring offsets and first keys of the current item's tiles from s_s0 / s_T"""
        for j in range(self.NT):
            if j == 0:
                p.s_mov(self.s_slot[0], self.s_s0)
            else:
                self.emit_wrap(p, self.s_slot[j], self.s_slot[j - 1], STG_BYTES)
            p.s_add_i32(self.s_tmp[3], self.s_T, j)
            p.s_lshl_b32(self.s_k0[j], self.s_tmp[3], 6)

    def emit_dma(self, p: Prog, tile, slot, spread_from=None):
        """LDS-DMA of K / V tile `tile` (SGPR, signed: negative or beyond the sequence = outside the buffer range, zeros)
        into ring offset `slot` (SGPR): this wave's pieces"""
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

    def emit_q_loads(self, p: Prog, buf, spread_from=None):
        for rb in range(2):
            for ks in range(self.DK):
                ins = p.buffer_load(self.QFB[buf][rb][ks], self.vl[rb], self.d_q, 0, offset=32 * ks)
                if spread_from is not None:      # (spread: the CU's address path takes ~56 cycles per such load)
                    self.gate(ins, spread_from + self.ld_step * (rb * self.DK + ks))

    def tile_sub(self, j):
        """sub-block classes (fwd.py: sub[kh][rb]) of the item's tile j.  The LAST tile is the diagonal one for every item (its
        keys are the item's own 64 positions): rows 0..31 never see keys 32..63 - dead in both block sets (its E / C run in
        the NEXT item's first block, whatever set that item takes).  Steady items (W = 64 (NT - 1) or one more, every tile
        exists): rows 32..63 never see the first tile's keys 0..31, and see all of the diagonal tile's keys 0..31; rows 0..31
        see all of the first tile's keys 32..63."""
        sub = [["mask", "mask"], ["mask", "mask"]]
        if j == self.NT - 1:
            sub[1][0] = "dead"
            if self.steady:
                sub[0][1] = "full"
        elif j == 0 and self.steady:
            sub[0][1] = "dead"
            sub[1][0] = "full"
        else:
            return None
        return sub

    def emit_A_at(self, p: Prog, par, j, i):
        """S^T of the item's tile j into SS[par]; address register pair i"""
        e, o = self.ak[i]
        p.v_add_u32(e, self.s_slot[j], self.l_row_e)
        p.v_xor(o, 32, e)
        self.sub = self.tile_sub(j)
        self.emit_A(p, par, e, o)
        self.sub = None

    def emit_E_at(self, p: Prog, par, j):
        self.sub = self.tile_sub(j)
        self.emit_E(p, par)
        self.sub = None

    def emit_M_at(self, p: Prog, par, j):
        """mask / maximum / reference point of tile j in SS[par].  General blocks: the one-compare mask with threshold W, or 0
        for a tile index below 0.  Steady blocks (every tile of the item exists, W = 64 (NT - 1) or one more - the gpt-oss
        window of 128): the last tile is the diagonal one (causal test only), the first lies wholly below the rows (window test
        only; two VALU per element instead of three), the ones between need no mask."""
        self.sub = self.tile_sub(j)
        if self.steady:
            keep = self.v_w
            self.v_w = self.v_wc
            # (steady blocks run only when W = 64 (NT - 1) or one more: the tiles between the first and the diagonal one are
            # then wholly inside every row's window and need no mask at all)
            self.emit_M(p, par, 3 if j == self.NT - 1 else (4 if j == 0 else 0), self.s_k0[j])
            self.v_w = keep
            self.sub = None
            return
        p.s_cmp("lt_i32", self.s_k0[j], 0)
        p.s_cselect(self.s_tmp[4], 0, P("W"))
        p.v_mov(self.v_wt, self.s_tmp[4])
        keep = self.v_w
        self.v_w = self.v_wt
        self.emit_M(p, par, 1, self.s_k0[j])
        self.v_w = keep
        self.sub = None

    def emit_C_at(self, p: Prog, par, j, first):
        """O^T += V^T P^T, l += 1 P^T for tile j in SS[par]; first: the item's first tile - O starts from 0 (srcC = 0, the
        accumulators still hold the previous item's output until then) and l from l0 alpha"""
        dt = self.dtype
        p.v_add_u32(self.a_tr0, self.s_slot[j], self.l_tr0)
        p.v_xor(self.a_tr1, 32, self.a_tr0)
        if first:
            for rb in range(2):
                p.v_mul_f32(self.tmp[rb], P("l0"), self.alpha[rb])
                for i in range(16):
                    p.v_accvgpr_write(self.LACC[rb][i], self.tmp[rb])
        sub = self.tile_sub(j)
        dead = lambda kh, rb: sub is not None and sub[kh][rb] == "dead"
        started = [False, False]          # the row block's first PV MFMA of the item starts O from 0
        for kh in range(2):
            for s in range(2):
                pf = [self.SS[par][kh][rb][4 * s:4 * s + 4] for rb in range(2)]
                for rb in range(2):
                    if not dead(kh, rb):
                        p.mfma(dt, self.LACC[rb], self.ONES, pf[rb], self.LACC[rb], tag="l")
                for db in range(self.DB):
                    f = self.pool()
                    off = 16384 + 8192 * kh + 512 * db
                    p.ds_read_b64_tr_b16(f[0:2], self.a_tr0, off + 2048 * (2 * s), mem=("stage_r",))
                    p.ds_read_b64_tr_b16(f[2:4], self.a_tr1, off + 2048 * (2 * s + 1), mem=("stage_r",))
                    for rb in range(2):
                        if dead(kh, rb):
                            continue
                        zero = first and not started[rb]
                        p.mfma(dt, self.OACC[rb][db], f, pf[rb], 0 if zero else self.OACC[rb][db], tag="PV")
                for rb in range(2):
                    if not dead(kh, rb):
                        started[rb] = True

    def emit_epilogue(self, p: Prog):
        """the finished item: O = O^T / l (l = 0 -> 1), LSE = ln2 (m + log2 l), stores; m of the finished item is read HERE,
        so this sits in front of the next item's M(t0) in program order"""
        dt = self.dtype
        t3, t2 = self.ep_t
        inv, lg = self.ep_inv, self.ep_lg
        for rb in range(2):
            p.v_accvgpr_read(t3, self.LACC[rb][0])
            p.v_mov(t2, fimm(1.0))
            p.v_cmp_m("eq_f32", self.s_f0, 0, t3)
            p.v_cndmask_m(t3, t3, t2, self.s_f0)
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
                    X, Y = self.ep[(2 * npair) % 4], self.ep[(2 * npair + 1) % 4]
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
                    ins = p.buffer_store(X[0:4], self.vo[rb], self.d_x, 0, offset=64 * db + 32 * gp)
                    if self.st_from is not None:
                        self.gate(ins, self.st_from + self.st_step * (npair - 1))
        p.buffer_store(lg[0], self.v_ls[0], self.d_o2, 0)
        p.buffer_store(lg[1], self.v_ls[1], self.d_o2, 0)
        # the next finished item sits 64 rows further
        p.s_lshl_b32(self.s_tmp[4], P("o_sn"), 6)
        for rb in range(2):
            p.v_add_u32(self.vo[rb], self.s_tmp[4], self.vo[rb])
            p.v_add_u32(self.v_ls[rb], 256, self.v_ls[rb])

    def n_store(self):
        return sum(1 for rb in range(2) for db in range(self.DB) for gp in range(2) if 32 * db + 16 * gp < self.D) + 2

    def emit_state_init(self, p: Prog):
        """m of the new item (the head's s_aux logit or -inf)"""
        for rb in range(2):
            p.v_mov(self.m[rb], self.s_m0)

    def emit_advance(self, p: Prog):
        """bookkeeping for the NEXT item: tile window, ring offset, row positions, Q load offsets"""
        p.s_add_i32(self.s_T, self.s_T, 1)
        self.emit_wrap(p, self.s_s0, self.s_s0, STG_BYTES)
        p.s_sub_u32(self.s_n, self.s_n, 1)

    def emit_rescale_check(self, p: Prog):
        l_no = self.label("nors")
        t = self.tmp
        p.s_cmp_lg_u64(self.s_flag, 0)
        p.s_cbranch("scc0", l_no)
        k = 0
        for rb in range(2):
            regs = [self.LACC[rb][i] for i in range(16)] + [self.OACC[rb][db][i] for db in range(self.DB) for i in range(16)]
            for a in regs:
                r = t[k % 4]
                k += 1
                p.v_accvgpr_read(r, a)
                p.v_mul_f32(r, self.alpha[rb], r)
                p.v_accvgpr_write(a, r)
        p.label(l_no)

    # ------------------------------------------------------------------ blocks
    def prologue(self) -> Prog:
        p = Prog()
        t0, t1, t2, t3 = self.tmp[:4]
        st = self.s_tmp
        lane, wv = self.lane, self.s_wave
        self.d_o2 = self.d_x2
        p.v_and(lane, 63, PV("tid"))
        p.v_lshrrev(t0, 6, PV("tid"))
        p.v_readfirstlane(wv, t0)
        p.v_and(self.lane31, 31, lane)
        p.v_mov(self.v_oob, imm(0x7FFFF000))
        p.s_mov(self.s_hh, wv)                                # hpw = 4: wave = head, one row group
        # lane parts of the LDS addresses (fwd.py)
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
        # K / V stream descriptors and lane offsets
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
        # strip state: first item's tiles T .. T + NT - 1 (T = q0 / 64 - (NT - 1), may be negative), ring offset 0
        p.s_lshr_b32(st[0], P("q0"), 6)
        p.s_sub_i32(self.s_T, st[0], self.NT - 1)
        p.s_mov(self.s_s0, 0)
        p.s_mov(self.s_n, P("n_it"))
        self.emit_item_scalars(p)
        for j in range(self.NT):
            p.s_add_i32(st[3], self.s_T, j)
            self.emit_dma(p, st[3], self.s_slot[j])
        # rows of the lane: q0 + 32 rb + r ; descriptors of Q, O, LSE (head = wave)
        p.v_add_u32(t0, P("q0"), self.lane31)
        p.v_mov(self.v_pos[0], t0)
        p.v_add_u32(self.v_pos[1], 32, self.v_pos[0])
        for nm, d in (("q", self.d_q), ("o", self.d_x)):
            p.s_mul_i32(st[1], self.s_hh, P(nm + "_hs"))
            p.s_mul_hi_u32(st[2], self.s_hh, P(nm + "_hs"))
            p.s_add_u32(d[0], P(nm + "_lo"), st[1])
            p.s_addc_u32(d[1], P(nm + "_hi"), st[2])
            p.s_mov(d[2], P(nm + "_rng"))
            p.s_mov(d[3], 0x00020000)
        p.s_mul_i32(st[1], self.s_hh, P("ld_hs"))
        p.s_mul_hi_u32(st[2], self.s_hh, P("ld_hs"))
        p.s_add_u32(self.d_o2[0], P("lse_lo"), st[1])
        p.s_addc_u32(self.d_o2[1], P("lse_hi"), st[2])
        p.s_lshl_b32(self.d_o2[2], P("nrows"), 2)
        p.s_mov(self.d_o2[3], 0x00020000)
        p.v_mul_lo_u32(t1, t0, P("q_sn"))
        p.v_lshl_add_u32(self.vl[0], t2, 4, t1)
        p.s_lshl_b32(st[1], P("q_sn"), 5)
        p.v_add_u32(self.vl[1], st[1], self.vl[0])
        p.v_mul_lo_u32(t1, t0, P("o_sn"))
        p.v_lshl_add_u32(self.vo[0], t2, 4, t1)
        p.s_lshl_b32(st[1], P("o_sn"), 5)
        p.v_add_u32(self.vo[1], st[1], self.vo[0])
        p.v_lshlrev(t1, 2, t0)
        p.v_mov(t3, imm(0x7FFFFFF0))
        p.v_cmp("gt_u32", 32, lane)
        p.v_cndmask(self.v_ls[0], t3, t1)                     # lanes >= 32: out of range
        p.v_add_u32(self.v_ls[1], 128, self.v_ls[0])
        self.emit_q_loads(p, 0)
        # the head's initial reference point, constants
        p.s_cmp("eq_u32", self.s_hh, 1)
        p.s_cselect(self.s_m0, P("m0_1"), P("m0_0"))
        p.s_cmp("eq_u32", self.s_hh, 2)
        p.s_cselect(self.s_m0, P("m0_2"), self.s_m0)
        p.s_cmp("eq_u32", self.s_hh, 3)
        p.s_cselect(self.s_m0, P("m0_3"), self.s_m0)
        ones = 0x3F803F80 if self.dtype == "bf16" else 0x3C003C00
        p.v_mov(t1, imm(ones))
        for i in range(4):
            p.v_accvgpr_write(self.ONES[i], t1)
        p.v_mov(self.v_ninf, imm(NEG_INF))
        p.v_mov(self.v_wc, P("W"))
        p.s_waitcnt(vmcnt=0, note="first item: its tiles and Q fragments")
        p.s_barrier()
        return p

    def advance_rows(self, p: Prog):
        """row positions and Q load offsets move on by one item (64 rows)"""
        for rb in range(2):
            p.v_add_u32(self.v_pos[rb], 64, self.v_pos[rb])
        p.s_lshl_b32(self.s_tmp[4], P("q_sn"), 6)
        for rb in range(2):
            p.v_add_u32(self.vl[rb], self.s_tmp[4], self.vl[rb])

    def big(self, buf: int, head: bool) -> Prog:
        """buf: Q buffer of the item that STARTS here; head: the strip's first item (nothing to finish)"""
        p = Prog()
        NT = self.NT
        self.pool_next = 0
        self.QF = self.QFB[buf]
        last = (NT - 1) & 1
        if not head:
            self.emit_E_at(p, last, NT - 1)                   # previous item's last tile (its slot offsets are still in s_slot)
            self.emit_C_at(p, last, NT - 1, first=False)
            # the new item: tile window + 1, ring offset + 1 slot, rows + 64
            self.emit_advance(p)
            self.emit_item_scalars(p)
            for rb in range(2):
                p.v_add_u32(self.v_pos[rb], 64, self.v_pos[rb])
        self.emit_A_at(p, 0, 0, 0)
        # requests for the NEXT item: its Q fragments into the other buffer, its newest tile into the free ring slot.  In
        # program order they stand in FRONT of the finished item's stores, which depend on them ("vmq"): the stores are the
        # youngest entries of the memory queue, so the wait at the next item's head can leave them in flight.
        p.s_lshl_b32(self.s_tmp[4], P("q_sn"), 6)
        for rb in range(2):
            p.v_add_u32(self.vl[rb], self.s_tmp[4], self.vl[rb])
        # positions (MFMA indices of the BIG block: 8 DB + 8 of the previous item's last PV, 8 DK of A(t0), 8 DK of A(t1), then
        # C(t0)): the finished item's stores and the next item's loads interleave between the end of the previous PV and C(t0)
        # - the stores must be out before C(t0) writes the accumulators - the tile's DMA pieces go under C(t0)
        n_sub = lambda j: 4 - sum(x == "dead" for row in (self.tile_sub(j) or []) for x in row)      # live sub-blocks of tile j
        n0 = 0 if head else (2 * self.DB + 2) * n_sub(NT - 1)
        n1 = n0 + self.DK * (n_sub(0) + (n_sub(1) if NT >= 2 else 0))
        nld, nst = 2 * self.DK, self.n_store() - 2
        self.ld_step = max(1, (n1 - n0 - 2) // nld)
        self.st_from, self.st_step = (None, 0) if head else (n0 + 2, max(1, (n1 - n0 - 4) // nst))
        self.emit_q_loads(p, buf ^ 1, spread_from=n0 + 1)
        p.s_add_i32(self.s_tmp[3], self.s_T, NT)
        self.emit_wrap(p, self.s_tmp[2], self.s_slot[NT - 1], STG_BYTES)
        self.emit_dma(p, self.s_tmp[3], self.s_tmp[2], spread_from=n1 + 4)
        if not head:
            self.emit_epilogue(p)                             # reads the finished item's m, l, O^T
        self.emit_state_init(p)
        self.emit_M_at(p, 0, 0)
        if NT >= 2:
            self.emit_A_at(p, 1, 1, 1)
        self.emit_E_at(p, 0, 0)
        self.emit_C_at(p, 0, 0, first=True)
        if NT >= 2:
            self.emit_M_at(p, 1, 1)
        return p

    def mid(self, buf: int, j: int) -> Prog:
        """[A(t_j) E(t_{j-1}) C(t_{j-1}) M(t_j)], 2 <= j < NT"""
        p = Prog()
        self.pool_next = 0
        self.QF = self.QFB[buf]
        self.emit_A_at(p, j & 1, j, 0)
        self.emit_E_at(p, (j - 1) & 1, j - 1)
        self.emit_C_at(p, (j - 1) & 1, j - 1, first=False)
        self.emit_M_at(p, j & 1, j)
        return p

    def tail(self) -> Prog:
        p = Prog()
        NT = self.NT
        self.pool_next = 0
        last = (NT - 1) & 1
        self.emit_E_at(p, last, NT - 1)
        self.emit_C_at(p, last, NT - 1, first=False)
        self.emit_epilogue(p)
        p.s_waitcnt(vmcnt=0)
        return p

    def build(self):
        items = finish_block(self.prologue().items)
        nst = self.n_store()
        nq = 2 * self.DK
        ndma = 2 * 2 * self.HALVES

        def block(prog, loop=False):
            b = prog.items
            if self.do_sched:
                b = schedule(b)
            return fix_hazards(insert_waits(b))

        def item_blocks(buf, head, steady=False):
            self.steady = steady
            out = []
            out += block(self.big(buf, head))
            p = Prog()
            self.emit_rescale_check(p)
            out += finish_block(p.items)
            for j in range(2, self.NT):
                out += block(self.mid(buf, j))
                p = Prog()
                self.emit_rescale_check(p)
                out += finish_block(p.items)
            self.steady = False
            return out

        # the strip's first item (Q buffer 0, general masks), then the loop: items alternate between the Q buffers; an item
        # whose tiles all exist (first tile index >= 0) takes the steady blocks when the window spans a whole tile
        items += item_blocks(0, True)
        p = Prog()
        p.s_mov(self.s_par, 1)
        p.s_cmp("le_u32", self.s_n, 1)
        p.s_cbranch("scc1", "L_tail%=")
        p.s_waitcnt(vmcnt=0, note="second item's Q fragments and newest tile (the first item has left no stores behind)")
        p.s_branch("L_disp%=")
        p.label("L_next%=")
        p.s_cmp("le_u32", self.s_n, 1)
        p.s_cbranch("scc1", "L_tail%=")
        p.s_waitcnt(vmcnt=0, note="next item's Q fragments and newest tile (own pieces)")
        p.label("L_disp%=")
        p.s_barrier()
        # the NEXT item's first tile is s_T + 1 (the advance happens inside its BIG block)
        p.s_cmp("ge_i32", self.s_T, -1)
        p.s_cselect(self.s_tmp[0], 2, 0)
        p.s_sub_u32(self.s_tmp[1], P("W"), 64 * (self.NT - 1))          # steady blocks: W = 64 (NT - 1) or one more (see emit_M_at)
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
            items += item_blocks(code & 1, False, steady=bool(code & 2))
            items.append(Instr("s_branch", mods={"label": "L_next%="}, kind="branch"))
        p = Prog()
        p.label("L_tail%=")
        items += finish_block(p.items)
        items += block(self.tail())
        return items

    def clobbers(self):
        c = ["v%d" % i for i in range(self.vfirst, 256)] + ["a%d" % i for i in range(256)]
        c += ["s%d" % i for i in range(self.sfirst, 100)] + ["vcc", "scc", "memory"]
        return c
