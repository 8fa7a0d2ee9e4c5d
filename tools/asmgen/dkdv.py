"""Generator of the hand-placed dK/dV backward kernel body (gfx950, head dim 128, bf16 / f16).

Replaces _sink_flash_attn_bwd_dkdv_kernel of the reference (sink_attention/sink_flash_attention.py:256-364) plus
its PyTorch GQA group sum (:648-651); same maths as csrc/sfa_bwd_mfma.hip's wave-specialised kernel, different
machine mapping:

  workgroup = 4 waves = 256 keys of one (batch, KV head); wave w owns keys [64w, 64w+64) as two 32-key blocks and keeps
  dK^T / dV^T of them in all 256 accumulator registers (one wave per SIMD, 512-register waves).  The workgroup sweeps
  the q heads of the GQA group x the 32-row query slices that can see its keys ("trips").  Per trip and wave:
      S'[q,key]  = Q K^T - LSE/scale        16 MFMA  (A: Q row fragments from LDS, B: K fragments pinned in VGPRs,
                                                      C: the row constant as the INITIAL accumulator, read from LDS)
      dP'[q,key] = dO V^T - Delta           16 MFMA  (A: dO row fragments, B: V row fragments, both from LDS)
      P = exp2(c S'), dS = P dP'            VALU, 32 elements per lane; packed to 16 bit in the accumulator layout,
                                            which IS the B operand layout of the two products below (no LDS trip)
      dV^T[d,key] += dO^T P                 16 MFMA  (A: transposed LDS reads of the dO slice image)
      dK^T[d,key] += Q^T dS                 16 MFMA  (A: transposed LDS reads of the Q slice image)
  64 MFMAs per trip and SIMD = 2048 cycles of matrix pipe; everything else (80 LDS reads, 5 LDS-DMA pieces, ~150 VALU)
  is placed into the gaps by tools/asmgen/sched.py.
  Q / dO slices (and the 64 row constants) arrive by LDS-DMA three slices ahead into a 4-deep ring; one s_barrier per
  trip; the wave's V rows sit in LDS for the whole workgroup, its K fragments in registers.
  Trip bodies per wave: "full" (no mask instructions; the only class the scheduled bodies compute for the next trip - everything
  else is told apart by a scalar selector behind the loop head that only non-full trips reach), "edge" (per-element masks), edge
  on ONE of the wave's two 32-key blocks (the other is out of every row's reach) and "dead" - no row of the slice sees
  any key of the wave (the block's sweep covers the rows that ANY of its 256 keys can see: a wave's 64 keys are out of reach
  in 6 of them per q head): the wave issues its LDS-DMA pieces, keeps its state and waits at the barrier.  Same results bit
  for bit; C3 dK/dV -2.9 %, W = 1024 -4 ... -8 %, W = 128 -9 % (in-process A/B, profiles/r03_ab_dkdv_dead.log): the idle
  waves' MFMAs cost power (the XCDs clock higher without them) and LDS bandwidth.

LDS map (bytes):   [0, 1024)            row constants, 4 stages x (32 x -LSE/scale | 32 x -Delta) f32
                   [1024, 66560)        4 stages x (Q slice image 8 KB | dO slice image 8 KB)
                   [66560, 132096)      V image of the workgroup's 256 keys
Slice / V images are the dual-use layout "8-row x 32-column subtiles" (conflict-free for row reads AND transposed
reads, 2 + 2 address registers): off(row, ch) = 2048 (row >> 3) + 512 (ch >> 2) + 64 (row & 7) + 16 ((ch & 3) ^ ((row >> 2) & 3)).
"""
from __future__ import annotations

from .core import A, Imm, Instr, M0, P, PV, Prog, Reg, S, V, VCC, fimm, imm
from .sched import finish_block, fix_hazards, insert_waits, schedule

CST_BASE = 0
STG_BASE = 1024
STG_BYTES = 16384
V_BASE = STG_BASE + 4 * STG_BYTES          # 66560
LDS_BYTES = V_BASE + 65536                 # 132096

# scalar inputs of the asm statement ("s" operands), all 32 bit
PARAMS = [
    "q_lo", "q_hi", "do_lo", "do_hi", "c_lo", "c_hi", "k_lo", "k_hi", "v_lo", "v_hi", "dk_lo", "dk_hi", "dv_lo", "dv_hi",
    "q_rng", "do_rng", "c_rng", "k_rng", "v_rng", "dk_rng", "dv_rng",
    "q_sn", "do_sn", "k_sn", "v_sn", "dk_sn", "dv_sn",          # row strides, bytes
    "q_hs", "do_hs", "c_hs",                                    # head strides, bytes
    "nq", "g", "q_row0", "kb0", "pos0", "W", "ns", "nrows", "cdelta", "c_log2", "scale",
    "pk_lo", "pk_hi", "pv_lo", "pv_hi", "p_rng",               # f32 partial dK / dV of a split sweep (p_rng = 0: none)
]
# q_*/do_*/c_* bases point at (first head of the group, the sequence's first row); q_row0 = first row of the first slice
# the block sweeps; kb0 = first key of the block; pos0 = position of query row 0 among the keys (N_kv - N_q);
# nrows = query rows of the sequence; cdelta = byte offset of the -Delta row of a head behind its -LSE/scale row.


class Alloc:
    def __init__(self, kind, lo, hi):
        self.kind, self.next, self.hi = kind, lo, hi
        self.names = {}

    def __call__(self, name, n=1, align=1):
        self.next = (self.next + align - 1) // align * align
        r = Reg(self.kind, self.next, n)
        self.next += n
        assert self.next <= self.hi + 1, "out of %s registers at %s" % (self.kind, name)
        self.names[name] = r
        return r


class DkdvGen:
    def __init__(self, dtype="bf16", sched=True, vfirst=4, sfirst=56, npool=12, stamps=False, ablate=(), dma_t0=40, dma_dt=125, D=128, ahead=3, dead=True, half_edges=True):
        assert dtype in ("bf16", "f16") and D in (64, 80, 96, 128)
        self.dtype = dtype
        self.do_sched = sched
        self.partials = True            # the f32 second epilogue of split sweeps
        self.dead = dead                # a third trip body for trips in which no row sees any key of the wave: no MFMA, no VALU
        self.half_edges = half_edges    # edge trips in which no row sees one of the wave's two 32-key blocks run on the other one
        self.ahead = ahead              # slices the LDS-DMA runs ahead of the trip (3 = what the 4-deep ring allows; 2: experiment)
        assert ahead in (2, 3)
        # head dim: DK k-steps of 16 in the d contractions, DB 32-wide output blocks of dK^T / dV^T, NCH valid 16-byte
        # chunks per row.  The LDS images keep 256-byte rows for every head dim; chunks >= NCH are fetched through
        # out-of-range offsets (zeros) when a 128-byte half holds some valid chunks, halves with none are not fetched.
        self.D, self.DK, self.DB, self.NCH = D, D // 16, (D + 31) // 32, D // 8
        self.HALVES = 2 if D > 64 else 1
        self.stamps = stamps            # diagnostic build: s_memtime around the loop head, sums stored per workgroup
        self.dma_t0, self.dma_dt = dma_t0, dma_dt
        self.ablate = set(ablate)       # timing-only builds (wrong results): parts of the trip left out, see trip_body
        if stamps:
            npool = min(npool, 10 if stamps == "phases" else 11)
        self.va = Alloc("v", vfirst, 255)
        self.sa = Alloc("s", sfirst, 99)      # s100 / s101 are reserved by the compiler
        self.vfirst, self.sfirst = vfirst, sfirst
        va, sa = self.va, self.sa
        # ---------------- VGPRs
        self.KF = [[va("kf%d_%d" % (kbi, ks), 4, 4) for ks in range(self.DK)] for kbi in range(2)]
        self.QROW = [va("qrow%d" % ks, 4, 4) for ks in range(8)]      # (8 slots for every head dim: two of them double as scratch)
        self.POOL = [va("pool%d" % i, 4, 4) for i in range(npool)]   # streamed MFMA operand fragments (LDS -> here -> MFMA)
        self.SACC = [va("sacc%d" % kbi, 16, 4) for kbi in range(2)]
        self.DPACC = [va("dpacc%d" % kbi, 16, 4) for kbi in range(2)]
        self.PPK = [[va("ppk%d_%d" % (kbi, s), 4, 4) for s in range(2)] for kbi in range(2)]
        self.lane = va("lane")
        self.l_row_e = va("l_row_e")       # lane part of the row-read address (k-step even), slice / V images
        self.l_tr0 = va("l_tr0")           # lane part of the transposed-read address (first 8-row half)
        self.l_c = va("l_c")               # lane part of the row-constant read address (16 h)
        self.a_row_e, self.a_row_o = va("a_row_e"), va("a_row_o")
        self.a_rown_e, self.a_rown_o = va("a_rown_e"), va("a_rown_o")
        self.a_tr0, self.a_tr1 = va("a_tr0"), va("a_tr1")
        self.a_v_e, self.a_v_o = va("a_v_e"), va("a_v_o")
        self.a_c, self.a_cn = va("a_c"), va("a_cn")
        self.vo_q = [va("vo_q0"), va("vo_q1")]      # running source offsets of the wave's pieces (second half: head dims > 64)
        self.vo_d = [va("vo_d0"), va("vo_d1")]
        self.vo_c, self.vo_ce = va("vo_c"), va("vo_ce")
        self.v_oob = va("v_oob")           # a byte offset no descriptor covers
        self.lane31 = va("lane31")
        self.v_kh = va("v_kh")
        self.v_weff = [va("v_weff0"), va("v_weff1")]
        self.v_d = [va("v_d0"), va("v_d1")]
        # scratch of the prologue / epilogue lives in registers the loop owns (the last two Q row fragments: loaded at
        # the very end of the prologue, dead in the epilogue); only tmp[0] is used inside the loop (edge trips)
        self.tmp = [va("tmp0"), self.QROW[6][0], self.QROW[6][1], self.QROW[6][2]]
        self.vo_k = [self.QROW[6][3], self.QROW[7][0]]      # K fragment load / dK, dV store offsets
        # ---------------- AGPRs
        self.DV = [[A((db * 2 + kbi) * 16, 16) for kbi in range(2)] for db in range(self.DB)]
        self.DKA = [[A(128 + (db * 2 + kbi) * 16, 16) for kbi in range(2)] for db in range(self.DB)]
        # ---------------- SGPRs
        self.d_q, self.d_do, self.d_c = sa("d_q", 4, 4), sa("d_do", 4, 4), sa("d_c", 4, 4)
        self.d_x = sa("d_x", 4, 4)               # K / V / dK / dV descriptor (prologue, epilogue)
        self.s_wave = sa("s_wave")
        self.s_t, self.s_n = sa("s_t"), sa("s_n")
        self.s_st, self.s_stn, self.s_std = sa("s_st"), sa("s_stn"), sa("s_std")
        self.s_cst, self.s_cstn, self.s_cstd = sa("s_cst"), sa("s_cstn"), sa("s_cstd")
        self.s_full = sa("s_full")
        self.s_ldq, self.s_ldh, self.s_ldrow = sa("s_ldq"), sa("s_ldh"), sa("s_ldrow")
        self.s_cq, self.s_q0p = sa("s_cq"), sa("s_q0p")
        self.s_kw63, self.s_kww, self.s_allsink = sa("s_kw63"), sa("s_kww"), sa("s_allsink")
        self.s_frng = self.s_allsink       # half_edges: the length of the wave's "full" range of q0p (the flag is folded into it)
        self.s_stepq, self.s_stepd = sa("s_stepq"), sa("s_stepd")     # 32 rows in bytes
        self.s_spanq, self.s_spand, self.s_spanc = sa("s_spanq"), sa("s_spand"), sa("s_spanc")
        self.s_wofs = sa("s_wofs")                # 2048 * wave: the wave's two DMA pieces inside a slice image
        self.s_tmp = [sa("s_tmp%d" % i) for i in range(4)]
        self.pool_next = 0
        if stamps:
            self.v_sum = [va("sum_top"), va("sum_body"), va("sum_n")] + \
                ([va("sum_ph%d" % i) for i in range(4)] if stamps == "phases" else [])
            self.s_ta, self.s_tb = self.d_x[0:2], self.d_x[2:4]      # the prologue / epilogue descriptor is idle in the loop

    def params(self):
        return PARAMS + (["dbg_lo", "dbg_hi", "bid"] if self.stamps else [])

    def emit_stamp(self, p: Prog, dst):
        p.add(Instr("s_memtime", [dst], [], kind="fence"))
        p.s_waitcnt(lgkmcnt=0)

    def emit_phase_stamp(self, p: Prog, k: int):
        """diagnostic builds (stamps="phases"): a fenced s_memtime between the four MFMA phases of a trip; the time since
        the previous stamp is added to sum k (the phases cannot overlap in such a build: read shares, not lengths)"""
        if self.stamps != "phases":
            return
        self.emit_stamp(p, self.s_ta)
        p.s_sub_u32(self.s_tmp[3], self.s_ta[0], self.s_tb[0])
        p.v_add_u32(self.v_sum[k], self.s_tmp[3], self.v_sum[k])
        p.s_mov(self.s_tb[0], self.s_ta[0])

    # ------------------------------------------------------------------ helpers
    def pool(self):
        r = self.POOL[self.pool_next % len(self.POOL)]
        self.pool_next += 1
        return r

    # LDS-DMA of one slice (Q, dO, row constants) into ring stage s_std / s_cstd, with the running source offsets
    def emit_dma_issue(self, p: Prog, spread: bool = False):
        """`spread`: inside a trip the five pieces get early, staggered deadlines (dma_t0 + i dma_dt cycles), so that the
        fetch is three trips ahead of its use in time as well.  A piece costs ~60 cycles of issue wherever it sits
        (fenced phase stamps: the S-chain phase, which carries them, takes 859 cycles against 512 of MFMA); placing them
        in the dV / dK half instead measured 0.7 % slower."""
        dl = (lambda i: {"alap": self.dma_t0 + self.dma_dt * i}) if spread else (lambda i: {})
        p.s_add_m0(self.s_std, self.s_wofs, note="Q piece 0 of this wave")
        p.buffer_load_lds(16, self.vo_q[0], self.d_q, 0, mem=("dma_stage",)).mods.update(dl(0))
        if self.HALVES == 2:
            p.s_add_m0(M0, 1024)
            p.buffer_load_lds(16, self.vo_q[1], self.d_q, 0, mem=("dma_stage",)).mods.update(dl(1))
        p.s_add_m0(M0, 8192 - 1024 * (self.HALVES - 1), note="dO piece 0")
        p.buffer_load_lds(16, self.vo_d[0], self.d_do, 0, mem=("dma_stage",)).mods.update(dl(2))
        if self.HALVES == 2:
            p.s_add_m0(M0, 1024)
            p.buffer_load_lds(16, self.vo_d[1], self.d_do, 0, mem=("dma_stage",)).mods.update(dl(3))
        # row constants: lanes 0..31 -LSE/scale, 32..63 -Delta of rows q0 .. q0+31; rows >= nrows are forced out of range
        # (they read 0: p = exp2(0) stays finite against the zero Q / dO rows, nothing reaches dK / dV)
        p.s_sub_i32(self.s_tmp[0], P("nrows"), self.s_ldrow, note="rows left in the sequence")
        p.v_cmp("gt_i32", self.s_tmp[0], self.lane31)
        p.v_cndmask(self.vo_ce, self.v_oob, self.vo_c)
        p.s_mov_m0(self.s_cstd)
        p.buffer_load_lds(4, self.vo_ce, self.d_c, 0, mem=("dma_stage",)).mods.update(dl(4))

    def emit_half1(self, p: Prog, dst, src, cslot, cbl, tmp):
        """dst = source offset of the lane in the SECOND 128-byte half of its row (src + 128), or an out-of-range offset
        when that chunk (8 + 4 cbl + cslot) lies beyond the head dim: the LDS image keeps zeros there"""
        if self.NCH == 16:
            p.v_add_u32(dst, 128, src)
            return
        p.v_lshl_add_u32(tmp, cbl, 2, cslot)
        p.v_add_u32(dst, 128, src)
        p.v_cmp("gt_u32", self.NCH - 8, tmp)
        p.v_cndmask(dst, self.v_oob, dst)

    def emit_dma_step(self, p: Prog):
        """advance the source offsets to the next slice of the same head"""
        p.v_add_u32(self.vo_q[0], self.s_stepq, self.vo_q[0])
        p.v_add_u32(self.vo_d[0], self.s_stepd, self.vo_d[0])
        if self.HALVES == 2:
            p.v_add_u32(self.vo_q[1], self.s_stepq, self.vo_q[1])
            p.v_add_u32(self.vo_d[1], self.s_stepd, self.vo_d[1])
        p.v_add_u32(self.vo_c, 128, self.vo_c)
        p.s_add_u32(self.s_ldq, self.s_ldq, 1)
        p.s_add_u32(self.s_ldrow, self.s_ldrow, 32)

    def emit_dma_headchange(self, p: Prog, lbl_done: str):
        """the next slice to fetch starts a new q head: move the descriptors, rewind the offsets; past the last head
        the descriptors get zero records (the remaining fetches of the pipeline then touch no memory)"""
        p.s_mov(self.s_ldq, 0)
        p.s_mov(self.s_ldrow, P("q_row0"))
        p.s_add_u32(self.s_ldh, self.s_ldh, 1)
        for d, hs in ((self.d_q, "q_hs"), (self.d_do, "do_hs"), (self.d_c, "c_hs")):
            p.s_add_u32(d[0], d[0], P(hs))
            p.s_addc_u32(d[1], d[1], 0)
        p.v_sub_u32(self.vo_q[0], self.vo_q[0], self.s_spanq)
        p.v_sub_u32(self.vo_d[0], self.vo_d[0], self.s_spand)
        if self.HALVES == 2:
            p.v_sub_u32(self.vo_q[1], self.vo_q[1], self.s_spanq)
            p.v_sub_u32(self.vo_d[1], self.vo_d[1], self.s_spand)
        p.v_sub_u32(self.vo_c, self.vo_c, self.s_spanc)
        p.s_cmp("lt_u32", self.s_ldh, P("g"))
        p.s_cbranch("scc1", lbl_done)
        p.s_mov(self.d_q[2], 0)
        p.s_mov(self.d_do[2], 0)
        p.s_mov(self.d_c[2], 0)

    def emit_dma_headcheck(self, p: Prog, uniq: str):
        lbl = "L_nohead_%s%%=" % uniq
        p.s_cmp("lt_u32", self.s_ldq, P("nq"))
        p.s_cbranch("scc1", lbl)
        self.emit_dma_headchange(p, lbl)
        p.label(lbl)

    def emit_class(self, p: Prog):
        """s_full = class of (the wave's 64 keys) x (the 32 rows at s_q0p): 1 <=> every key is causal for every row and
        (all keys are sinks or all are inside every row's window): kw63 <= q0p and (kw63 < ns or kw0 + W > q0p + 31);
        2 (dead=True builds) <=> no row sees any key: every key lies behind every row (kw0 > q0p + 31), or the wave holds no
        sink key and every key has left every row's window (q0p - kw63 >= W) - the trips a wave spends on slices that only
        the OTHER waves' keys can see (6 of every sweep in lock step); else 0 (edge: per-element masks)"""
        t0, t1 = self.s_tmp[1], self.s_tmp[2]
        if self.half_edges:
            # (the only class the scheduled bodies need is "full": q0p in [kw63, kw0 + W - 32] - or [kw63, oo) for a wave of
            # sink keys only -, one unsigned compare against a per-wave constant; the selector tells the rest apart)
            p.s_sub_u32(t0, self.s_q0p, self.s_kw63)
            p.s_cmp("lt_u32", t0, self.s_frng)
            p.s_cselect(self.s_full, 1, 0)
            return
        p.s_cmp("le_i32", self.s_kw63, self.s_q0p)
        p.s_cselect(t0, 1, 0)
        p.s_add_u32(t1, self.s_q0p, 31)
        p.s_cmp("gt_i32", self.s_kww, t1)
        p.s_cselect(t1, 1, 0)
        p.s_or_b32(t1, t1, self.s_allsink)
        p.s_and_b32(self.s_full, t0, t1)
        if self.dead and not self.half_edges:       # (with half_edges the selector behind the loop head finds the dead trips: nothing
            t2 = self.s_tmp[0]                      # of this in the scheduled bodies)
            p.s_add_u32(t1, self.s_q0p, 94)
            p.s_cmp("gt_i32", self.s_kw63, t1)                  # kw0 = kw63 - 63 > q0p + 31
            p.s_cselect(t0, 1, 0)
            p.s_sub_u32(t2, self.s_kw63, 63)
            p.s_cmp("ge_i32", t2, P("ns"))                      # no sink key in the wave
            p.s_cselect(t2, 1, 0)
            p.s_add_u32(t1, self.s_kww, 63)
            p.s_cmp("ge_i32", self.s_q0p, t1)                   # q0p >= kw63 + W
            p.s_cselect(t1, 1, 0)
            p.s_and_b32(t1, t1, t2)
            p.s_or_b32(t0, t0, t1)
            p.s_cmp("lg_u32", t0, 0)
            p.s_cselect(self.s_full, 2, self.s_full)

    def emit_next_prefetch(self, p: Prog, deadline=None):
        """operands of the next trip's S chains: Q row fragments and the -LSE/scale rows as initial accumulators.
        `deadline`: latest issue time inside a trip (nothing in the trip needs them, but the next trip starts with
        them: they have to be on their way well before the trip ends)"""
        n0 = len(p.items)
        self.emit_qrow_prefetch(p, self.a_rown_e, self.a_rown_o)
        for kbi in range(2):
            for g4 in range(4):
                p.ds_read_b128(self.SACC[kbi][4 * g4:4 * g4 + 4], self.a_cn, 32 * g4, mem=("stage_r",))
        if deadline is not None:
            for k, it in enumerate(p.items[n0:]):
                it.mods["alap"] = deadline + 26 * k      # one per MFMA slot or so: no burst on the shared LDS

    def emit_qrow_prefetch(self, p: Prog, e, o):
        for ks in range(self.DK):
            base = o if ks & 1 else e
            p.ds_read_b128(self.QROW[ks], base, 512 * (ks >> 1), mem=("stage_r",), note="Q rows, k-step %d" % ks)

    # ------------------------------------------------------------------ prologue
    def prologue(self) -> Prog:
        p = Prog()
        t0, t1, t2, t3 = self.tmp
        lane, wv = self.lane, self.s_wave
        p.v_and(lane, 63, PV("tid"))
        p.v_lshrrev(t0, 6, PV("tid"))
        p.v_readfirstlane(wv, t0)
        p.v_and(self.lane31, 31, lane)
        p.v_mov(self.v_oob, imm(0x7FFFFFF0))
        # r = lane & 31, h = lane >> 5 ; l_row_e = 2048 (r >> 3) + 64 (r & 7) + 16 (h ^ ((r >> 2) & 3))
        p.v_lshrrev(t0, 3, self.lane31)                       # r >> 3
        p.v_lshlrev(t0, 11, t0)
        p.v_and(t1, 7, lane)                                  # r & 7
        p.v_lshl_add_u32(t0, t1, 6, t0)
        p.v_bfe_u32(t1, lane, 2, 2)                           # (r >> 2) & 3
        p.v_lshrrev(t2, 5, lane)                              # h
        p.v_xor(t1, t1, t2)
        p.v_lshl_add_u32(self.l_row_e, t1, 4, t0)
        p.v_lshlrev(self.l_c, 4, t2)                          # 16 h
        # l_tr0 = 64 (4 h + q4) + 16 ((2 g1 + (p4 >> 1)) ^ h) + 8 (p4 & 1);  q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1
        p.v_bfe_u32(t0, lane, 2, 2)                           # q4
        p.v_lshl_add_u32(t0, t2, 2, t0)                       # 4 h + q4
        p.v_lshlrev(t0, 6, t0)
        p.v_bfe_u32(t1, lane, 4, 1)                           # g1
        p.v_bfe_u32(t3, lane, 1, 1)                           # p4 >> 1
        p.v_lshl_add_u32(t1, t1, 1, t3)
        p.v_xor(t1, t1, t2)
        p.v_lshl_add_u32(t0, t1, 4, t0)
        p.v_and(t1, 1, lane)
        p.v_lshl_add_u32(self.l_tr0, t1, 3, t0)
        # V image row-read addresses: V_BASE + 16384 wave + l_row_e
        p.s_lshl_b32(self.s_tmp[0], wv, 14)
        p.s_add_u32(self.s_tmp[0], self.s_tmp[0], V_BASE)
        p.v_add_u32(self.a_v_e, self.s_tmp[0], self.l_row_e)
        p.v_xor(self.a_v_o, 32, self.a_v_e)
        p.s_lshl_b32(self.s_wofs, wv, 11)

        # ---- descriptors of the three streamed tensors
        for d, nm in ((self.d_q, "q"), (self.d_do, "do"), (self.d_c, "c")):
            p.s_mov(d[0], P(nm + "_lo"))
            p.s_mov(d[1], P(nm + "_hi"))
            p.s_mov(d[2], P(nm + "_rng"))
            p.s_mov(d[3], 0x00020000)

        # ---- V image of this wave's 64 keys: 16 LDS-DMA pieces (8 row groups x 2 halves of 128 bytes)
        p.s_mov(self.d_x[0], P("v_lo"))
        p.s_mov(self.d_x[1], P("v_hi"))
        p.s_mov(self.d_x[2], P("v_rng"))
        p.s_mov(self.d_x[3], 0x00020000)
        # lane pattern of a piece: rr = (lane >> 2) & 7, slot = lane & 3, cbl = lane >> 5
        rr, slot = t0, t1
        p.v_bfe_u32(rr, lane, 2, 3)
        p.v_and(slot, 3, lane)
        p.s_lshl_b32(self.s_tmp[0], wv, 6)
        p.s_add_u32(self.s_tmp[0], self.s_tmp[0], P("kb0"))       # first key of the wave
        p.s_mul_i32(self.s_tmp[1], self.s_tmp[0], P("v_sn"))      # its byte offset
        p.v_mul_lo_u32(t2, rr, P("v_sn"))
        p.v_add_u32(t2, self.s_tmp[1], t2)                        # row (wave key0 + rr) bytes
        p.v_lshrrev(t3, 5, lane)
        p.v_lshl_add_u32(t2, t3, 6, t2)                           # + 64 cbl
        p.s_lshl_b32(self.s_tmp[2], wv, 14)
        p.s_add_u32(self.s_tmp[2], self.s_tmp[2], V_BASE)         # LDS base of the wave's part
        p.s_lshl_b32(self.s_tmp[3], P("v_sn"), 3)                 # 8 rows
        for rgl in range(8):
            # x = (2 rgl + (rr >> 2)) & 3 ; chunk-in-block = slot ^ x
            p.v_lshrrev(t3, 2, rr)
            p.v_add_u32(t3, 2 * rgl, t3)
            p.v_and(t3, 3, t3)
            p.v_xor(t3, t3, slot)
            p.v_lshl_add_u32(self.vo_k[0], t3, 4, t2)
            p.s_add_m0(self.s_tmp[2], 2048 * rgl)
            p.buffer_load_lds(16, self.vo_k[0], self.d_x, 0, mem=("v_img",))
            if self.HALVES == 2:
                p.v_lshrrev(self.vo_k[1], 5, lane)
                self.emit_half1(p, self.vo_k[1], self.vo_k[0], t3, self.vo_k[1], self.vo_ce)
                p.s_add_m0(M0, 1024)
                p.buffer_load_lds(16, self.vo_k[1], self.d_x, 0, mem=("v_img",))
            if rgl < 7:
                p.v_add_u32(t2, self.s_tmp[3], t2)

        # ---- K fragments: lane (r, h) of key block kbi holds K[key][16 ks + 8 h ..+8)
        p.s_mov(self.d_x[0], P("k_lo"))
        p.s_mov(self.d_x[1], P("k_hi"))
        p.s_mov(self.d_x[2], P("k_rng"))
        p.s_mul_i32(self.s_tmp[1], self.s_tmp[0], P("k_sn"))
        p.v_mul_lo_u32(t2, self.lane31, P("k_sn"))
        p.v_add_u32(t2, self.s_tmp[1], t2)
        p.v_lshrrev(t3, 5, lane)
        p.v_lshl_add_u32(self.vo_k[0], t3, 4, t2)
        p.s_lshl_b32(self.s_tmp[1], P("k_sn"), 5)
        p.v_add_u32(self.vo_k[1], self.s_tmp[1], self.vo_k[0])
        for kbi in range(2):
            for ks in range(self.DK):
                p.buffer_load(self.KF[kbi][ks], self.vo_k[kbi], self.d_x, 0, offset=32 * ks)

        # ---- mask constants: key0 = wave key0 + r ; v_kh = key0 - 4 h ; weff = key < ns ? 2^31 : W
        p.v_add_u32(t2, self.s_tmp[0], self.lane31)           # key of block 0
        p.v_lshrrev(t3, 5, lane)
        p.v_lshlrev(t3, 2, t3)
        p.v_sub_u32(self.v_kh, t2, t3)
        p.v_mov(t3, P("W"))
        p.v_mov(t0, imm(0x80000000))
        p.v_cmp("le_u32", P("ns"), t2)                        # not a sink key
        p.v_cndmask(self.v_weff[0], t0, t3)                   # (a literal next to VCC would be two constant-bus reads)
        p.v_add_u32(t2, 32, t2)
        p.v_cmp("le_u32", P("ns"), t2)
        p.v_cndmask(self.v_weff[1], t0, t3)
        # wave-level classification constants
        p.s_add_u32(self.s_kw63, self.s_tmp[0], 63)
        p.s_add_u32(self.s_kww, self.s_tmp[0], P("W"))
        p.s_cmp("lt_i32", self.s_kw63, P("ns"))
        p.s_cselect(self.s_allsink, 1, 0)
        if self.half_edges:
            # full <=> kw63 <= q0p and (all sinks or kw0 + W > q0p + 31): q0p - kw63 <u (kw0 + W - 31) - kw63 = W - 94, or <u 2^31
            p.s_sub_i32(self.s_tmp[1], P("W"), 94)
            p.s_max_i32(self.s_tmp[1], self.s_tmp[1], 0)
            p.s_cmp("lg_u32", self.s_allsink, 0)
            p.s_cselect(self.s_frng, 0x7FFFFFFF, self.s_tmp[1])

        # ---- slice stream: source offsets of this wave's pieces.  piece (2 wave + e): rows 8 wave + rr, 16-byte chunk
        #      4 (2 e + cbl) + (slot ^ ((2 wave + (rr >> 2)) & 3))
        p.v_bfe_u32(rr, lane, 2, 3)
        p.v_and(slot, 3, lane)
        p.v_lshrrev(t3, 2, rr)
        p.s_lshl_b32(self.s_tmp[1], wv, 1)
        p.v_add_u32(t3, self.s_tmp[1], t3)
        p.v_and(t3, 3, t3)
        p.v_xor(t3, t3, slot)                                 # chunk in block
        p.v_lshrrev(t2, 5, lane)
        p.v_lshl_add_u32(t3, t2, 2, t3)                       # + 4 cbl
        p.v_lshlrev(t3, 4, t3)                                # bytes
        p.s_lshl_b32(self.s_tmp[1], wv, 3)
        p.s_add_u32(self.s_tmp[1], self.s_tmp[1], P("q_row0"))    # first row of the wave's pieces in slice 0
        p.v_add_u32(t2, self.s_tmp[1], rr)                    # row
        for vo, sn in ((self.vo_q, "q_sn"), (self.vo_d, "do_sn")):
            p.v_mul_lo_u32(vo[0], t2, P(sn))
            p.v_add_u32(vo[0], vo[0], t3)
            if self.HALVES == 2:
                p.v_lshrrev(rr, 4, t3)                            # chunk index inside the half: 4 cbl + (slot ^ x)
                self.emit_half1(p, vo[1], vo[0], rr, 0, slot)     # (rr / slot are free again: only `row` t2 and t3 live)
        # row constants: lane L < 32 -> -LSE/scale of row q_row0 + L ; L >= 32 -> -Delta of row q_row0 + L - 32
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
        p.s_mov(self.s_ldq, 0)
        p.s_mov(self.s_ldh, 0)
        p.s_mov(self.s_ldrow, P("q_row0"))
        p.s_mul_i32(self.s_n, P("nq"), P("g"))
        # no trips at all (a block no row can see): nothing to fetch
        p.s_cmp("lg_u32", self.s_n, 0)
        p.s_cbranch("scc1", "L_some%=")
        p.s_mov(self.d_q[2], 0)
        p.s_mov(self.d_do[2], 0)
        p.s_mov(self.d_c[2], 0)
        p.label("L_some%=")

        # ---- accumulators
        for acc in (self.DV, self.DKA):
            for db in range(self.DB):
                for kbi in range(2):
                    for i in range(16):
                        p.v_accvgpr_write(acc[db][kbi][i], 0)

        # ---- first slices into stages 0, 1 (, 2)
        for j in range(self.ahead):
            p.s_mov(self.s_std, STG_BASE + j * STG_BYTES)
            p.s_mov(self.s_cstd, CST_BASE + j * 256)
            self.emit_dma_headcheck(p, "pro%d" % j)
            self.emit_dma_issue(p)
            self.emit_dma_step(p)
        p.s_waitcnt(vmcnt=(self.ahead - 1) * (2 * self.HALVES + 1), note="V image, K fragments, slice 0 landed (slices 1, 2 in flight)")
        p.s_barrier()
        # operands of the first trip
        p.v_add_u32(self.a_rown_e, STG_BASE, self.l_row_e)
        p.v_xor(self.a_rown_o, 32, self.a_rown_e)
        p.v_add_u32(self.a_cn, CST_BASE, self.l_c)
        self.emit_next_prefetch(p)
        p.s_mov(self.s_t, 0)
        p.s_mov(self.s_cq, 0)
        p.s_add_u32(self.s_q0p, P("q_row0"), P("pos0"))
        self.emit_class(p)
        p.s_mov(self.s_st, STG_BASE)
        p.s_mov(self.s_stn, STG_BASE + STG_BYTES)
        p.s_mov(self.s_std, STG_BASE + self.ahead * STG_BYTES)
        p.s_mov(self.s_cst, CST_BASE)
        p.s_mov(self.s_cstn, CST_BASE + 256)
        p.s_mov(self.s_cstd, CST_BASE + 256 * self.ahead)
        if self.stamps:
            for r in self.v_sum:
                p.v_mov(r, 0)
            p.s_waitcnt(lgkmcnt=0)
            self.emit_stamp(p, self.s_tb)
        return p

    # ------------------------------------------------------------------ loop head (scalar, branchy)
    def loop_top(self) -> Prog:
        """common path: 8 scalar instructions + the trip's wait / barrier; the two head changes are out of line"""
        p = Prog()
        p.label("L_top%=")
        if self.stamps:      # body time of the trip that just ended: now - (stamp behind its barrier)
            self.emit_stamp(p, self.s_ta)
            p.s_sub_u32(self.s_tmp[3], self.s_ta[0], self.s_tb[0])
            p.v_add_u32(self.v_sum[1], self.s_tmp[3], self.v_sum[1])
        p.s_cmp("ge_u32", self.s_t, self.s_n)
        p.s_cbranch("scc1", "L_done%=")
        p.s_cmp("lt_u32", self.s_ldq, P("nq"))
        p.s_cbranch("scc0", "L_dmahead%=")
        p.label("L_top_a%=")
        p.s_cmp("lt_u32", self.s_cq, P("nq"))
        p.s_cbranch("scc0", "L_cmphead%=")
        p.label("L_top_b%=")
        p.s_cmp("lg_u32", self.s_full, 0)
        p.s_waitcnt(vmcnt=(self.ahead - 2) * (2 * self.HALVES + 1), note="slice t+1 landed (own pieces); slice t+2 may be in flight")
        p.s_barrier()
        p.s_waitcnt(lgkmcnt=0, note="S-chain operands of this slice (fetched at the end of the last trip)")
        if self.stamps:      # head time: loop head + waits + barrier
            p.s_mov(self.s_tmp[3], self.s_full)
            self.emit_stamp(p, self.s_tb)
            p.s_sub_u32(self.s_tmp[2], self.s_tb[0], self.s_ta[0])
            p.v_add_u32(self.v_sum[0], self.s_tmp[2], self.v_sum[0])
            p.v_add_u32(self.v_sum[2], 1, self.v_sum[2])
            p.s_cmp("lg_u32", self.s_tmp[3], 0)
        p.s_cbranch("scc0", "L_edgesel%=" if self.half_edges else "L_edge%=")
        if self.dead and not self.half_edges:
            p.s_cmp("eq_u32", self.s_tmp[3] if self.stamps else self.s_full, 2)
            p.s_cbranch("scc1", "L_dead%=")
        return p

    def out_of_line(self) -> Prog:
        p = Prog()
        p.label("L_dmahead%=")
        self.emit_dma_headchange(p, "L_top_a%=")
        p.s_branch("L_top_a%=")
        p.label("L_cmphead%=")
        p.s_mov(self.s_cq, 0)
        p.s_add_u32(self.s_q0p, P("q_row0"), P("pos0"))
        self.emit_class(p)
        p.s_branch("L_top_b%=")
        if self.half_edges:
            # An edge trip (only those come here): is one of the wave's two 32-key blocks out of every row's reach?  Block kb
            # (keys kw0 + 32 kb .. + 31) is dead when all its keys lie behind every row (kw0 + 32 kb > q0p + 31) or when it
            # holds no sink key and every key has left every row's window (q0p - (kw0 + 32 kb + 31) >= W).  The long case:
            # block 0's sink tail - rows far beyond the window sweep 64 keys for the few sink keys among the first 32.
            t0, t1, t2, t3 = self.s_tmp[0], self.s_tmp[1], self.s_tmp[2], self.s_tmp[3]
            p.label("L_edgesel%=")
            dead = []
            for kb, acc in ((0, t2), (1, t3)):
                p.s_add_u32(t0, self.s_q0p, 94 - 32 * kb)           # kw63 - 63 + 32 kb > q0p + 31
                p.s_cmp("gt_i32", self.s_kw63, t0)
                p.s_cselect(acc, 1, 0)
                p.s_sub_u32(t0, self.s_kw63, 63 - 32 * kb)          # first key of the block >= ns
                p.s_cmp("ge_i32", t0, P("ns"))
                p.s_cselect(t0, 1, 0)
                p.s_add_u32(t1, self.s_kww, 31 + 32 * kb)           # q0p >= kw0 + 32 kb + 31 + W
                p.s_cmp("ge_i32", self.s_q0p, t1)
                p.s_cselect(t1, 1, 0)
                p.s_and_b32(t0, t0, t1)
                p.s_or_b32(acc, acc, t0)
            if self.dead:
                p.s_and_b32(t0, t2, t3)
                p.s_cmp("lg_u32", t0, 0)
                p.s_cbranch("scc1", "L_dead%=")                    # both: the wave has nothing to do in this trip
            p.s_cmp("lg_u32", t3, 0)
            p.s_cbranch("scc1", "L_e0%=")                          # second block dead: the first one alone
            p.s_cmp("lg_u32", t2, 0)
            p.s_cbranch("scc1", "L_e1%=")
            p.s_branch("L_edge%=")
        return p

    # ------------------------------------------------------------------ one trip
    def trip_body(self, edge: bool, live=(0, 1)) -> Prog:
        """live: the wave's 32-key blocks that take part (edge trips: the other one is seen by no row of the slice)"""
        p = Prog()
        dt = self.dtype
        self.pool_next = 0
        # stage-relative addresses of this trip
        p.v_add_u32(self.a_row_e, self.s_st, self.l_row_e)
        p.v_xor(self.a_row_o, 32, self.a_row_e)
        p.v_add_u32(self.a_tr0, self.s_st, self.l_tr0)
        p.v_xor(self.a_tr1, 32, self.a_tr0)
        p.v_add_u32(self.a_c, self.s_cst, self.l_c)
        p.v_add_u32(self.a_rown_e, self.s_stn, self.l_row_e)
        p.v_xor(self.a_rown_o, 32, self.a_rown_e)
        p.v_add_u32(self.a_cn, self.s_cstn, self.l_c)
        # dP accumulators start from -Delta: registers 4 g4 .. +3 <- rows 8 g4 + 4 h + {0..3}
        for kbi in live:
            for g4 in range(4):
                p.ds_read_b128(self.DPACC[kbi][4 * g4:4 * g4 + 4], self.a_c, 128 + 32 * g4, mem=("stage_r",))
        # fetch slice t + 3
        self.emit_dma_issue(p, spread=True)
        self.emit_dma_step(p)
        # ---- [A] S' = Q K^T - LSE/scale, key block after key block (operands and initial accumulators were fetched
        #      at the end of the previous trip)
        for kbi in live:
            for ks in range(self.DK):
                p.mfma(dt, self.SACC[kbi], self.QROW[ks], self.KF[kbi][ks], self.SACC[kbi], tag="S")
        self.emit_phase_stamp(p, 3)
        if edge:
            p.v_sub_u32(self.v_d[0], self.s_q0p, self.v_kh, note="(q0 + pos0 + 4 h) - key")
            p.v_sub_u32(self.v_d[1], self.v_d[0], 32)
        # ---- P = exp2(c S') (+ mask), packed
        for kbi in live:
            for v in range(16):
                x = self.SACC[kbi][v]
                if "mulc" not in self.ablate:
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
        # ---- [B] dP' = dO V^T - Delta, k-step after k-step (dO row fragment shared by the two key blocks)
        for ks in range(self.DK):
            base = self.a_row_o if ks & 1 else self.a_row_e
            fa = self.pool()
            p.ds_read_b128(fa, base, 8192 + 512 * (ks >> 1), mem=("stage_r",), note="dO rows, k-step %d" % ks)
            for kbi in live:
                vb = self.a_v_o if ks & 1 else self.a_v_e
                fv = self.pool()
                p.ds_read_b128(fv, vb, 8192 * kbi + 512 * (ks >> 1), mem=("v_img_r",), note="V rows")
                p.mfma(dt, self.DPACC[kbi], fa, fv, self.DPACC[kbi], tag="dP")
        self.emit_phase_stamp(p, 4)
        # ---- dS = P dP', packed IN PLACE into dP registers [kbi][4 s + j] (the S registers are free from here on: the
        #      next trip's initial accumulators go there while this trip's last MFMAs run)
        for kbi in live:
            for v in range(16):
                p.v_mul_f32(self.DPACC[kbi][v], self.SACC[kbi][v], self.DPACC[kbi][v])
            for s in range(2):
                for j in range(4):
                    p.v_cvt_pk(dt, self.DPACC[kbi][4 * s + j], self.DPACC[kbi][8 * s + 2 * j], self.DPACC[kbi][8 * s + 2 * j + 1])
        # ---- [C] dV^T += dO^T P ; [D] dK^T += Q^T dS   (A operands: transposed reads, rows 16 s + 8 half + ..)
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
            self.emit_phase_stamp(p, 5 if which == "dV" else 6)
        # operands of the next trip (stage t + 1, landed before this trip's barrier)
        n_mfma = (4 * self.DK + 8 * self.DB) * len(live) // 2
        self.emit_next_prefetch(p, deadline=max(200, n_mfma * 32 - 900))
        self.emit_trip_state(p)
        self.apply_ablate(p)
        return p

    def apply_ablate(self, p: Prog):
        """knock-out builds for tools/ab.py (what does a trip cost without ...; wrong results, timing only): "dma" the LDS-DMA
        pieces, "exp" the v_exp, "valu" all VALU of the softmax / dS part, "vfrag" the V fragment reads, "consts" the row
        constants, "tr" the transposed reads, "mfma_s" the S / dP chains, "mfma_acc" the dV / dK MFMAs"""
        if not self.ablate:
            return
        drop = []
        for it in p.items:
            k = it.kind
            if "dma" in self.ablate and k == "dma": drop.append(it)
            if "exp" in self.ablate and k == "trans": drop.append(it)
            if "valu" in self.ablate and (k == "trans" or it.op.startswith(("v_mul_f32", "v_cvt_pk"))): drop.append(it)
            if "vfrag" in self.ablate and k == "ds_read" and "v_img_r" in it.mem_r: drop.append(it)
            if "consts" in self.ablate and k == "ds_read" and it.src[0] in (self.a_c, self.a_cn): drop.append(it)
            if "tr" in self.ablate and it.op == "ds_read_b64_tr_b16": drop.append(it)
            if "mfma_s" in self.ablate and k == "mfma" and it.tag in ("S", "dP"): drop.append(it)
            if "mfma_acc" in self.ablate and k == "mfma" and it.tag in ("dV", "dK"): drop.append(it)
        ids = set(id(x) for x in drop)
        p.items = [it for it in p.items if id(it) not in ids]

    def emit_trip_state(self, p: Prog):
        """scalar state of the next trip (its class assumes the same q head; a head change redoes it out of line)"""
        p.s_add_u32(self.s_q0p, self.s_q0p, 32)
        p.s_add_u32(self.s_t, self.s_t, 1)
        p.s_add_u32(self.s_cq, self.s_cq, 1)
        self.emit_class(p)
        t0 = self.s_tmp[3]
        p.s_mov(self.s_st, self.s_stn)
        p.s_add_u32(t0, self.s_stn, STG_BYTES - STG_BASE)
        p.s_and_b32(t0, t0, 0xFFFF)
        p.s_add_u32(self.s_stn, t0, STG_BASE)
        p.s_add_u32(t0, self.s_std, STG_BYTES - STG_BASE)
        p.s_and_b32(t0, t0, 0xFFFF)
        p.s_add_u32(self.s_std, t0, STG_BASE)
        p.s_mov(self.s_cst, self.s_cstn)
        p.s_add_u32(t0, self.s_cstn, 256)
        p.s_and_b32(self.s_cstn, t0, 1023)
        p.s_add_u32(t0, self.s_cstd, 256)
        p.s_and_b32(self.s_cstd, t0, 1023)

    def dead_body(self) -> Prog:
        """a trip in which no row of the slice sees any key of this wave: the wave keeps the slice stream and its scalar state
        going (its LDS-DMA pieces, the next trip's S-chain operands) and computes nothing"""
        p = Prog()
        self.pool_next = 0
        p.v_add_u32(self.a_rown_e, self.s_stn, self.l_row_e)
        p.v_xor(self.a_rown_o, 32, self.a_rown_e)
        p.v_add_u32(self.a_cn, self.s_cstn, self.l_c)
        self.emit_dma_issue(p, spread=False)
        self.emit_dma_step(p)
        self.emit_next_prefetch(p)
        self.emit_trip_state(p)
        return p

    # ------------------------------------------------------------------ epilogue
    def epilogue(self) -> Prog:
        p = Prog()
        dt = self.dtype
        t0, t1, t2, t3 = self.tmp
        p.label("L_done%=")
        p.s_waitcnt(vmcnt=0, lgkmcnt=0)
        if self.stamps:
            # lane 0..2 of every wave: sum_top, sum_body, trips at dbg[(4 bid + wave) * 4 + lane] (other lanes out of range)
            p.s_mov(self.d_x[0], P("dbg_lo"))
            p.s_mov(self.d_x[1], P("dbg_hi"))
            p.s_mov(self.d_x[3], 0x00020000)
            p.s_lshl_b32(self.s_tmp[0], P("bid"), 2)
            p.s_add_u32(self.s_tmp[0], self.s_tmp[0], self.s_wave)
            p.s_lshl_b32(self.s_tmp[0], self.s_tmp[0], 4)
            p.s_add_u32(self.d_x[2], self.s_tmp[0], 16, note="records end behind this wave's 16 bytes")
            for k, r in enumerate(self.v_sum[:3] if self.stamps != "phases" else [self.v_sum[i] for i in (3, 4, 5, 6)]):
                p.v_mov(t0, self.s_tmp[0])
                p.v_cmp("eq_u32", 0, self.lane)
                p.v_cndmask(t0, self.v_oob, t0)
                p.buffer_store(r, t0, self.d_x, 0, offset=4 * k)
            p.s_waitcnt(vmcnt=0)
        # store offsets: key (block kbi) * row stride + 16 h bytes ; immediate 64 db + 32 gp for the pair of column groups
        p.s_lshl_b32(self.s_tmp[0], self.s_wave, 6)
        p.s_add_u32(self.s_tmp[0], self.s_tmp[0], P("kb0"))
        p.v_add_u32(t2, self.s_tmp[0], self.lane31)
        p.v_lshrrev(t3, 5, self.lane)
        p.v_lshlrev(t3, 4, t3)
        npair = 0
        for which, acc in (("dk", self.DKA), ("dv", self.DV)):
            p.s_mov(self.d_x[0], P(which + "_lo"))
            p.s_mov(self.d_x[1], P(which + "_hi"))
            p.s_mov(self.d_x[2], P(which + "_rng"))
            p.s_mov(self.d_x[3], 0x00020000)
            p.v_mul_lo_u32(self.vo_k[0], t2, P(which + "_sn"))
            p.v_add_u32(self.vo_k[0], self.vo_k[0], t3)
            p.s_lshl_b32(self.s_tmp[1], P(which + "_sn"), 5)
            p.v_add_u32(self.vo_k[1], self.s_tmp[1], self.vo_k[0])
            # column groups k / k+1 exchanged between the half-waves: 16 contiguous bytes per lane, one dwordx4 store
            # per pair (the store tail is issue-bound: T21 of the CDNA guide)
            for kbi in range(2):
                for db in range(self.DB):
                    for gp in range(2):
                        if 32 * db + 16 * gp >= self.D:
                            continue                                        # padding columns of the last block
                        X, Y = self.POOL[(2 * npair) % 8], self.POOL[(2 * npair + 1) % 8]
                        npair += 1
                        for e in range(4):
                            p.v_accvgpr_read(X[e], acc[db][kbi][8 * gp + e])
                            p.v_accvgpr_read(Y[e], acc[db][kbi][8 * gp + 4 + e])
                        if which == "dk":
                            for e in range(4):
                                p.v_mul_f32(X[e], P("scale"), X[e])
                                p.v_mul_f32(Y[e], P("scale"), Y[e])
                        p.v_cvt_pk(dt, X[0], X[0], X[1])
                        p.v_cvt_pk(dt, X[1], X[2], X[3])
                        p.v_cvt_pk(dt, X[2], Y[0], Y[1])
                        p.v_cvt_pk(dt, X[3], Y[2], Y[3])
                        p.v_permlane32_swap(X[0], X[2])
                        p.v_permlane32_swap(X[1], X[3])
                        p.buffer_store(X[0:4], self.vo_k[kbi], self.d_x, 0, offset=64 * db + 32 * gp)
        # ---- split sweeps (sink split / row split, csrc/sfa_bwd_mfma.hip): this chunk's dK / dV in f32, [key][D] rows of
        # 4 D bytes at p?_lo (moved back by the block's first key: the key index is absolute); bwd_part_reduce_kernel adds the
        # chunks up and rounds ONCE.  The accumulator layout gives every lane four consecutive columns per register quad:
        # registers 4 g .. 4 g + 3 of block (db, kbi) = columns 32 db + 8 g + 4 h ... + 3 of the lane's key.
        if not self.partials:
            p.s_waitcnt(vmcnt=0)
            return p
        p.s_cmp("eq_u32", P("p_rng"), 0)
        p.s_cbranch("scc1", "L_nopart%=")
        p.v_mul_u32_u24(self.vo_k[0], 4 * self.D, t2)
        p.v_add_u32(self.vo_k[0], self.vo_k[0], t3)                         # + 16 h bytes (4 f32)
        p.v_add_u32(self.vo_k[1], 32 * 4 * self.D, self.vo_k[0])
        for which, acc in (("pk", self.DKA), ("pv", self.DV)):
            p.s_mov(self.d_x[0], P(which + "_lo"))
            p.s_mov(self.d_x[1], P(which + "_hi"))
            p.s_mov(self.d_x[2], P("p_rng"))
            p.s_mov(self.d_x[3], 0x00020000)
            n = 0
            for kbi in range(2):
                for db in range(self.DB):
                    for g4 in range(4):
                        if 32 * db + 8 * g4 >= self.D:
                            continue                                        # padding columns of the last block
                        X = self.POOL[n % 8]
                        n += 1
                        for e in range(4):
                            p.v_accvgpr_read(X[e], acc[db][kbi][4 * g4 + e])
                        if which == "pk":
                            for e in range(4):
                                p.v_mul_f32(X[e], P("scale"), X[e])
                        p.buffer_store(X[0:4], self.vo_k[kbi], self.d_x, 0, offset=4 * (32 * db + 8 * g4))
        p.label("L_nopart%=")
        p.s_waitcnt(vmcnt=0)
        return p

    def trip_bodies(self):
        """(label or None for the fall-through body behind the loop head, program) of the trip bodies"""
        b = [(None, self.trip_body(False)), ("L_edge%=", self.trip_body(True))]
        if self.half_edges:
            b += [("L_e0%=", self.trip_body(True, live=(0,))), ("L_e1%=", self.trip_body(True, live=(1,)))]
        return b

    # ------------------------------------------------------------------ whole program
    def build(self):
        items = []
        items += finish_block(self.prologue().items)
        items += insert_waits(self.loop_top().items)
        for lbl, prog in self.trip_bodies():
            body = prog.items
            if lbl:
                items.append(Instr("label", mods={"label": lbl}, kind="label", cost=0))
            if self.do_sched:
                body = schedule(body)
            if "waits" not in self.ablate:      # (knock-out build: no LDS waits at all, timing only)
                body = insert_waits(body, strict_tail=False)
            body = fix_hazards(body, loop=True)
            items += body
            items.append(Instr("s_branch", mods={"label": "L_top%="}, kind="branch"))
        if self.dead:
            items.append(Instr("label", mods={"label": "L_dead%="}, kind="label", cost=0))
            items += fix_hazards(insert_waits(self.dead_body().items, strict_tail=False), loop=True)
            items.append(Instr("s_branch", mods={"label": "L_top%="}, kind="branch"))
        items += finish_block(self.out_of_line().items)
        items += finish_block(self.epilogue().items)
        return items

    def clobbers(self):
        c = ["v%d" % i for i in range(self.vfirst, 256)] + ["a%d" % i for i in range(256)]
        c += ["s%d" % i for i in range(self.sfirst, 100)] + ["vcc", "scc", "memory"]
        return c
