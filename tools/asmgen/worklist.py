"""Work-list (persistent workgroup) machinery shared by the forward and the dQ generators.

A workgroup no longer owns ONE (batch, KV head, head set, query tile) but walks a list of such work items; the HIP shell
writes one 128-byte descriptor per item into LDS behind the K / V tile ring and passes their number.  Two cursors run
over the list:

  * the COMPUTE cursor (s_item; per-item scalars s_nt / s_ts_hi / s_tw_off, row positions, Q fragments ...), advanced at
    the end of an item's tile loop;
  * the K / V STREAM cursor (s_ditem, s_dit), three tiles ahead of the compute cursor: every loop iteration fetches "the
    next tile of the stream" into the next ring slot, and when the current item's tiles are exhausted the stream moves on
    to the next item's first tiles.  So the 4-deep ring simply continues across items: when an item's loop ends, the
    first three tiles of the next item are already landed or in flight, and its Q fragments are requested before the
    finished item's epilogue runs.  What used to be ~10 000 cycles of exposed prologue latency per workgroup (two
    dependent memory round trips) is paid once per CU instead of once per item.

Measured and left out (profiles/r03_ab_l2pf.log): pulling the next item's Q / dO rows into the L2 a few iterations early
(one dword per 128-byte line by LDS-DMA into a scratch area) - the fragment loads of an item are bound by the CU's
address / tag path (32 cache lines per instruction), not by where the lines come from: forward unchanged, dQ 0.9 % slower.

Descriptor fields are read with uniform-address ds_read_b128 (every lane the same address: a broadcast) into scratch
VGPRs and moved to SGPRs with v_readfirstlane.
"""
from __future__ import annotations

from .core import P, Prog

DESC_BYTES = 128
DESC_MAX = 256                 # items per asm invocation (32 KB of descriptors: ring + table = the CU's 160 KB)
# K / V stream part of a descriptor (dwords): the same in both kernels
STREAM = {"k_lo": 16, "k_hi": 17, "v_lo": 18, "v_hi": 19, "k_rng": 20, "v_rng": 21, "nt": 22, "ts_hi": 23, "tw_off": 24}


class WorkList:
    """mixin of FwdGen / DqGen (persist=True).  The host class provides: DESC (field -> dword), DESC_BASE, s_tmp, vt,
    POOL, l_dma, l_dma1, d_k, d_v, s_koff, s_voff, s_std, s_wofs, HALVES, dma_t0, dma_dt."""

    def wl_alloc(self, sa):
        self.s_item = sa("s_item")                       # compute cursor
        self.s_nt, self.s_ts_hi, self.s_tw_off = sa("s_nt"), sa("s_ts_hi"), sa("s_tw_off")
        self.s_ditem, self.s_dit = sa("s_ditem"), sa("s_dit")   # stream cursor: item, tile inside the item
        self.s_dnt, self.s_dts_hi, self.s_dtw_off = sa("s_dnt"), sa("s_dts_hi"), sa("s_dtw_off")
        self.s_krng, self.s_vrng = sa("s_krng"), sa("s_vrng")
        self.d_y = sa("d_y", 4, 4)                       # descriptor of the next item's loads (d_x: the finished item's stores)
        self._wl_uid = 0

    def wl_label(self, stem):
        self._wl_uid += 1
        return "L_%s%d_%%=" % (stem, self._wl_uid)

    # ------------------------------------------------------------------ descriptor access
    def desc_read(self, p: Prog, item, groups, land, addr):
        """ds_read_b128 of the 16-byte groups `groups` of item `item`'s descriptor into land[group] (VGPR quads);
        addr: a scratch VGPR"""
        t = self.s_tmp
        p.s_lshl_b32(t[0], item, 7)
        p.s_add_u32(t[0], t[0], self.DESC_BASE)
        p.v_mov(addr, t[0])
        for g in groups:
            p.ds_read_b128(land[g], addr, 16 * g, mem=("desc",), note="descriptor group %d" % g)

    def desc_get(self, p: Prog, dst, land, name):
        d = self.DESC[name]
        p.v_readfirstlane(dst, land[d >> 2][d & 3])

    # ------------------------------------------------------------------ K / V stream
    def emit_stream_fields(self, p: Prog, land):
        for dst, nm in ((self.d_k[0], "k_lo"), (self.d_k[1], "k_hi"), (self.d_v[0], "v_lo"), (self.d_v[1], "v_hi"),
                        (self.s_krng, "k_rng"), (self.s_vrng, "v_rng"), (self.s_dnt, "nt"), (self.s_dts_hi, "ts_hi"),
                        (self.s_dtw_off, "tw_off")):
            self.desc_get(p, dst, land, nm)

    def emit_stream_open(self, p: Prog):
        """stream cursor on the first tile of item 0 (prologue)"""
        land = {4: self.POOL[4], 5: self.POOL[5], 6: self.POOL[6]}
        p.s_mov(self.s_ditem, 0)
        self.desc_read(p, self.s_ditem, (4, 5, 6), land, self.vt[0])
        self.emit_stream_fields(p, land)
        p.s_mov(self.s_dit, 0)
        p.s_mov(self.d_k[3], 0x00020000)
        p.s_mov(self.d_v[3], 0x00020000)

    def emit_stream_advance(self, p: Prog):
        """when the stream item's tiles are used up: next item (or, behind the last one, an empty stream whose fetches go
        through zero-record descriptors).  Inline, two instructions on the common path.  Scratch: POOL[4..6], vt[0]
        (free wherever this is placed: loop head, prologue)."""
        l_na, l_end = self.wl_label("sna"), self.wl_label("send")
        land = {4: self.POOL[4], 5: self.POOL[5], 6: self.POOL[6]}
        p.s_cmp("lt_u32", self.s_dit, self.s_dnt)
        p.s_cbranch("scc1", l_na)
        p.s_add_u32(self.s_ditem, self.s_ditem, 1)
        p.s_cmp("ge_u32", self.s_ditem, P("n_items"))
        p.s_cbranch("scc1", l_end)
        self.desc_read(p, self.s_ditem, (4, 5, 6), land, self.vt[0])
        p.s_waitcnt(lgkmcnt=0)
        self.emit_stream_fields(p, land)
        p.s_mov(self.s_dit, 0)
        p.s_branch(l_na)
        p.label(l_end)
        p.s_mov(self.s_ditem, P("n_items"))
        p.s_mov(self.s_dnt, 0)
        p.s_mov(self.s_dit, 0)
        p.label(l_na)

    def emit_tile_of_pk(self, p: Prog, dst, it, ts_hi, tw_off):
        p.s_add_u32(dst, it, tw_off)
        p.s_cmp("lt_u32", it, ts_hi)
        p.s_cselect(dst, it, dst)

    def emit_dma_stream_tile(self, p: Prog, spread=False):
        """LDS-DMA of the K and V images of the stream's next tile into stage s_std (this wave's pieces), then the cursor
        moves on.  Behind the last item the stream is empty (s_dnt = 0): zero-record descriptors, nothing is read."""
        t = self.s_tmp
        self.emit_tile_of_pk(p, t[0], self.s_dit, self.s_dts_hi, self.s_dtw_off)
        p.s_lshl_b32(t[0], t[0], 6)
        p.s_mul_i32(self.s_koff, t[0], P("k_sn"))
        p.s_mul_i32(self.s_voff, t[0], P("v_sn"))
        p.s_cmp("lt_u32", self.s_dit, self.s_dnt)
        p.s_cselect(self.d_k[2], self.s_krng, 0)
        p.s_cselect(self.d_v[2], self.s_vrng, 0)
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
        p.s_add_u32(self.s_dit, self.s_dit, 1)
