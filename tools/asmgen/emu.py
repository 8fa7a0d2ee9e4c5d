"""CPU emulator of the gfx950 instruction subset the generators use (one workgroup, wave64).

TEST INFRASTRUCTURE: lets the hand-placed kernels be checked against the oracle in this container (no GPU) before
they ever run on hardware, and checks what a passing GPU run cannot show:
  * every register written by an outstanding LDS / global load is waited for (s_waitcnt) before it is touched,
  * LDS bytes written by an LDS-DMA are read only after the issuing wave's covering vmcnt wait and, by other waves,
    after a barrier behind that wait; bytes are not overwritten while another wave may still read them in the same
    barrier interval.
Lane maps (MFMA 32x32x16 operands / results, ds_read_b64_tr_b16, v_permlane32_swap) are the ones tools/probes.hip
verified on MI355X (profiles/r01_probes.log).
"""
from __future__ import annotations

import numpy as np

from .core import Imm, Instr, Reg

U32 = np.uint32
LANES = np.arange(64)


class EmuError(Exception):
    pass


def _f32(u):
    return u.view(np.float32)


def _u32(f):
    return np.asarray(f, dtype=np.float32).view(np.uint32)


def bf16_to_f32(h):
    return (h.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_rne(f):
    u = np.asarray(f, dtype=np.float32).view(np.uint32).astype(np.uint64)
    nan = np.isnan(f)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) & 0xFFFF
    r = np.where(nan, 0x7FC0, r)
    return r.astype(np.uint32)


def f16_to_f32(h):
    return h.astype(np.uint16).view(np.float16).astype(np.float32)


def f32_to_f16_rne(f):
    with np.errstate(over="ignore"):
        return np.asarray(f, dtype=np.float32).astype(np.float16).view(np.uint16).astype(np.uint32)


class Memory:
    """Flat device memory: named buffers at fake 64-bit addresses."""

    def __init__(self):
        self.bufs = []   # (base, np.uint8 array)
        self.next = 0x10000000

    def alloc(self, arr: np.ndarray) -> int:
        raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy()
        base = self.next
        self.bufs.append((base, raw))
        self.next = (base + raw.size + 0xFFFF) & ~0xFFF
        self.next += 0x10000    # guard gap
        return base

    def alloc_zero(self, nbytes: int) -> int:
        return self.alloc(np.zeros(nbytes, dtype=np.uint8))

    def find(self, addr):
        for base, raw in self.bufs:
            if base <= addr < base + raw.size:
                return base, raw
        raise EmuError("access to unmapped address 0x%x" % addr)

    def read(self, base_addr) -> np.ndarray:
        for base, raw in self.bufs:
            if base == base_addr:
                return raw
        raise KeyError(base_addr)

    def load(self, addrs, nbytes):
        """addrs: int64[64]; returns uint8[64, nbytes] (every lane's access must be fully mapped)"""
        out = np.zeros((64, nbytes), dtype=np.uint8)
        for l in range(64):
            a = int(addrs[l])
            if a < 0:
                continue
            base, raw = self.find(a)
            if a + nbytes > base + raw.size:
                raise EmuError("load crosses the end of a buffer: 0x%x + %d" % (a, nbytes))
            out[l] = raw[a - base:a - base + nbytes]
        return out

    def store(self, addrs, data):
        nbytes = data.shape[1]
        for l in range(64):
            a = int(addrs[l])
            if a < 0:
                continue
            base, raw = self.find(a)
            if a + nbytes > base + raw.size:
                raise EmuError("store crosses the end of a buffer: 0x%x + %d" % (a, nbytes))
            raw[a - base:a - base + nbytes] = data[l]


class Wave:
    def __init__(self, wid, nv=256, na=256):
        self.wid = wid
        self.v = np.zeros((nv, 64), dtype=U32)
        self.a = np.zeros((na, 64), dtype=U32)
        self.s = np.zeros(128, dtype=U32)
        self.vcc = np.zeros(64, dtype=bool)
        self.scc = 0
        self.m0 = 0
        self.pc = 0
        self.done = False
        self.at_barrier = False
        # outstanding memory operations, in issue order: ("regs", set_of_regs) or ("dma", seq)
        self.lgkm = []
        self.vm = []
        self.dma_seq = 0            # DMAs issued by this wave
        self.dma_retired = 0        # ... of which this many are known landed (covered by a vmcnt wait)
        self.icount = 0
        self.stats = {}


class Workgroup:
    """Executes a program (list of Instr) on `nwaves` waves sharing one LDS."""

    def __init__(self, prog, nwaves, mem: Memory, params: dict, lds_bytes=160 * 1024, check_races=True):
        self.prog = prog
        self.labels = {it.mods["label"]: i for i, it in enumerate(prog) if it.kind == "label"}
        self.waves = [Wave(w) for w in range(nwaves)]
        self.mem = mem
        self.params = {k: (int(v) & 0xFFFFFFFF) for k, v in params.items()}
        # per-lane inputs: threadIdx.x
        self.vparams = {"tid": [(64 * w + LANES).astype(U32) for w in range(nwaves)]}
        self.lds = np.zeros(lds_bytes, dtype=np.uint8)
        self.check_races = check_races
        self.epoch = 0
        n = lds_bytes
        # race bookkeeping per LDS byte
        self.w_wave = np.full(n, -1, dtype=np.int16)        # last writer
        self.w_seq = np.zeros(n, dtype=np.int32)            # its DMA sequence number (per wave)
        self.w_vis_epoch = np.full(n, -1, dtype=np.int32)   # epoch from which other waves may read (retire epoch + 1); -1: not retired
        self.r_epoch = np.full(n, -1, dtype=np.int32)       # last epoch in which the byte was read
        self.r_wave = np.full(n, -1, dtype=np.int16)        # by whom (-2: several waves)
        self.retire_log = [dict() for _ in range(nwaves)]   # wave -> {seq: epoch of the covering wait}
        self.max_steps = 50_000_000

    # ------------------------------------------------------------ operand access
    def rd_s(self, w: Wave, o):
        """scalar operand -> python int (uint32)"""
        if isinstance(o, Imm):
            return o.bits()
        if o.kind == "s":
            return int(w.s[o.idx])
        if o.kind == "p":
            return self.params[o.idx]
        if o.kind == "m0":
            return w.m0
        if o.kind == "vcc":
            raise EmuError("vcc as scalar operand not modelled")
        raise EmuError("bad scalar operand %r" % (o,))

    def rd_v(self, w: Wave, o):
        """vector operand -> uint32[64]"""
        if isinstance(o, Imm):
            return np.full(64, o.bits(), dtype=U32)
        if o.kind == "v":
            return w.v[o.idx]
        if o.kind == "a":
            return w.a[o.idx]
        if o.kind in ("s", "p", "m0"):
            return np.full(64, self.rd_s(w, o), dtype=U32)
        if o.kind == "pv":
            return self.vparams[o.idx][w.wid]
        raise EmuError("bad vector operand %r" % (o,))

    def rd_vn(self, w: Wave, o: Reg):
        bank = w.v if o.kind == "v" else w.a
        return bank[o.idx:o.idx + o.n]

    def wr_v(self, w: Wave, d: Reg, val):
        bank = w.v if d.kind == "v" else w.a
        bank[d.idx] = np.asarray(val).astype(U32) if np.asarray(val).dtype != U32 else val

    def wr_s(self, w: Wave, d: Reg, val):
        val = int(val) & 0xFFFFFFFF
        if d.kind == "s":
            w.s[d.idx] = val
        elif d.kind == "m0":
            w.m0 = val
        else:
            raise EmuError("bad scalar destination %r" % (d,))

    # ------------------------------------------------------------ waitcnt model
    def _pending_regs(self, w: Wave):
        s = set()
        for kind, x in w.lgkm:
            if kind == "regs":
                s |= x
        for kind, x in w.vm:
            if kind == "regs":
                s |= x
        return s

    def _check_no_pending(self, w: Wave, ins: Instr):
        if not w.lgkm and not w.vm:
            return
        pend = self._pending_regs(w)
        if not pend:
            return
        for r in ins.reads() + ins.writes():
            if r in pend:
                raise EmuError("wave %d pc %d: %s touches %s while a load into it is outstanding (missing s_waitcnt)"
                               % (w.wid, w.pc, ins.text(), r))

    def _retire_vm(self, w: Wave, keep: int):
        while len(w.vm) > keep:
            kind, x = w.vm.pop(0)
            if kind == "dma":
                w.dma_retired = max(w.dma_retired, x)
                self.retire_log[w.wid][x] = self.epoch

    # ------------------------------------------------------------ LDS race checks
    def _lds_read_check(self, w: Wave, idx: np.ndarray, ins):
        if not self.check_races:
            return
        ww = self.w_wave[idx]
        dma = ww >= 0
        if dma.any():
            own = dma & (ww == w.wid)
            if own.any() and (self.w_seq[idx][own] > w.dma_retired).any():
                raise EmuError("wave %d pc %d: %s reads LDS bytes of its own LDS-DMA that no vmcnt wait covers yet"
                               % (w.wid, w.pc, ins.text()))
            other = dma & (ww != w.wid)
            if other.any():
                oi = idx[other]
                # visible iff the writer's covering wait happened in an EARLIER epoch (a barrier lies between)
                for b, wv, sq in zip(oi[:1024:37], self.w_wave[oi][:1024:37], self.w_seq[oi][:1024:37]):
                    ep = self.retire_log[int(wv)].get(int(sq))
                    if ep is None or ep >= self.epoch:
                        raise EmuError("wave %d pc %d epoch %d: %s reads LDS byte %d written by wave %d's LDS-DMA #%d "
                                       "(retired in epoch %s): no vmcnt wait + barrier in between"
                                       % (w.wid, w.pc, self.epoch, ins.text(), int(b), int(wv), int(sq), ep))
        cur = self.r_epoch[idx] == self.epoch
        same = cur & (self.r_wave[idx] != w.wid)
        self.r_wave[idx] = np.where(same, -2, w.wid).astype(np.int16)
        self.r_epoch[idx] = self.epoch

    def _lds_write_check(self, w: Wave, idx: np.ndarray, ins, seq):
        if self.check_races:
            cur = self.r_epoch[idx] == self.epoch
            bad = cur & (self.r_wave[idx] != w.wid)
            if bad.any():
                b = int(idx[bad][0])
                raise EmuError("wave %d pc %d epoch %d: %s overwrites LDS byte %d that wave %d read in the same barrier "
                               "interval" % (w.wid, w.pc, self.epoch, ins.text(), b, int(self.r_wave[b])))
        self.w_wave[idx] = w.wid
        self.w_seq[idx] = seq

    # ------------------------------------------------------------ execution
    def run(self):
        steps = 0
        while True:
            progressed = False
            for w in self.waves:
                if w.done or w.at_barrier:
                    continue
                progressed = True
                while not (w.done or w.at_barrier):
                    self.step(w)
                    steps += 1
                    if steps > self.max_steps:
                        raise EmuError("step limit exceeded (endless loop?)")
            live = [w for w in self.waves if not w.done]
            if not live:
                return
            if all(w.at_barrier for w in live):
                for w in live:
                    w.at_barrier = False
                self.epoch += 1
                continue
            if not progressed:
                raise EmuError("deadlock: some waves ended while others wait at a barrier")

    def step(self, w: Wave):
        if w.pc >= len(self.prog):
            w.done = True
            if w.vm or w.lgkm:
                pass
            return
        ins = self.prog[w.pc]
        w.pc += 1
        k = ins.kind
        if k == "label":
            return
        w.icount += 1
        w.stats[k] = w.stats.get(k, 0) + 1
        if k not in ("wait", "barrier", "nop", "branch"):
            self._check_no_pending(w, ins)
        getattr(self, "x_" + k)(w, ins)

    # ---- kinds
    def x_nop(self, w, ins): pass

    def x_fence(self, w, ins):
        self.x_misc(w, ins)

    def x_misc(self, w, ins):
        if ins.op == "s_memtime":
            w.s[ins.dst[0].idx] = w.icount & 0xFFFFFFFF       # a clock that ticks once per instruction
            w.s[ins.dst[0].idx + 1] = 0

    def x_wait(self, w, ins):
        m = ins.mods
        if "lgkmcnt" in m:
            while len(w.lgkm) > m["lgkmcnt"]:
                w.lgkm.pop(0)
        if "vmcnt" in m:
            self._retire_vm(w, m["vmcnt"])

    def x_barrier(self, w, ins):
        w.at_barrier = True

    def x_branch(self, w, ins):
        op = ins.op
        take = op == "s_branch" or (op == "s_cbranch_scc1" and w.scc) or (op == "s_cbranch_scc0" and not w.scc)
        if take:
            w.pc = self.labels[ins.mods["label"]]

    def _rd64(self, w, o):
        if isinstance(o, Imm):
            return o.bits()
        if o.kind == "vcc":
            return int(sum(1 << i for i in range(64) if w.vcc[i]))
        if o.kind == "s" and o.n == 2:
            return int(w.s[o.idx]) | (int(w.s[o.idx + 1]) << 32)
        raise EmuError("bad 64-bit scalar operand %r" % (o,))

    def _wr64(self, w, d, val):
        if d.kind == "vcc":
            w.vcc = np.array([(val >> i) & 1 for i in range(64)], dtype=bool)
        else:
            w.s[d.idx] = val & 0xFFFFFFFF
            w.s[d.idx + 1] = (val >> 32) & 0xFFFFFFFF

    def x_salu(self, w, ins):
        op = ins.op
        if op == "s_mov_b64":
            self._wr64(w, ins.dst[0], self._rd64(w, ins.src[0])); return
        if op == "s_or_b64":
            r = self._rd64(w, ins.src[0]) | self._rd64(w, ins.src[1]); w.scc = int(r != 0); self._wr64(w, ins.dst[0], r); return
        if op == "s_cmp_lg_u64":
            w.scc = int(self._rd64(w, ins.src[0]) != self._rd64(w, ins.src[1])); return
        s = [self.rd_s(w, o) for o in ins.src]
        M = 0xFFFFFFFF
        sg = lambda x: x - (1 << 32) if x & 0x80000000 else x
        if op == "s_mov_b32":
            self.wr_s(w, ins.dst[0], s[0]); return
        if op.startswith("s_cmp_"):
            cond, ty = op[6:].split("_")
            a, b = (sg(s[0]), sg(s[1])) if ty == "i32" else (s[0], s[1])
            w.scc = int({"lt": a < b, "le": a <= b, "gt": a > b, "ge": a >= b, "eq": a == b, "lg": a != b}[cond]); return
        if op == "s_add_u32":
            r = s[0] + s[1]; w.scc = int(r > M); self.wr_s(w, ins.dst[0], r); return
        if op == "s_addc_u32":
            r = s[0] + s[1] + w.scc; w.scc = int(r > M); self.wr_s(w, ins.dst[0], r); return
        if op == "s_sub_u32":
            r = s[0] - s[1]; w.scc = int(s[1] > s[0]); self.wr_s(w, ins.dst[0], r); return
        if op == "s_subb_u32":
            r = s[0] - s[1] - w.scc; w.scc = int(s[1] + w.scc > s[0]); self.wr_s(w, ins.dst[0], r); return
        if op == "s_add_i32":
            r = sg(s[0]) + sg(s[1]); w.scc = int(not (-(1 << 31) <= r < (1 << 31))); self.wr_s(w, ins.dst[0], r); return
        if op == "s_sub_i32":
            r = sg(s[0]) - sg(s[1]); w.scc = int(not (-(1 << 31) <= r < (1 << 31))); self.wr_s(w, ins.dst[0], r); return
        if op == "s_mul_i32":
            self.wr_s(w, ins.dst[0], (sg(s[0]) * sg(s[1]))); return
        if op == "s_mul_hi_u32":
            self.wr_s(w, ins.dst[0], (s[0] * s[1]) >> 32); return
        if op == "s_lshl_b32":
            r = (s[0] << (s[1] & 31)) & M; w.scc = int(r != 0); self.wr_s(w, ins.dst[0], r); return
        if op == "s_lshr_b32":
            r = s[0] >> (s[1] & 31); w.scc = int(r != 0); self.wr_s(w, ins.dst[0], r); return
        if op == "s_and_b32":
            r = s[0] & s[1]; w.scc = int(r != 0); self.wr_s(w, ins.dst[0], r); return
        if op == "s_or_b32":
            r = s[0] | s[1]; w.scc = int(r != 0); self.wr_s(w, ins.dst[0], r); return
        if op == "s_min_i32":
            a, b = sg(s[0]), sg(s[1]); w.scc = int(a < b); self.wr_s(w, ins.dst[0], min(a, b)); return
        if op == "s_max_i32":
            a, b = sg(s[0]), sg(s[1]); w.scc = int(a > b); self.wr_s(w, ins.dst[0], max(a, b)); return
        if op == "s_cselect_b32":
            self.wr_s(w, ins.dst[0], s[0] if w.scc else s[1]); return
        raise EmuError("salu op not modelled: " + op)

    def x_trans(self, w, ins):
        a = _f32(self.rd_v(w, ins.src[0]))
        with np.errstate(all="ignore"):
            if ins.op == "v_exp_f32":
                r = np.exp2(a.astype(np.float64)).astype(np.float32)
            elif ins.op == "v_log_f32":
                r = np.log2(a.astype(np.float64)).astype(np.float32)
            elif ins.op == "v_rcp_f32":
                r = (1.0 / a.astype(np.float64)).astype(np.float32)
            else:
                raise EmuError("trans op not modelled: " + ins.op)
        self.wr_v(w, ins.dst[0], _u32(r))

    def x_valu(self, w, ins):
        op = ins.op
        if op.startswith("v_cmp_"):
            cond, ty = op[6:].split("_")[:2]
            a, b = self.rd_v(w, ins.src[0]), self.rd_v(w, ins.src[1])
            if ty == "i32":
                a, b = a.view(np.int32), b.view(np.int32)
            elif ty == "f32":
                a, b = _f32(a), _f32(b)
            with np.errstate(invalid="ignore"):
                r = {"lt": a < b, "le": a <= b, "gt": a > b, "ge": a >= b, "eq": a == b, "ne": a != b, "lg": a != b,
                     "neq": ~(a == b)}[cond].copy()
            if ins.dst[0].kind == "vcc":
                w.vcc = r
            else:
                self._wr64(w, ins.dst[0], sum(1 << i for i in range(64) if r[i]))
            return
        if op == "v_cndmask_b32_e64":
            a, b = self.rd_v(w, ins.src[0]), self.rd_v(w, ins.src[1])
            bits = self._rd64(w, ins.src[2])
            msk = np.array([(bits >> i) & 1 for i in range(64)], dtype=bool)
            self.wr_v(w, ins.dst[0], np.where(msk, b, a)); return
        if op == "v_cndmask_b32":
            a, b = self.rd_v(w, ins.src[0]), self.rd_v(w, ins.src[1])
            self.wr_v(w, ins.dst[0], np.where(w.vcc, b, a)); return
        if op == "v_permlane32_swap_b32":
            d0, d1 = ins.dst
            x, y = self.rd_v(w, d0).copy(), self.rd_v(w, d1).copy()
            nx, ny = x.copy(), y.copy()
            nx[32:] = y[:32]
            ny[:32] = x[32:]
            self.wr_v(w, d0, nx); self.wr_v(w, d1, ny); return
        if op == "v_permlane16_swap_b32":
            d0, d1 = ins.dst
            x, y = self.rd_v(w, d0).copy(), self.rd_v(w, d1).copy()
            nx, ny = x.copy(), y.copy()
            nx[16:32] = y[0:16]; ny[0:16] = x[16:32]
            nx[48:64] = y[32:48]; ny[32:48] = x[48:64]
            self.wr_v(w, d0, nx); self.wr_v(w, d1, ny); return
        if op == "v_readfirstlane_b32":
            self.wr_s(w, ins.dst[0], int(self.rd_v(w, ins.src[0])[0])); return
        if op == "v_pk_mul_f32":
            d, a, b = ins.dst[0], ins.src[0], ins.src[1]
            bc = ins.mods.get("pk_bcast")
            res = []
            for h in range(2):
                x = _f32(self.rd_v(w, Reg(a.kind, a.idx + h, 1)))
                y = _f32(self.rd_v(w, Reg(b.kind, b.idx + (0 if bc else h), 1)))
                with np.errstate(all="ignore"):
                    res.append(_u32(x * y))
            for h in range(2):
                self.wr_v(w, Reg(d.kind, d.idx + h, 1), res[h])
            return
        s = [self.rd_v(w, o) for o in ins.src]
        d = ins.dst[0]
        u64 = lambda x: x.astype(np.uint64)
        with np.errstate(all="ignore"):
            if op in ("v_mov_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"):
                r = s[0].copy()
            elif op == "v_add_u32": r = (u64(s[0]) + u64(s[1])).astype(U32)
            elif op == "v_sub_u32": r = (u64(s[0]) - u64(s[1])).astype(U32)
            elif op == "v_subrev_u32": r = (u64(s[1]) - u64(s[0])).astype(U32)
            elif op == "v_add3_u32": r = (u64(s[0]) + u64(s[1]) + u64(s[2])).astype(U32)
            elif op == "v_mul_lo_u32": r = (u64(s[0]) * u64(s[1])).astype(U32)
            elif op == "v_mul_u32_u24": r = ((u64(s[0]) & 0xFFFFFF) * (u64(s[1]) & 0xFFFFFF)).astype(U32)
            elif op == "v_lshlrev_b32": r = (u64(s[1]) << (u64(s[0]) & 31)).astype(U32)
            elif op == "v_lshrrev_b32": r = (s[1] >> (s[0] & 31)).astype(U32)
            elif op == "v_and_b32": r = s[0] & s[1]
            elif op == "v_or_b32": r = s[0] | s[1]
            elif op == "v_xor_b32": r = s[0] ^ s[1]
            elif op == "v_lshl_add_u32": r = ((u64(s[0]) << (u64(s[1]) & 31)) + u64(s[2])).astype(U32)
            elif op == "v_lshl_or_b32": r = ((u64(s[0]) << (u64(s[1]) & 31)).astype(U32)) | s[2]
            elif op == "v_and_or_b32": r = (s[0] & s[1]) | s[2]
            elif op == "v_bfe_u32": r = ((s[0] >> (s[1] & 31)) & ((np.uint64(1) << (u64(s[2]) & 31)) - 1).astype(U32)).astype(U32)
            elif op == "v_fma_f32":
                r = _u32((_f32(s[0]).astype(np.float64) * _f32(s[1]).astype(np.float64) + _f32(s[2]).astype(np.float64)).astype(np.float32))
            elif op == "v_mul_f32": r = _u32(_f32(s[0]) * _f32(s[1]))
            elif op == "v_add_f32": r = _u32(_f32(s[0]) + _f32(s[1]))
            elif op == "v_sub_f32": r = _u32(_f32(s[0]) - _f32(s[1]))
            elif op == "v_max_f32": r = _u32(np.fmax(_f32(s[0]), _f32(s[1])))
            elif op == "v_max3_f32": r = _u32(np.fmax(np.fmax(_f32(s[0]), _f32(s[1])), _f32(s[2])))
            elif op == "v_cvt_pk_bf16_f32":
                r = f32_to_bf16_rne(_f32(s[0])) | (f32_to_bf16_rne(_f32(s[1])) << 16)
            elif op == "v_cvt_pk_f16_f32":
                r = f32_to_f16_rne(_f32(s[0])) | (f32_to_f16_rne(_f32(s[1])) << 16)
            else:
                raise EmuError("valu op not modelled: " + op)
        self.wr_v(w, d, r.astype(U32))

    def x_mfma(self, w, ins):
        d, (a, b, c) = ins.dst[0], ins.src
        is_bf16 = ins.op.endswith("bf16")
        A = self.rd_vn(w, a)   # [4, 64]
        B = self.rd_vn(w, b)
        cvt = bf16_to_f32 if is_bf16 else f16_to_f32
        # element j (0..7) of a lane: register j >> 1, half j & 1
        def unpack(R):
            e = np.zeros((8, 64), dtype=np.float32)
            for j in range(8):
                e[j] = cvt((R[j >> 1] >> (16 * (j & 1))) & 0xFFFF)
            return e
        ea, eb = unpack(A), unpack(B)
        r, h = LANES & 31, LANES >> 5
        Am = np.zeros((32, 16), dtype=np.float64)   # A[i][k]
        Bm = np.zeros((16, 32), dtype=np.float64)   # B[k][j]
        for j in range(8):
            Am[r, 8 * h + j] = ea[j]
            Bm[8 * h + j, r] = eb[j]
        Dm = Am @ Bm
        if isinstance(c, Imm):
            C = np.zeros((16, 64), dtype=np.float32) + np.float32(_f32(np.array([c.bits()], dtype=U32))[0])
        else:
            C = _f32(self.rd_vn(w, c))
        out = np.zeros((16, 64), dtype=np.float32)
        for v in range(16):
            rows = (v & 3) + 8 * (v >> 2) + 4 * h
            out[v] = (Dm[rows, r] + C[v].astype(np.float64)).astype(np.float32)
        bank = w.v if d.kind == "v" else w.a
        bank[d.idx:d.idx + 16] = out.view(U32)

    def _lds_addr(self, w, ins, nbytes):
        addr = self.rd_v(w, ins.src[0]).astype(np.int64) + ins.mods.get("offset", 0)
        if (addr % (8 if nbytes == 8 else (16 if nbytes == 16 else 4)) != 0).any():
            raise EmuError("wave %d pc %d: misaligned LDS access %s" % (w.wid, w.pc, ins.text()))
        if (addr + nbytes > self.lds.size).any() or (addr < 0).any():
            raise EmuError("wave %d pc %d: LDS access out of range %s (max %d)" % (w.wid, w.pc, ins.text(), int(addr.max())))
        return addr

    def x_ds_read(self, w, ins):
        op, d = ins.op, ins.dst[0]
        nbytes = {"ds_read_b128": 16, "ds_read_b64_tr_b16": 8, "ds_read_b32": 4, "ds_read_b64": 8}[op]
        addr = self._lds_addr(w, ins, nbytes)
        idx = (addr[:, None] + np.arange(nbytes)[None, :])          # [64, nbytes]
        self._lds_read_check(w, idx.reshape(-1), ins)
        data = self.lds[idx]                                         # uint8 [64, nbytes]
        words = data.reshape(64, nbytes // 4, 4).astype(np.uint32)
        words = words[:, :, 0] | (words[:, :, 1] << 8) | (words[:, :, 2] << 16) | (words[:, :, 3] << 24)   # [64, n]
        if op == "ds_read_b64_tr_b16":
            # per 16-lane group: lane 4q+p supplies row q, columns 4p..4p+3; lane i receives column i of rows 0..3
            h16 = data.reshape(64, 4, 2).astype(np.uint32)
            h16 = h16[:, :, 0] | (h16[:, :, 1] << 8)                 # [64 lanes, 4 elements]
            out = np.zeros((64, 4), dtype=np.uint32)
            for g in range(4):
                for i in range(16):
                    for q in range(4):
                        src_lane = 16 * g + 4 * q + (i >> 2)
                        out[16 * g + i, q] = h16[src_lane, i & 3]
            words = np.stack([out[:, 0] | (out[:, 1] << 16), out[:, 2] | (out[:, 3] << 16)], axis=1)
        bank = w.v if d.kind == "v" else w.a
        for j in range(d.n):
            bank[d.idx + j] = words[:, j].astype(U32)
        w.lgkm.append(("regs", set(d.regs())))
        if len(w.lgkm) > 15:          # 4-bit counter: the 16th operation issues only once the oldest has completed
            del w.lgkm[:len(w.lgkm) - 15]

    def x_ds_write(self, w, ins):
        nbytes = {"ds_write_b128": 16, "ds_write_b64": 8, "ds_write_b32": 4}[ins.op]
        addr = self._lds_addr(w, ins, nbytes)
        data = self.rd_vn(w, ins.src[1])                              # [n, 64]
        idx = (addr[:, None] + np.arange(nbytes)[None, :])
        self._lds_write_check(w, idx.reshape(-1), ins, 0)
        self.w_wave[idx.reshape(-1)] = -1                             # plain write: ordered by lgkmcnt + barrier, not tracked
        for j in range(nbytes // 4):
            for b in range(4):
                self.lds[addr + 4 * j + b] = ((data[j] >> (8 * b)) & 0xFF).astype(np.uint8)
        w.lgkm.append(("regs", set()))

    def _buffer_addr(self, w, ins, voff_op, rsrc_op, soff_op, nbytes):
        rs = [int(w.s[rsrc_op.idx + i]) if rsrc_op.kind == "s" else None for i in range(4)]
        base = rs[0] | ((rs[1] & 0xFFFF) << 32)
        num_records = rs[2]
        soff = self.rd_s(w, soff_op)
        if soff != 0:
            raise EmuError("soffset != 0 is not modelled (bounds-check semantics differ between parts)")
        voff = self.rd_v(w, voff_op).astype(np.int64) + ins.mods.get("offset", 0)
        inb = voff + nbytes <= num_records
        addr = np.where(inb, base + voff, -1)
        return addr

    def x_dma(self, w, ins):
        nbytes = ins.mods["nbytes"]
        addr = self._buffer_addr(w, ins, ins.src[0], ins.src[1], ins.src[2], nbytes)
        data = self.mem.load(addr, nbytes)                            # OOB lanes: zeros
        lds_base = w.m0 + ins.mods.get("offset", 0)
        if lds_base % 4:
            raise EmuError("LDS-DMA destination not dword aligned")
        dst = lds_base + LANES * nbytes
        if dst.max() + nbytes > self.lds.size:
            raise EmuError("wave %d pc %d: LDS-DMA destination out of range (%d)" % (w.wid, w.pc, int(dst.max())))
        idx = (dst[:, None] + np.arange(nbytes)[None, :]).reshape(-1)
        w.dma_seq += 1
        self._lds_write_check(w, idx, ins, w.dma_seq)
        self.lds[idx] = data.reshape(-1)
        w.vm.append(("dma", w.dma_seq))

    def x_vload(self, w, ins):
        d = ins.dst[0]
        nbytes = 4 * d.n
        addr = self._buffer_addr(w, ins, ins.src[0], ins.src[1], ins.src[2], nbytes)
        data = self.mem.load(addr, nbytes).reshape(64, d.n, 4).astype(np.uint32)
        words = data[:, :, 0] | (data[:, :, 1] << 8) | (data[:, :, 2] << 16) | (data[:, :, 3] << 24)
        bank = w.v if d.kind == "v" else w.a
        for j in range(d.n):
            bank[d.idx + j] = words[:, j].astype(U32)
        w.vm.append(("regs", set(d.regs())))

    def x_vstore(self, w, ins):
        data_op = ins.src[0]
        nbytes = 4 * data_op.n
        addr = self._buffer_addr(w, ins, ins.src[1], ins.src[2], ins.src[3], nbytes)
        regs = self.rd_vn(w, data_op)                                 # [n, 64]
        out = np.zeros((64, nbytes), dtype=np.uint8)
        for j in range(data_op.n):
            for b in range(4):
                out[:, 4 * j + b] = ((regs[j] >> (8 * b)) & 0xFF).astype(np.uint8)
        self.mem.store(addr, out)
        w.vm.append(("regs", set()))
