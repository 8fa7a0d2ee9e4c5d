"""gfx950 instruction objects for the hand-placed kernels (generator side).

A kernel body is a list of `Instr` objects built with the helpers below.  The SAME objects are (a) printed as the
text of one inline-asm statement (emit.py), (b) executed by the CPU emulator (emu.py) and (c) reordered by the gap
scheduler (sched.py), which uses the register read / write sets recorded here.

Registers are physical and named by tuples: ('v', i) arch VGPR, ('a', i) accumulator VGPR, ('s', i) SGPR, ('p', name)
a read-only scalar input of the inline-asm statement (an "s" operand the compiler placed; printed as %[name]), and the
specials ('vcc',), ('scc',), ('m0',).  A register RANGE is (kind, first, count).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple, Union


# ------------------------------------------------------------------ operands
class Reg:
    """A range of `n` consecutive 32-bit registers of one kind."""
    __slots__ = ("kind", "idx", "n")

    def __init__(self, kind: str, idx, n: int = 1):
        self.kind, self.idx, self.n = kind, idx, n

    def __getitem__(self, i):
        if isinstance(i, slice):
            start = i.start or 0
            stop = self.n if i.stop is None else i.stop
            assert 0 <= start < stop <= self.n
            return Reg(self.kind, self.idx + start, stop - start)
        assert 0 <= i < self.n, (i, self.n)
        return Reg(self.kind, self.idx + i, 1)

    def regs(self):
        if self.kind in ("p", "pv", "vcc", "scc", "m0"):
            return [(self.kind, self.idx)]
        return [(self.kind, self.idx + i) for i in range(self.n)]

    def text(self) -> str:
        if self.kind in ("p", "pv"):
            return "%%[%s]" % self.idx
        if self.kind in ("vcc", "m0", "scc"):
            return self.kind
        if self.n == 1:
            return "%s%d" % (self.kind, self.idx)
        return "%s[%d:%d]" % (self.kind, self.idx, self.idx + self.n - 1)

    def __repr__(self):
        return self.text()

    def __eq__(self, o):
        return isinstance(o, Reg) and (self.kind, self.idx, self.n) == (o.kind, o.idx, o.n)

    def __hash__(self):
        return hash((self.kind, self.idx, self.n))


def V(i, n=1): return Reg("v", i, n)
def A(i, n=1): return Reg("a", i, n)
def S(i, n=1): return Reg("s", i, n)
def P(name): return Reg("p", name, 1)      # scalar input operand of the asm statement ("s" constraint)
def PV(name): return Reg("pv", name, 1)    # per-lane input operand ("v" constraint)


VCC = Reg("vcc", 0, 2)
M0 = Reg("m0", 0, 1)
SCC = Reg("scc", 0, 1)


class Imm:
    """Integer immediate (printed in decimal or hex) or float literal (printed as its bit pattern unless inline)."""
    __slots__ = ("val", "is_float")
    INLINE_F = {0.0: "0", 0.5: "0.5", 1.0: "1.0", 2.0: "2.0", 4.0: "4.0", -0.5: "-0.5", -1.0: "-1.0", -2.0: "-2.0", -4.0: "-4.0"}

    def __init__(self, val, is_float=False):
        self.val, self.is_float = val, is_float

    def bits(self) -> int:
        if self.is_float:
            return struct.unpack("<I", struct.pack("<f", self.val))[0]
        return self.val & 0xFFFFFFFF

    def text(self) -> str:
        if self.is_float:
            if self.val in self.INLINE_F:
                return self.INLINE_F[self.val]
            return "0x%08x" % self.bits()
        v = self.val
        if -16 <= v <= 64:
            return str(v)
        return "0x%x" % (v & 0xFFFFFFFF)

    def __repr__(self):
        return self.text()


def imm(v): return Imm(int(v))
def fimm(v): return Imm(float(v), True)


Operand = Union[Reg, Imm]


def _op(x) -> Operand:
    if isinstance(x, (Reg, Imm)):
        return x
    if isinstance(x, int):
        return Imm(x)
    if isinstance(x, float):
        return Imm(x, True)
    raise TypeError(x)


# ------------------------------------------------------------------ instructions
@dataclass
class Instr:
    op: str                                   # mnemonic
    dst: List[Reg] = field(default_factory=list)
    src: List[Operand] = field(default_factory=list)
    mods: dict = field(default_factory=dict)  # offset, lds, offen, sc/nt bits, label, wait counters ...
    kind: str = "valu"                        # mfma valu trans salu ds_read ds_write dma vload vstore wait barrier branch label nop misc
    cost: int = 4                             # issue cycles (scheduler's model)
    note: str = ""
    tag: object = None                        # free for the kernel generators (grouping, priorities)
    # extra dependences that registers do not show (LDS regions read / written), as hashable tokens
    mem_r: Tuple = ()
    mem_w: Tuple = ()

    def reads(self):
        out = []
        for s in self.src:
            if isinstance(s, Reg):
                out += s.regs()
        out += [(r,) if isinstance(r, str) else r for r in self.mods.get("implicit_r", [])]
        return out

    def writes(self):
        out = []
        for d in self.dst:
            out += d.regs()
        out += [(r,) if isinstance(r, str) else r for r in self.mods.get("implicit_w", [])]
        return out

    # ---- text
    def text(self) -> str:
        k, m = self.kind, self.mods
        if k == "label":
            return "%s:" % m["label"]
        if k == "branch":
            return "%s %s" % (self.op, m["label"])
        if k == "wait":
            parts = []
            if "vmcnt" in m: parts.append("vmcnt(%d)" % m["vmcnt"])
            if "lgkmcnt" in m: parts.append("lgkmcnt(%d)" % m["lgkmcnt"])
            return "s_waitcnt " + " ".join(parts)
        if k == "barrier":
            return "s_barrier"
        if k == "nop":
            return "s_nop %d" % m["n"]
        if self.op == "s_setprio":
            return "s_setprio %d" % m["n"]
        if self.op == "s_memtime":
            return "s_memtime %s" % self.dst[0].text()
        if self.op in ("v_permlane32_swap_b32", "v_permlane16_swap_b32"):
            return "%s %s, %s" % (self.op, self.dst[0].text(), self.dst[1].text())
        if k in ("ds_read", "ds_write"):
            ops = [d.text() for d in self.dst] + [s.text() for s in self.src]
            t = "%s %s" % (self.op, ", ".join(ops))
            if m.get("offset"):
                t += " offset:%d" % m["offset"]
            return t
        if k in ("dma", "vload", "vstore"):
            # buffer_<op> vdata, voffset, srsrc, soffset offen [offset:N] [lds] [nt]
            if k == "dma":
                ops = [self.src[0].text(), self.src[1].text(), self.src[2].text()]
            elif k == "vload":
                ops = [self.dst[0].text(), self.src[0].text(), self.src[1].text(), self.src[2].text()]
            else:
                ops = [s.text() for s in self.src]
            t = "%s %s offen" % (self.op, ", ".join(ops))
            if m.get("offset"):
                t += " offset:%d" % m["offset"]
            if m.get("sc0"): t += " sc0"
            if m.get("sc1"): t += " sc1"
            if m.get("nt"): t += " nt"
            if k == "dma":
                t += " lds"
            return t
        if self.op == "v_pk_mul_f32":
            b = self.src[1]
            bt = "%s[%d:%d]" % (b.kind, b.idx, b.idx + 1) if m.get("pk_bcast") else b.text()
            return "v_pk_mul_f32 %s, %s, %s%s" % (self.dst[0].text(), self.src[0].text(), bt, " op_sel_hi:[1,0]" if m.get("pk_bcast") else "")
        ops = [d.text() for d in self.dst if d.kind not in ("scc",) and not (d.kind == "vcc" and m.get("implicit_vcc"))]
        ops += [s.text() for s in self.src if not (isinstance(s, Reg) and s.kind == "scc")]
        t = self.op + (" " + ", ".join(ops) if ops else "")
        return t


class Prog:
    """An instruction list under construction."""

    def __init__(self):
        self.items: List[Instr] = []

    def add(self, ins: Instr) -> Instr:
        self.items.append(ins)
        return ins

    def extend(self, items):
        for i in items:
            self.add(i)

    # ---------------- scalar
    def s_mov(self, d, a, note=""):
        return self.add(Instr("s_mov_b32", [d], [_op(a)], kind="salu", note=note))

    def _salu2(self, op, d, a, b, scc=True, note="", reads_scc=False):
        src = [_op(a), _op(b)]
        mods = {}
        if scc: mods["implicit_w"] = [("scc", 0)]
        if reads_scc: mods["implicit_r"] = [("scc", 0)]
        return self.add(Instr(op, [d], src, mods=mods, kind="salu", note=note))

    def s_add_u32(self, d, a, b, note=""): return self._salu2("s_add_u32", d, a, b, note=note)
    def s_addc_u32(self, d, a, b, note=""): return self._salu2("s_addc_u32", d, a, b, note=note, reads_scc=True)
    def s_sub_u32(self, d, a, b, note=""): return self._salu2("s_sub_u32", d, a, b, note=note)
    def s_subb_u32(self, d, a, b, note=""): return self._salu2("s_subb_u32", d, a, b, note=note, reads_scc=True)
    def s_add_i32(self, d, a, b, note=""): return self._salu2("s_add_i32", d, a, b, note=note)
    def s_sub_i32(self, d, a, b, note=""): return self._salu2("s_sub_i32", d, a, b, note=note)
    def s_mul_i32(self, d, a, b, note=""): return self._salu2("s_mul_i32", d, a, b, scc=False, note=note)
    def s_mul_hi_u32(self, d, a, b, note=""): return self._salu2("s_mul_hi_u32", d, a, b, scc=False, note=note)
    def s_lshl_b32(self, d, a, b, note=""): return self._salu2("s_lshl_b32", d, a, b, note=note)
    def s_lshr_b32(self, d, a, b, note=""): return self._salu2("s_lshr_b32", d, a, b, note=note)
    def s_and_b32(self, d, a, b, note=""): return self._salu2("s_and_b32", d, a, b, note=note)
    def s_or_b32(self, d, a, b, note=""): return self._salu2("s_or_b32", d, a, b, note=note)
    def s_min_i32(self, d, a, b, note=""): return self._salu2("s_min_i32", d, a, b, note=note)
    def s_max_i32(self, d, a, b, note=""): return self._salu2("s_max_i32", d, a, b, note=note)
    def s_cselect(self, d, a, b, note=""): return self._salu2("s_cselect_b32", d, a, b, scc=False, note=note, reads_scc=True)

    # 64-bit lane masks: VCC <-> an SGPR pair
    def s_mov_b64(self, d, a, note=""):
        assert d.n == 2
        return self.add(Instr("s_mov_b64", [d], [_op(a)], kind="salu", note=note))

    def s_or_b64(self, d, a, b, note=""):
        assert d.n == 2
        return self.add(Instr("s_or_b64", [d], [_op(a), _op(b)], mods={"implicit_w": [("scc", 0)]}, kind="salu", note=note))

    def s_cmp_lg_u64(self, a, b, note=""):
        return self.add(Instr("s_cmp_lg_u64", [], [_op(a), _op(b)], mods={"implicit_w": [("scc", 0)]}, kind="salu", note=note))

    def s_cmp(self, cond, a, b, note=""):
        """cond in lt_i32 le_i32 gt_i32 ge_i32 eq_i32 lg_i32 lt_u32 le_u32 gt_u32 ge_u32 eq_u32 lg_u32"""
        return self.add(Instr("s_cmp_" + cond, [], [_op(a), _op(b)], mods={"implicit_w": [("scc", 0)]}, kind="salu", note=note))

    def s_cbranch(self, which, label, note=""):
        """which: scc0 | scc1"""
        return self.add(Instr("s_cbranch_" + which, [], [], mods={"label": label, "implicit_r": [("scc", 0)]}, kind="branch", note=note))

    def s_branch(self, label, note=""):
        return self.add(Instr("s_branch", [], [], mods={"label": label}, kind="branch", note=note))

    def label(self, name):
        return self.add(Instr("label", mods={"label": name}, kind="label", cost=0))

    def s_waitcnt(self, vmcnt=None, lgkmcnt=None, note=""):
        m = {}
        if vmcnt is not None: m["vmcnt"] = vmcnt
        if lgkmcnt is not None: m["lgkmcnt"] = lgkmcnt
        return self.add(Instr("s_waitcnt", mods=m, kind="wait", note=note))

    def s_barrier(self, note=""):
        return self.add(Instr("s_barrier", kind="barrier", note=note))

    def s_nop(self, n, note=""):
        return self.add(Instr("s_nop", mods={"n": n}, kind="nop", cost=4 * (n + 1), note=note))

    def s_setprio(self, n):
        return self.add(Instr("s_setprio", mods={"n": n}, kind="misc"))

    # ---------------- vector ALU
    def _valu(self, op, d, srcs, note="", kind="valu", cost=4, mods=None):
        return self.add(Instr(op, [d] if d is not None else [], [_op(s) for s in srcs], mods=mods or {}, kind=kind, cost=cost, note=note))

    def v_mov(self, d, a, note=""): return self._valu("v_mov_b32", d, [a], note)
    def v_add_u32(self, d, a, b, note=""): return self._valu("v_add_u32", d, [a, b], note)
    def v_sub_u32(self, d, a, b, note=""): return self._valu("v_sub_u32", d, [a, b], note)          # d = a - b
    def v_subrev_u32(self, d, a, b, note=""): return self._valu("v_subrev_u32", d, [a, b], note)    # d = b - a
    def v_mul_lo_u32(self, d, a, b, note=""): return self._valu("v_mul_lo_u32", d, [a, b], note, cost=16)
    def v_mul_u32_u24(self, d, a, b, note=""): return self._valu("v_mul_u32_u24", d, [a, b], note)
    def v_lshlrev(self, d, sh, a, note=""): return self._valu("v_lshlrev_b32", d, [sh, a], note)     # d = a << sh
    def v_lshrrev(self, d, sh, a, note=""): return self._valu("v_lshrrev_b32", d, [sh, a], note)     # d = a >> sh
    def v_and(self, d, a, b, note=""): return self._valu("v_and_b32", d, [a, b], note)
    def v_or(self, d, a, b, note=""): return self._valu("v_or_b32", d, [a, b], note)
    def v_xor(self, d, a, b, note=""): return self._valu("v_xor_b32", d, [a, b], note)
    def v_lshl_add_u32(self, d, a, sh, c, note=""): return self._valu("v_lshl_add_u32", d, [a, sh, c], note)   # (a << sh) + c
    def v_lshl_or_b32(self, d, a, sh, c, note=""): return self._valu("v_lshl_or_b32", d, [a, sh, c], note)     # (a << sh) | c
    def v_and_or_b32(self, d, a, b, c, note=""): return self._valu("v_and_or_b32", d, [a, b, c], note)         # (a & b) | c
    def v_add3_u32(self, d, a, b, c, note=""): return self._valu("v_add3_u32", d, [a, b, c], note)
    def v_bfe_u32(self, d, a, off, w, note=""): return self._valu("v_bfe_u32", d, [a, off, w], note)          # (a >> off) & ((1<<w)-1)

    def v_fma_f32(self, d, a, b, c, note=""): return self._valu("v_fma_f32", d, [a, b, c], note)
    def v_mul_f32(self, d, a, b, note=""): return self._valu("v_mul_f32", d, [a, b], note)
    def v_add_f32(self, d, a, b, note=""): return self._valu("v_add_f32", d, [a, b], note)

    def v_pk_mul_f32(self, d, a, b, bcast_b=False, note=""):
        """two f32 products per lane in one issue slot: d[0:1] = a[0:1] * b[0:1]; bcast_b: b is ONE register whose value
        multiplies both halves (op_sel_hi:[1,0]; the register pair named in the text starts at b, its second register is
        not read).  Register pairs are even-aligned (gfx90a+)."""
        assert d.n == 2 and a.n == 2 and d.idx % 2 == 0 and a.idx % 2 == 0
        assert (b.n == 1 and b.idx % 2 == 0) if bcast_b else (b.n == 2 and b.idx % 2 == 0)
        return self.add(Instr("v_pk_mul_f32", [d], [a, b], mods={"pk_bcast": bool(bcast_b)}, kind="valu", cost=4, note=note))
    def v_sub_f32(self, d, a, b, note=""): return self._valu("v_sub_f32", d, [a, b], note)
    def v_max_f32(self, d, a, b, note=""): return self._valu("v_max_f32", d, [a, b], note)
    def v_max3_f32(self, d, a, b, c, note=""): return self._valu("v_max3_f32", d, [a, b, c], note)
    def v_exp_f32(self, d, a, note=""): return self._valu("v_exp_f32", d, [a], note, kind="trans", cost=8)
    def v_log_f32(self, d, a, note=""): return self._valu("v_log_f32", d, [a], note, kind="trans", cost=8)
    def v_rcp_f32(self, d, a, note=""): return self._valu("v_rcp_f32", d, [a], note, kind="trans", cost=8)
    def v_cvt_pk(self, dtype, d, lo, hi, note=""):
        """d = pack(cvt(lo), cvt(hi)) to 16 bit, round to nearest even"""
        op = "v_cvt_pk_bf16_f32" if dtype == "bf16" else "v_cvt_pk_f16_f32"
        return self._valu(op, d, [lo, hi], note)

    def v_cmp(self, cond, a, b, note=""):
        """writes VCC; cond like lt_u32, gt_i32, ... (VOPC: compares a <cond> b)"""
        return self.add(Instr("v_cmp_" + cond, [VCC], [_op(a), _op(b)], mods={"implicit_vcc": False}, kind="valu", note=note))

    def v_cndmask(self, d, a, b, note=""):
        """d = vcc ? b : a"""
        return self.add(Instr("v_cndmask_b32", [d], [_op(a), _op(b), VCC], kind="valu", note=note))

    def v_cmp_m(self, cond, m, a, b, note=""):
        """lane mask of (a <cond> b) into the SGPR pair m (VOP3 form: VCC stays free, several masks can be in flight)"""
        assert m.n == 2
        return self.add(Instr("v_cmp_%s_e64" % cond, [m], [_op(a), _op(b)], kind="valu", note=note))

    def v_cndmask_m(self, d, a, b, m, note=""):
        """d = m ? b : a, m an SGPR pair"""
        assert m.n == 2
        return self.add(Instr("v_cndmask_b32_e64", [d], [_op(a), _op(b), m], kind="valu", note=note))

    def v_accvgpr_write(self, d, a, note=""): return self._valu("v_accvgpr_write_b32", d, [a], note)
    def v_accvgpr_read(self, d, a, note=""): return self._valu("v_accvgpr_read_b32", d, [a], note)

    def v_permlane32_swap(self, a, b, note=""):
        """lanes 32..63 of a swap with lanes 0..31 of b"""
        return self.add(Instr("v_permlane32_swap_b32", [a, b], [a, b], kind="valu", note=note))

    def v_permlane16_swap(self, a, b, note=""):
        """16-lane rows 1 and 3 of a swap with rows 0 and 2 of b (lanes 16..31 <-> 0..15, 48..63 <-> 32..47)"""
        return self.add(Instr("v_permlane16_swap_b32", [a, b], [a, b], kind="valu", note=note))

    def v_readfirstlane(self, d, a, note=""):
        return self.add(Instr("v_readfirstlane_b32", [d], [a], kind="valu", note=note))

    # ---------------- matrix
    def mfma(self, dtype, d, a, b, c, note="", tag=None):
        """d[16] = A(4 regs) x B(4 regs) + c (16 regs or the constant 0), v_mfma_f32_32x32x16_{bf16,f16}"""
        op = "v_mfma_f32_32x32x16_" + ("bf16" if dtype == "bf16" else "f16")
        assert d.n == 16 and a.n == 4 and b.n == 4
        return self.add(Instr(op, [d], [a, b, _op(c)], kind="mfma", cost=8, note=note, tag=tag))

    # ---------------- LDS
    def ds_read_b128(self, d, addr, offset=0, mem=(), note=""):
        assert d.n == 4 and 0 <= offset < 65536
        return self.add(Instr("ds_read_b128", [d], [addr], mods={"offset": offset}, kind="ds_read", cost=4, note=note, mem_r=tuple(mem)))

    def ds_read_b64_tr_b16(self, d, addr, offset=0, mem=(), note=""):
        assert d.n == 2 and 0 <= offset < 65536
        return self.add(Instr("ds_read_b64_tr_b16", [d], [addr], mods={"offset": offset}, kind="ds_read", cost=4, note=note, mem_r=tuple(mem)))

    def ds_read_b32(self, d, addr, offset=0, mem=(), note=""):
        return self.add(Instr("ds_read_b32", [d], [addr], mods={"offset": offset}, kind="ds_read", cost=4, note=note, mem_r=tuple(mem)))

    def ds_write_b128(self, addr, data, offset=0, mem=(), note=""):
        assert data.n == 4
        return self.add(Instr("ds_write_b128", [], [addr, data], mods={"offset": offset}, kind="ds_write", cost=16, note=note, mem_w=tuple(mem)))

    def ds_write_b64(self, addr, data, offset=0, mem=(), note=""):
        assert data.n == 2
        return self.add(Instr("ds_write_b64", [], [addr, data], mods={"offset": offset}, kind="ds_write", cost=8, note=note, mem_w=tuple(mem)))

    # ---------------- buffer memory
    def buffer_load_lds(self, nbytes, voff, rsrc, soff, offset=0, mem=(), note="", nt=False):
        """LDS-DMA: LDS[M0 + inst_offset? (no: M0 only) + lane*nbytes] = mem[rsrc.base + voff + soff + offset]; M0 implicit"""
        assert rsrc.n == 4 and nbytes in (4, 16) and 0 <= offset < 4096
        op = "buffer_load_dwordx4" if nbytes == 16 else "buffer_load_dword"
        return self.add(Instr(op, [], [voff, rsrc, _op(soff)], mods={"offset": offset, "nbytes": nbytes, "nt": nt, "implicit_r": [("m0", 0)]},
                              kind="dma", cost=60, note=note, mem_w=tuple(mem)))

    def buffer_load(self, d, voff, rsrc, soff, offset=0, note=""):
        assert rsrc.n == 4 and 0 <= offset < 4096
        op = {1: "buffer_load_dword", 2: "buffer_load_dwordx2", 4: "buffer_load_dwordx4"}[d.n]
        return self.add(Instr(op, [d], [voff, rsrc, _op(soff)], mods={"offset": offset}, kind="vload", cost=16, note=note))

    def buffer_store(self, data, voff, rsrc, soff, offset=0, note=""):
        assert rsrc.n == 4 and 0 <= offset < 4096
        op = {1: "buffer_store_dword", 2: "buffer_store_dwordx2", 4: "buffer_store_dwordx4"}[data.n]
        return self.add(Instr(op, [], [data, voff, rsrc, _op(soff)], mods={"offset": offset}, kind="vstore", cost=16, note=note))

    def s_mov_m0(self, a, note=""):
        return self.add(Instr("s_mov_b32", [M0], [_op(a)], kind="salu", note=note))

    def s_add_m0(self, a, b, note=""):
        return self.add(Instr("s_add_u32", [M0], [_op(a), _op(b)], mods={"implicit_w": [("scc", 0)]}, kind="salu", note=note))


def emit_text(items: Sequence[Instr], comments: bool = True) -> str:
    """Text of the instruction list, one per line (for the inline asm string)."""
    lines = []
    for it in items:
        t = it.text()
        if comments and it.note:
            t += "   ; " + it.note
        lines.append(t)
    return "\n".join(lines)
