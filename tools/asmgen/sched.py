"""Placement of a straight-line instruction list around its MFMA spine, plus the two passes every emitted block
goes through: counted s_waitcnt insertion for LDS / global loads into registers, and gfx950 wait-state padding.

schedule(items): the MFMAs keep their program order and pace the block (one per 32 cycles on a SIMD); every other
instruction is a "filler" placed into the gaps between them, earliest program order first, as soon as the
instructions it depends on have been placed (register RAW / WAR / WAW and declared LDS-region dependences are taken
from the instruction objects, so any order the scheduler produces computes what the program order computes).
Fences (barriers, hand-written waits, M0 / SCC / VCC chains through registers) are respected the same way.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

from .core import Imm, Instr, Prog, Reg

MFMA_PIPE = 32      # cycles of the matrix pipe per v_mfma_f32_32x32x16
MFMA_ISSUE = 8      # cycles the MFMA holds the wave's issue
LDS_LAT = 200       # issue -> data under load, modelled (a wait is inserted wherever the consumer really sits)
MFMA_LAT = 64       # issue -> result readable by a non-MFMA instruction, modelled
VALU_LAT = 8
SATURATION_WAITS = False
LDS_CYCLES_PER_SLOT = 8  # LDS array cycles one wave may use per MFMA slot (b128 read: 4, b64 read: 2); 4 waves share 32
MAX_LDS_INFLIGHT = 12   # LDS reads a wave keeps in flight (the hardware counter saturates at 15)


def _is_fence(it: Instr) -> bool:
    return it.kind in ("barrier", "label", "branch", "fence") or (it.kind == "wait")


def build_deps(items: Sequence[Instr]) -> List[List[int]]:
    """deps[i] = indices j < i that must be placed before i."""
    n = len(items)
    deps: List[set] = [set() for _ in range(n)]
    last_w: Dict[object, int] = {}
    readers: Dict[object, List[int]] = {}
    last_fence = -1
    since_fence: List[int] = []
    last_mfma = -1
    for i, it in enumerate(items):
        if last_fence >= 0:
            deps[i].add(last_fence)
        rd = list(it.reads()) + [("mem", t) for t in it.mem_r]
        wr = list(it.writes()) + [("mem", t) for t in it.mem_w]
        for r in rd:
            if r in last_w:
                deps[i].add(last_w[r])
        for r in wr:
            if r in last_w:
                deps[i].add(last_w[r])
            for j in readers.get(r, ()):
                if j != i:
                    deps[i].add(j)
        if it.kind == "mfma":
            if last_mfma >= 0:
                deps[i].add(last_mfma)          # the spine keeps its order
            last_mfma = i
        if _is_fence(it):
            for j in since_fence:
                deps[i].add(j)
            last_fence = i
            since_fence = []
        else:
            since_fence.append(i)
        for r in rd:
            readers.setdefault(r, []).append(i)
        for r in wr:
            last_w[r] = i
            readers[r] = []
        deps[i].discard(i)
    return [sorted(d) for d in deps]


def _latency(pj: Instr, it: Instr, true_dep: bool) -> float:
    """cycles from the issue of pj to the earliest issue of the dependent `it`"""
    if not true_dep:
        return 0.0 if pj.kind != "mfma" else 4.0
    if pj.kind == "ds_read":
        return LDS_LAT
    if pj.kind == "mfma":
        return MFMA_PIPE if it.kind == "mfma" else MFMA_LAT
    if pj.kind == "trans":
        return 16.0
    if pj.kind == "vload":
        return 800.0
    if pj.kind in ("salu",):
        return 4.0 if it.kind != "dma" else 8.0
    return VALU_LAT


def schedule(items: Sequence[Instr], verbose: bool = False, trace=None) -> List[Instr]:
    """Reorder `items` (no labels / branches inside, fences allowed) around the MFMA spine.

    List scheduling on a cycle model (MFMA: 8 cycles of issue, 32 of matrix pipe; fillers: their issue cost) with
    as-late-as-possible deadlines as priorities: the k-th MFMA should start at 32 k; every other instruction gets the
    latest issue time that still lets everything depending on it meet that (LDS latency 128, MFMA result 64, ...).
    In every gap the ready instruction with the earliest deadline goes first; one that does not fit into the gap is
    placed anyway when the next MFMA's ideal start is past its deadline (it would stall that or a later MFMA),
    otherwise the MFMA goes.  `mods["alap"]` overrides a deadline (LDS-DMA: early in the block, see the kernels);
    `mods["after_mfma"] = k` keeps an instruction behind the k-th MFMA of the spine (the strip kernels spread their global
    requests over an item's block this way, with `alap` = 32 (k + 1) so that it goes out right there: the CU's address path
    takes ~56 cycles per fragment load and blocks the in-order wave when requests come in a burst)."""
    items = list(items)
    n = len(items)
    deps = build_deps(items)
    users: List[List[int]] = [[] for _ in range(n)]
    for i, d in enumerate(deps):
        for j in d:
            users[j].append(i)
    spine = [i for i in range(n) if items[i].kind == "mfma"]
    rank = {i: k for k, i in enumerate(spine)}
    t_end = MFMA_PIPE * len(spine)
    wsets = [set(it.writes()) for it in items]
    rsets = [set(it.reads()) for it in items]

    def lat(j: int, i: int) -> float:
        return _latency(items[j], items[i], bool(wsets[j] & rsets[i]))

    # ---- deadlines (backward over program order: every user has a larger index)
    alap = [0.0] * n
    for i in range(n - 1, -1, -1):
        it = items[i]
        cost = MFMA_ISSUE if it.kind == "mfma" else it.cost
        a = float(t_end) - cost
        for u in users[i]:
            a = min(a, alap[u] - max(lat(i, u), cost))
        if it.kind == "mfma":
            a = min(a, float(MFMA_PIPE * rank[i]))
        if "alap" in it.mods:
            a = min(a, float(it.mods["alap"]))
        alap[i] = a

    placed_at = [-1.0] * n
    ndeps = [len(d) for d in deps]
    ready = set(i for i in range(n) if ndeps[i] == 0)
    out: List[int] = []
    t = 0.0
    pipe_free = 0.0
    next_spine = 0
    lds_inflight: List[float] = []      # modelled completion times of the LDS reads in flight (lgkmcnt counts to 15)

    def lds_room(at: float) -> bool:
        while lds_inflight and lds_inflight[0] <= at:
            lds_inflight.pop(0)
        return len(lds_inflight) < MAX_LDS_INFLIGHT

    # LDS bandwidth share of ONE wave: the four waves of the workgroup run the same schedule in step and share one LDS
    # (256 B per clock): 2 ds_read_b128 (or 4 ds_read_b64) per wave and MFMA slot saturate it; bursts beyond that stall
    # the issue of all four waves.  Token bucket: LDS_CYCLES_PER_SLOT array cycles per 32 clocks, bucket of one slot.
    bucket = [float(LDS_CYCLES_PER_SLOT), 0.0]           # tokens, time of the last update

    def lds_cost(i: int) -> float:
        return 4.0 if items[i].op == "ds_read_b128" else 2.0

    def lds_tokens(at: float) -> float:
        return min(float(LDS_CYCLES_PER_SLOT), bucket[0] + (at - bucket[1]) * LDS_CYCLES_PER_SLOT / MFMA_PIPE)

    def lds_take(i: int, at: float):
        bucket[0] = lds_tokens(at) - lds_cost(i)
        bucket[1] = at

    def data_ready(i: int) -> float:
        tt = 0.0
        for j in deps[i]:
            tt = max(tt, placed_at[j] + lat(j, i))
        return tt

    def place(i: int, at: float):
        nonlocal t, pipe_free, next_spine
        it = items[i]
        placed_at[i] = at
        if it.kind == "mfma":
            pipe_free = at + MFMA_PIPE
            t = at + MFMA_ISSUE
            next_spine += 1
        else:
            t = at + it.cost
            if it.kind == "ds_read":
                lds_inflight.append(at + LDS_LAT)
                lds_take(i, at)
        out.append(i)
        ready.discard(i)
        for u in users[i]:
            ndeps[u] -= 1
            if ndeps[u] == 0:
                ready.add(u)
        if trace is not None:
            trace.append((at, it, alap[i]))

    while len(out) < n:
        mf = spine[next_spine] if next_spine < len(spine) and spine[next_spine] in ready else None
        fillers = [i for i in ready if items[i].kind != "mfma"]
        if mf is not None:
            start = max(pipe_free, data_ready(mf), t)
            ideal_next = float(MFMA_PIPE * next_spine)
            # slip of the real timeline against the ideal one: deadlines move with it
            slip = max(0.0, start - ideal_next)
            best = None
            for i in fillers:
                if items[i].mods.get("after_mfma", 0) > next_spine:
                    continue                      # "not before the k-th MFMA": spreads global requests over a long block
                dr = max(data_ready(i), t)
                if items[i].kind == "ds_read" and not lds_room(dr):
                    continue                      # the LDS queue is full: reads wait for a later gap
                key = (alap[i], i)
                fits = dr + items[i].cost <= start + 0.5
                urgent = alap[i] + slip < start + MFMA_ISSUE      # waiting for the MFMA would make it late
                if items[i].kind == "ds_read" and lds_tokens(dr) < lds_cost(i) and not urgent:
                    continue                      # this wave's share of the LDS is used up for now
                if (fits or urgent) and (best is None or key < best[0]):
                    best = (key, i, dr)
            if best is not None:
                place(best[1], best[2])
                continue
            place(mf, start)
            continue
        if not fillers:
            raise RuntimeError("scheduler stuck: dependence cycle?")
        open_f = [i for i in fillers if items[i].mods.get("after_mfma", 0) <= next_spine]
        if open_f:
            fillers = open_f                      # (gated fillers go only when nothing else can: the MFMA they wait behind needs them)
        # the next MFMA waits for fillers: most urgent first
        cand = sorted((alap[i], i) for i in fillers)
        i = cand[0][1]
        at = max(data_ready(i), t) if items[i].kind not in ("wait", "barrier") else t
        if items[i].kind == "ds_read" and not lds_room(at):
            at = max(at, lds_inflight[0])
            lds_room(at)
        place(i, at)
    if verbose:
        print("schedule: %d instructions, %d MFMA, modelled %.0f cycles (MFMA floor %d)"
              % (n, len(spine), max(t, pipe_free), t_end))
    return [items[i] for i in out]


# ------------------------------------------------------------------ waits
def insert_waits(items: Sequence[Instr], strict: bool = False, strict_tail: bool = False) -> List[Instr]:
    """Insert counted s_waitcnt lgkmcnt(N) / vmcnt(N) in front of every instruction that touches a register with an
    outstanding ds_read / buffer_load into it.  LDS ops complete in order, so do vector memory ops (loads, stores
    and LDS-DMA share the vmcnt queue).  Hand-written waits in `items` are honoured (they shorten the queues).
    Labels and branches: the bookkeeping restarts empty there (`strict`: register loads outstanding there are an error;
    `strict_tail`: a global load into a register still outstanding at the END of the list is an error)."""
    out: List[Instr] = []
    lgkm: List[set] = []      # outstanding LDS ops, oldest first: set of destination registers (may be empty)
    vm: List[set] = []
    for it in items:
        k = it.kind
        if k in ("label", "branch"):
            # control flow joins / leaves here: what is outstanding is the hand-written waits' business (the emulator
            # checks them); the automatic bookkeeping restarts empty
            if strict and (any(lgkm) or any(vm)):
                raise RuntimeError("register load outstanding across %s: wait before it" % it.text())
            lgkm, vm = [], []
            out.append(it)
            continue
        if k == "wait":
            m = it.mods
            if "lgkmcnt" in m:
                del lgkm[:max(0, len(lgkm) - m["lgkmcnt"])]
            if "vmcnt" in m:
                del vm[:max(0, len(vm) - m["vmcnt"])]
            out.append(it)
            continue
        touched = set(it.reads()) | set(it.writes())
        need_l = need_v = None
        for pos, regs in enumerate(lgkm):
            if regs & touched:
                need_l = len(lgkm) - 1 - pos     # ops younger than this one may stay outstanding
        for pos, regs in enumerate(vm):
            if regs & touched:
                need_v = len(vm) - 1 - pos
        if need_l is not None or need_v is not None:
            w = Instr("s_waitcnt", kind="wait", mods={})
            if need_l is not None:
                w.mods["lgkmcnt"] = min(need_l, 15)
                del lgkm[:len(lgkm) - min(need_l, 15)]
            if need_v is not None:
                w.mods["vmcnt"] = min(need_v, 63)
                del vm[:len(vm) - min(need_v, 63)]
            out.append(w)
        if k == "ds_read":
            lgkm.append(set(it.writes()))
        elif k == "ds_write":
            lgkm.append(set())
        elif k == "vload":
            vm.append(set(it.writes()))
        elif k in ("dma", "vstore"):
            vm.append(set())
        # lgkmcnt is a 4-bit counter: the hardware holds the 16th LDS operation back until the oldest one has
        # completed, so at most 15 are ever outstanding and the oldest entries beyond that are known to be done
        if len(lgkm) > 15:
            if SATURATION_WAITS:     # A/B knob: the explicit form (an s_waitcnt lgkmcnt(15) behind every such read)
                out.append(it)
                out.append(Instr("s_waitcnt", kind="wait", mods={"lgkmcnt": 15}))
                del lgkm[:len(lgkm) - 15]
                continue
            del lgkm[:len(lgkm) - 15]
        if len(vm) > 63:
            del vm[:len(vm) - 63]
        out.append(it)
    if strict_tail and any(vm):
        raise RuntimeError("global load into a register outstanding at the end of the block")
    return out


# ------------------------------------------------------------------ hazards
def _wait_states(it: Instr) -> int:
    if it.kind == "nop":
        return it.mods["n"] + 1
    if it.kind == "label":
        return 0
    return 1


def fix_hazards(items: Sequence[Instr], loop: bool = False) -> List[Instr]:
    """Pad with s_nop where gfx950 needs software wait states and the stream does not provide them:
        MFMA (8 pass) result  -> any non-MFMA access to those registers, or an MFMA reading them as A / B : 14 states
                                 (as srcC of the next MFMA on the same registers: 0 ... accumulate chains are free)
        VALU / LDS-return write -> MFMA A / B / C operand                                                  :  2
        transcendental result   -> VALU use                                                               :  1
        VALU write              -> v_permlane32_swap / v_readfirstlane of it                              :  2
        SALU write of M0        -> LDS-DMA                                                                :  1
        VALU write of an SGPR (v_readfirstlane) -> VMEM using it                                          :  5
        buffer store of more than 64 bits -> a write of the VGPRs that hold its data                      :  2
                                 (the store reads its data after it has issued: the finished item's stores of the strip
                                 forward, spread by the scheduler, were followed at once by the next pair's v_accvgpr_read
                                 into the same registers - garbage in every second column group on MI355X, right in the emulator)
    `loop`: the list is a loop body, the tail feeds the head (checked by running over two copies)."""
    def pass_once(seq: List[Instr], carry_in) -> (List[Instr], list):
        out: List[Instr] = []
        # recent history: list of (instr, states_since)
        hist: List[list] = [list(h) for h in carry_in]

        def need(it: Instr) -> int:
            req = 0
            rd, wr = set(it.reads()), set(it.writes())
            for prev, since in hist:
                pw = set(prev.writes())
                if prev.kind == "mfma":
                    hit_rw = pw & (rd | wr)
                    if hit_rw:
                        if it.kind == "mfma":
                            # same registers taken whole as srcC (accumulate): free; as A/B operand or partial: hazard
                            c = it.src[2]
                            ab = set(it.src[0].regs()) | set(it.src[1].regs())
                            if pw & ab:
                                req = max(req, 14 - since)
                            elif isinstance(c, Reg) and set(c.regs()) == pw and set(it.dst[0].regs()) == pw:
                                pass
                            else:
                                req = max(req, 14 - since)
                        else:
                            req = max(req, 14 - since)
                elif prev.kind in ("valu", "trans"):
                    if it.kind == "mfma" and pw & rd:
                        req = max(req, 2 - since)
                    if prev.kind == "trans" and it.kind in ("valu", "trans", "mfma") and pw & rd:
                        req = max(req, 1 - since)
                    if it.op in ("v_permlane32_swap_b32", "v_permlane16_swap_b32", "v_readfirstlane_b32") and pw & rd:
                        req = max(req, 2 - since)
                    if prev.op == "v_readfirstlane_b32" and it.kind in ("dma", "vload", "vstore") and pw & rd:
                        req = max(req, 5 - since)
                elif prev.kind == "salu":
                    if ("m0", 0) in pw and it.kind == "dma":
                        req = max(req, 1 - since)
                elif prev.kind == "vstore":
                    if prev.src[0].n > 2 and set(prev.src[0].regs()) & wr:
                        req = max(req, 2 - since)
            return req

        for it in seq:
            if it.kind != "label":
                r = need(it)
                while r > 0:
                    n = min(r, 8)
                    nop = Instr("s_nop", mods={"n": n - 1}, kind="nop", cost=4 * n, note="hazard pad")
                    out.append(nop)
                    for h in hist:
                        h[1] += n
                    r -= n
            out.append(it)
            ws = _wait_states(it)
            for h in hist:
                h[1] += ws
            hist = [h for h in hist if h[1] < 16]
            if it.kind in ("mfma", "valu", "trans", "salu", "vstore"):
                hist.append([it, 0])
        return out, hist

    seq = list(items)
    out, hist = pass_once(seq, [])
    if loop:
        out, _ = pass_once(seq, hist)
    return out


def finish_block(items: Sequence[Instr], loop: bool = False) -> List[Instr]:
    """waits + hazard padding for a block whose order is final"""
    return fix_hazards(insert_waits(items), loop=loop)
