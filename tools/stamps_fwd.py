#!/usr/bin/env python3
"""Where a workgroup of the hand-placed forward kernel spends its life (library built with ASMGEN_FWD_STAMPS=1: every
body carries five s_memtime stamps, the HIP shell one at kernel entry).  Per workgroup class (tiles per workgroup):
cycles from entry to the asm statement (block-id decode, kernel arguments, s_aux), prologue (Q fragments, first three
tiles, S^T + softmax bookkeeping of tile 0), loop (per tile), epilogue until the stores are issued, store drain.
usage: python tools/stamps_fwd.py [--cfg C3]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sink-flash-attention-kernel_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch

from kbench import CFG
from sink_attention import _native, sink_flash_attention


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="C3")
    args = ap.parse_args()
    B, Hq, Hkv, N, D, ns, W, aux = CFG[args.cfg]
    dev = "cuda"
    torch.manual_seed(1)
    q = torch.randn(B, Hq, N, D, device=dev, dtype=torch.bfloat16)
    k = torch.randn(B, Hkv, N, D, device=dev, dtype=torch.bfloat16)
    v = torch.randn(B, Hkv, N, D, device=dev, dtype=torch.bfloat16)
    sa = torch.randn(Hq, device=dev) if aux else None
    lib = _native.lib()
    g = Hq // Hkv
    hpw = 4 if g % 4 == 0 else (2 if g % 2 == 0 else 1)
    nblk = B * Hkv * (g // hpw) * ((N + 64 * (4 // hpw) - 1) // (64 * (4 // hpw)))
    dbg = torch.zeros(nblk * 4 * 8, dtype=torch.int32, device=dev)
    lib.sfa_debug_set_ptr(dbg.data_ptr())
    for _ in range(3):
        sink_flash_attention(q, k, v, num_sink=ns, window_size=W, s_aux=sa)
    torch.cuda.synchronize()
    print(_native.last_path())
    d = dbg.view(nblk, 4, 8).cpu().long() & 0xFFFFFFFF
    nt = d[:, 0, 6]
    live = nt > 0
    dif = lambda a, b: ((d[:, :, a] - d[:, :, b]) & 0xFFFFFFFF).float()
    shell, pro, loop, epi, drain = dif(1, 0), dif(2, 1), dif(3, 2), dif(4, 3), dif(5, 4)
    total = dif(5, 0)
    print("workgroups:", int(live.sum()), "of", nblk)
    edges = [1, 8, 24, 48, 64, 1000]
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = live & (nt >= lo) & (nt < hi)
        if not m.any():
            continue
        f = lambda x: x[m].mean().item()
        t = nt[m].float().mean().item()
        print("  %3d..%3d tiles (%4d workgroups, mean %5.1f tiles): entry->asm %6.0f  prologue %6.0f  loop %7.0f (%5.0f per tile)  "
              "epilogue %6.0f  store drain %6.0f  | total %7.0f cycles, outside the loop %5.0f" % (
                  lo, hi - 1, int(m.sum()), t, f(shell), f(pro), f(loop), f(loop) / max(t - 1, 1), f(epi), f(drain), f(total),
                  f(total) - f(loop)))
    # gaps between consecutive workgroups of a CU cannot be seen from inside; kernel time / rounds - mean total is the rest


if __name__ == "__main__":
    main()
