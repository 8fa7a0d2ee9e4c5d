#!/usr/bin/env python3
"""Where a wave of the strip kernel spends its cycles (library built with EXTRA=-DSFA_STRIP_STAMPS): per q tile, cycles
in [top: missing tiles + barrier | prefetch issue | tile loop | normalise + stores | end barrier + LDS write + Q copy].
usage: python tools/stamps_strip.py [--cfg C4b4]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sink-flash-attention-kernel_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch

from kbench import CFG
from sink_attention import _native, sink_flash_attention

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="C4b4")
args = ap.parse_args()
B, Hq, Hkv, N, D, ns, W, aux = CFG[args.cfg]
dev = "cuda"
torch.manual_seed(1)
q = torch.randn(B, Hq, N, D, device=dev, dtype=torch.bfloat16)
k = torch.randn(B, Hkv, N, D, device=dev, dtype=torch.bfloat16)
v = torch.randn(B, Hkv, N, D, device=dev, dtype=torch.bfloat16)
sa = torch.randn(Hq, device=dev) if aux else None
lib = _native.lib()
dbg = torch.zeros(4096 * 8 * 8, dtype=torch.int64, device=dev)
lib.sfa_debug_set_ptr(dbg.data_ptr())
for _ in range(3):
    sink_flash_attention(q, k, v, num_sink=ns, window_size=W, s_aux=sa)
torch.cuda.synchronize()
path = _native.last_path()
print(path)
strip = int(path.split("strip")[1].split("_")[0])
d = dbg.view(-1, 8).cpu()
live = d[:, 2] > 0
t = d[live][:, :5].float() / strip
names = ["top (missing tiles + barrier)", "prefetch issue", "tile loop", "normalise + stores", "end barrier + LDS write + Q copy"]
print("waves:", int(live.sum()), " q tiles per strip:", strip, " cycles per q tile and wave (mean over waves):")
for i, n in enumerate(names):
    print("  %-36s %8.0f" % (n, t[:, i].mean().item()))
print("  %-36s %8.0f" % ("total", t.sum(1).mean().item()))
