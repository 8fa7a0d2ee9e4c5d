"""Loader for tests/golden/*.npz (written by tests/golden/make_golden.py).

Undoes the storage compaction: float16 arrays -> float32, ``<key>__bf16`` uint16
arrays (high half of the fp32 word) -> float32 under ``<key>``.
"""
import glob
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        a = z[k]
        if k.endswith("__bf16"):
            a = (a.astype(np.uint32) << 16).view(np.float32)
            k = k[: -len("__bf16")]
        elif a.dtype == np.float16:
            a = a.astype(np.float32)
        out[k] = torch.from_numpy(np.ascontiguousarray(a)) if a.dtype != np.int64 or k != "meta" else a
    return out


def names(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def fwd_bwd_meta(g):
    B, Hq, Hkv, N, D, ns, W, seed = (int(x) for x in g["meta"])
    return dict(B=B, Hq=Hq, Hkv=Hkv, N=N, D=D, ns=ns, W=W)
