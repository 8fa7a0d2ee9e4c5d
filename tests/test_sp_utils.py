"""SURVEY section 8 f-4 on CPU: sequence-parallel index math against the oracle, the collectives on 2 gloo ranks, and
the generation patch's install / restore logic."""
import os
import socket
import subprocess
import sys

import torch

from oracle import sink_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sp_extended_keys_reproduce_the_global_rows():
    from sink_attention.sp_utils import sp_extended_kv
    g = torch.Generator().manual_seed(3)
    B, Hq, Hkv, D = 1, 4, 2, 8
    for (P, n, ns, W) in [(4, 16, 4, 10), (2, 24, 0, 24), (3, 8, 5, 40), (4, 8, 12, 3), (2, 16, 4, 0)]:
        N = P * n
        q, k, v = (torch.randn(B, h, N, D, generator=g, dtype=torch.float64) for h in (Hq, Hkv, Hkv))
        sa = torch.randn(Hq, generator=g, dtype=torch.float64)
        ref, _ = O.sink_attention_dense(q, k, v, ns, W, sa)
        for r in range(P):
            k_ext, v_ext, lead = sp_extended_kv(k, v, r, n, ns, W)
            q_ext = torch.cat([torch.zeros(B, Hq, lead, D, dtype=torch.float64), q[:, :, r * n:(r + 1) * n]], dim=2)
            out, _ = O.sink_attention_dense(q_ext, k_ext, v_ext, ns, W, sa)
            assert (out[:, :, lead:] - ref[:, :, r * n:(r + 1) * n]).abs().max() < 1e-12, (P, n, ns, W, r)
            assert k_ext.shape[2] <= min(ns, r * n) + max(W - 1, 0) + n


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sp_collectives_two_gloo_ranks():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "sp_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"),
                         cwd=ROOT)
    assert out.returncode == 0 and "SP_WORKER_OK" in out.stdout, out.stderr[-3000:]


def test_generation_patch_installs_and_restores():
    import transformers.modeling_flash_attention_utils as fa_utils
    import sink_attention.generate_patch as gp
    from sink_attention import SinkAttentionCache, patch_for_generation, unpatch_generation
    orig = fa_utils._flash_attention_forward
    cache = patch_for_generation(None, num_sink=2, window_size=16)
    try:
        assert isinstance(cache, SinkAttentionCache) and cache.num_sink == 2 and cache.window_size == 16
        assert fa_utils._flash_attention_forward is gp._generation_flash_attention_forward
        patch_for_generation(None, num_sink=3, window_size=8)      # patching twice keeps the TRUE original
        assert gp._original_flash_attention_forward is orig and gp._GENERATION_CONFIG["num_sink"] == 3
        # unsupported calls reach the saved original untouched
        seen = {}
        gp._original_flash_attention_forward = lambda *a, **k: seen.setdefault("args", (a, k)) or "fallback"
        q = torch.zeros(1, 4, 2, 8)
        gp._generation_flash_attention_forward(q, q, q, None, 4, is_causal=False)
        assert seen["args"][1]["is_causal"] is False
        gp._original_flash_attention_forward = orig
    finally:
        unpatch_generation()
    assert fa_utils._flash_attention_forward is orig and gp._original_flash_attention_forward is None
