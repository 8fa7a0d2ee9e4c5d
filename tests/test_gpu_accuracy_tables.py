"""Accuracy tables: max / mean absolute error and cosine similarity of the HIP path against the fp64 oracle, for the
configurations the reference tabulates in its README (reference: tests/numerical_accuracy.py:32-77 forward table,
:79-120 gradient table), plus the head dims and dtypes this build adds.  The tables are written to
gpurun_out/accuracy_tables.log (copied to profiles/ per round); the assertions are the reference's tolerances.
"""
import os

import pytest
import torch

from util import make_qkv, oracle_bwd, oracle_fwd, rand

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FWD = [  # B, Hq, Hkv, N, D, ns, W, dtype, label         (numerical_accuracy.py:38-49, then this build's additions)
    (1, 4, 4, 256, 64, 4, 64, torch.float16, "MHA fp16 N=256"),
    (1, 4, 4, 512, 64, 4, 64, torch.float16, "MHA fp16 N=512"),
    (1, 4, 4, 1024, 64, 4, 128, torch.float16, "MHA fp16 N=1024"),
    (1, 4, 4, 2048, 64, 4, 256, torch.float16, "MHA fp16 N=2048"),
    (1, 8, 2, 512, 64, 4, 64, torch.float16, "GQA 4:1 fp16 N=512"),
    (1, 32, 8, 512, 128, 4, 128, torch.float16, "GQA 4:1 fp16 D=128 N=512"),
    (1, 4, 4, 512, 64, 4, 64, torch.bfloat16, "MHA bf16 N=512"),
    (1, 4, 4, 512, 64, 0, 64, torch.float16, "pure window fp16 N=512"),
    (1, 4, 4, 512, 64, 16, 128, torch.float16, "16 sinks fp16 N=512"),
    (1, 4, 4, 512, 64, 4, 1, torch.float16, "sink+self only fp16 N=512"),
    (1, 8, 2, 2048, 128, 4, 1024, torch.bfloat16, "GQA 4:1 bf16 D=128 N=2048 W=1024 (hand-placed kernels)"),
    (1, 8, 2, 2048, 128, 4, 1024, torch.float16, "GQA 4:1 fp16 D=128 N=2048 W=1024 (hand-placed kernels)"),
    (1, 8, 1, 2048, 80, 0, 128, torch.bfloat16, "gpt-oss sliding layer bf16 D=80 N=2048 W=128"),
    (1, 4, 2, 1024, 256, 4, 256, torch.bfloat16, "GQA 2:1 bf16 D=256 N=1024"),
    (1, 4, 4, 512, 32, 4, 64, torch.float16, "MHA fp16 D=32 N=512"),
    (1, 4, 4, 256, 64, 4, 64, torch.float32, "MHA fp32 N=256 (exact-f32 kernels)"),
]
GRAD = [  # numerical_accuracy.py:85-90 (fp16 kernel against fp32 values), then additions
    (1, 4, 4, 128, 64, 4, 32, torch.float16, "MHA N=128 sink=4 win=32"),
    (1, 4, 4, 256, 64, 4, 64, torch.float16, "MHA N=256 sink=4 win=64"),
    (1, 8, 2, 256, 64, 4, 64, torch.float16, "GQA N=256 sink=4 win=64"),
    (1, 4, 4, 256, 128, 4, 64, torch.float16, "MHA D=128 N=256 sink=4 win=64"),
    (1, 8, 2, 2048, 128, 4, 1024, torch.bfloat16, "GQA bf16 D=128 N=2048 W=1024 (hand-placed kernels)"),
    (1, 8, 1, 2048, 80, 0, 128, torch.bfloat16, "gpt-oss sliding layer bf16 D=80 N=2048 W=128"),
]


def _stats(a, e):
    a, e = a.detach().double().cpu(), e.detach().double().cpu()
    d = (a - e).abs()
    cos = torch.nn.functional.cosine_similarity(a.reshape(1, -1), e.reshape(1, -1)).item()
    return d.max().item(), d.mean().item(), cos


def test_accuracy_tables():
    from sink_attention import sink_flash_attention
    lines = ["Forward: HIP path vs fp64 oracle on the same (rounded) inputs",
             "%58s | %12s | %12s | %10s" % ("Config", "Max Abs Err", "Mean Abs Err", "Cosine Sim")]
    for B, Hq, Hkv, N, D, ns, W, dt, label in FWD:
        q, k, v, _ = make_qkv(B, Hq, Hkv, N, D, dt)
        out = sink_flash_attention(q.to(DEV), k.to(DEV), v.to(DEV), num_sink=ns, window_size=W)
        ref, _ = oracle_fwd(q, k, v, ns, W)
        mx, mean, cos = _stats(out, ref)
        lines.append("%58s | %12.6f | %12.6f | %10.8f" % (label, mx, mean, cos))
        tol = {torch.float16: 1e-2, torch.bfloat16: 4e-2, torch.float32: 2e-5}[dt]   # reference: test_sink_attention.py:68, test_s_aux.py:180-183
        assert mx < tol and cos > 0.9999, (label, mx, cos)
    lines += ["", "Gradients: HIP path vs fp64 oracle", "%58s | %10s | %10s | %10s" % ("Config", "dQ MaxErr", "dK MaxErr", "dV MaxErr")]
    for B, Hq, Hkv, N, D, ns, W, dt, label in GRAD:
        q, k, v, g = make_qkv(B, Hq, Hkv, N, D, dt)
        do = rand((B, Hq, N, D), g, dt)
        qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
        sink_flash_attention(qd, kd, vd, num_sink=ns, window_size=W).backward(do.to(DEV))
        dq_r, dk_r, dv_r, _ = oracle_bwd(q, k, v, do, ns, W)
        errs = [_stats(a.grad, r)[0] for a, r in ((qd, dq_r), (kd, dk_r), (vd, dv_r))]
        lines.append("%58s | %10.6f | %10.6f | %10.6f" % (label, *errs))
        scale = max(1.0, dk_r.abs().max().item())
        assert max(errs) < 5e-2 * scale, (label, errs)                             # test_sink_attention.py:94-96
    text = "\n".join(lines)
    print(text)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "accuracy_tables.log"), "w") as f:
        f.write(text + "\n")
