"""GPU: bench.py prints exactly one JSON line that honours the driver's contract (keys, types, roofline / config
objects); a short run (the measured value itself is not asserted, only its plausibility)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract(tmp_path):
    dump = str(tmp_path / "slice.pt")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2",
                          "--no-cpu-baseline", "--sustain-seconds", "0.5", "--dump-slice", dump], capture_output=True, text=True, timeout=900,
                         cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict)):
        assert isinstance(d[key], typ), (key, d[key])
    assert d["steps"] == 3 and d["warmup"] == 2 and d["n_gpus"] == 1 and d["vs_baseline"] is None
    assert d["unit"] == "TFLOP/s" and d["dtype"] == "bf16" and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2516.6
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.05 < r["frac"] < 1.0
    assert set(r["stage_ms"]) == {"fwd", "bwd_preprocess", "bwd_dkdv", "bwd_dq"}
    # whole-job value = algorithmic FLOPs / time, and the stages add up to (about) the step
    gflop = d["config"]["algorithmic_gflop_per_step_per_gpu"]
    assert abs(d["value"] - gflop / d["ms_per_step"]) / d["value"] < 1e-2
    assert 0.8 < sum(r["stage_ms"].values()) / d["ms_per_step"] <= 1.02
    assert 100.0 < d["value"] < 2516.6
    assert set(r["per_kernel"]) == {"fwd", "bwd_dkdv", "bwd_dq"} and r["fwd_frac"] == r["per_kernel"]["fwd"]["frac"]
    assert r["traffic"] is None or "profiles/" in r["traffic_source"]
    # the sustained region: back-to-back steps for at least the requested time, same units, a plausible rate
    su = d["sustained"]
    assert su["seconds"] >= 0.5 and su["steps"] >= 3 and abs(su["ms_per_step"] - su["seconds"] / su["steps"] * 1e3) < 1e-2
    assert 0.7 < su["value"] / d["value"] < 1.3

    # the numbers above must belong to a RIGHT answer: one (batch, KV head) unit of the bench's own tensors (results of
    # one more step on the same inputs) against the banded fp64 oracle, tolerances of the C3 parity test
    import torch
    from util import assert_close, oracle_bwd, oracle_fwd
    t = torch.load(dump)
    o_r, _ = oracle_fwd(t["q"], t["k"], t["v"], t["ns"], t["W"], banded=True)
    dq_r, dk_r, dv_r, _ = oracle_bwd(t["q"], t["k"], t["v"], t["do"], t["ns"], t["W"], banded=True)
    assert_close(t["o"], o_r.bfloat16(), 2e-2, 2e-2, "bench fwd")
    assert_close(t["dq"], dq_r, 5e-2, 5e-2, "bench dq")
    assert_close(t["dk"], dk_r, 5e-2 * max(1.0, dk_r.abs().max().item()), 5e-2, "bench dk")
    assert_close(t["dv"], dv_r, 5e-2 * max(1.0, dv_r.abs().max().item()), 5e-2, "bench dv")
