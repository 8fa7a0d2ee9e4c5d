"""N>1 control path of bench.py on CPU: 2 gloo ranks, same launch line the driver uses (torch.distributed.run),
checking rendezvous on 127.0.0.1, the barrier, the MAX-over-ranks step time and the whole-job (weak-scaling) value.
The data path has no collective: each rank owns its own batch."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, steps=3):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", str(steps), "--warmup", "0", "--dry-run"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_two_gloo_ranks_report_max_time_and_job_value():
    d2 = _run(2)
    assert d2["n_gpus"] == 2 and d2["scaling"] == "weak" and d2["global_batch"] == 8
    # rank 1 sleeps 20 ms per step, rank 0 10 ms: the job's step time is the slowest rank's
    assert 19.0 <= d2["ms_per_step"] < 60.0, d2
    # whole-job value = 2 ranks' FLOPs / max time
    from oracle.sink_oracle import flops_fwd_bwd
    f = flops_fwd_bwd(4, 32, 8192, 128, 4, 4096)
    assert abs(d2["value"] - 2 * f / (d2["ms_per_step"] * 1e-3) / 1e12) / d2["value"] < 1e-2


def test_single_rank_dry_run():
    d1 = _run(1)
    assert d1["n_gpus"] == 1 and 9.0 <= d1["ms_per_step"] < 40.0
