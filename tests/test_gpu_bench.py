"""bench.py's multi-rank control flow before the driver's 8-GPU run meets it: two ranks launched exactly like the driver does
(python -m torch.distributed.run, one process per rank), both on device 0 of this one-GPU box through the test hooks
VSE_DIST_BACKEND=gloo / VSE_BENCH_DEVICE=0 (bench.py main()).  What runs: process-group init, per-rank frame shards, the
barrier-bracketed timed region, the all_reduce(MAX) of the ranks' times, the ONE variable-length record gather
and the single JSON line of rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, env):
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_one_json_line(ctx):
    env = dict(os.environ, VSE_DIST_BACKEND="gloo", VSE_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    flags = ["--steps", "2", "--warmup", "1", "--batch", "16", "--no-cpu-baseline", "--no-roofline", "--other-mode-steps", "0"]
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), "bench.py", "--gpus", "2"] + flags, env)
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["warmup"] == 1 and two["scaling"] == "weak"
    assert two["config"]["records_gathered"] == 2 * 16 * 2            # ranks x frames per step x steps, gathered once
    assert two["config"]["frames_per_gpu_step"] == 16 and two["value"] > 0
    assert abs(two["value"] - 2 * 16 * 2 / (two["ms_per_step"] * 2 / 1e3)) < 0.02 * two["value"]      # whole-job frames / max time
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + flags, dict(os.environ))
    assert one["n_gpus"] == 1 and one["config"]["records_gathered"] == 16 * 2
    assert one["config"]["boxes_last_step"] > 0
