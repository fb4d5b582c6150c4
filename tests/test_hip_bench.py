"""bench.py's output contract: exactly ONE line on stdout, a JSON object with the driver's keys -- also when RCCL is in the
process (its version banner goes through the C stdio and would otherwise land behind the JSON line)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline")


def _run(extra_env, *flags):
    env = dict(os.environ, **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "3", "--no-cpu-baseline",
                        "--no-fast-extra", *flags], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_prints_one_json_line():
    d = _run({})
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9


def test_bench_prints_one_json_line_with_rccl_in_the_process():
    port = 29700 + os.getpid() % 200
    d = _run({"FNEUS_DP_SINGLE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"},
             "--no-profile")
    # the job has measured both forms of its gradient exchange at start-up and says which one it kept, with both timings
    assert ("four hipGraph replays" in d["config"]["launch"] or "three hipGraph replays" in d["config"]["launch"]) and d["value"] > 0
    assert "chosen at start-up" in d["config"]["parallelism"] and " ms vs single " in d["config"]["parallelism"]


def test_bench_under_torchrun_with_two_ranks_prints_one_json_line():
    """the driver's N > 1 launch (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`), here with two
    ranks sharing the box's one GPU over gloo: ONE JSON line on the launcher's stdout, whole-job throughput, max-over-ranks time"""
    port = 29100 + os.getpid() % 100
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, FNEUS_DIST_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "dp2" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 512 * 128 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert "four hipGraph replays" in d["config"]["launch"] or "three hipGraph replays" in d["config"]["launch"]
    assert "chosen at start-up" in d["config"]["parallelism"]
    # one line per rank on stderr, and each rank's decision (counted in the text: two ranks write to one pipe, lines can interleave)
    assert r.stderr.count("] device cuda:") == 2 and r.stderr.count("gradient exchange:") >= 2, r.stderr[-2000:]
