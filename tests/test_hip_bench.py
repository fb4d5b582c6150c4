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
    assert "four hipGraph replays" in d["config"]["launch"] and d["value"] > 0
