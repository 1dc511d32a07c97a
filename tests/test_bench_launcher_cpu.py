"""`python bench.py --gpus N` with no torchrun around it must launch its own ranks (the plain command
died on a WORLD_SIZE check).  Here: the launcher path with N=2 CPU ranks over gloo and a stand-in step
(--launcher-selftest: launch, barrier, max-over-ranks timing, all-gather of metric rows; no forward, no GPU)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout          # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_plain_command_self_launches_two_ranks():
    line = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--launcher-selftest"])
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["backend"] == "gloo"
    assert line["steps"] == 3 and line["warmup"] == 1
    assert len(line["per_rank_ms_per_step"]) == 2
    assert line["rows_gathered"] == 8 and line["rank_sum"] == 4.0       # 4 rows of rank 0 (0.0) + 4 of rank 1 (1.0)


def test_single_rank_needs_no_launcher():
    line = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--batch", "3", "--launcher-selftest"])
    assert line["n_gpus"] == 1 and line["world_size"] == 1 and line["rows_gathered"] == 3
