"""`python bench.py --gpus N` must start its own ranks (reference launcher: cs_train.py:164-174).
CPU: the launcher path with --dry-run (gloo rendezvous, barrier, max-over-ranks, rank 0's JSON line, exit code)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_self_launch_two_ranks_dry_run():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_world"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert sorted(out["ranks"]) == [[0, 0], [1, 1]]
    assert out["max_over_ranks"] == 2.0


def test_self_launch_propagates_a_failing_rank():
    # a rank that dies must turn into a non-zero exit code of `python bench.py --gpus 2`, and no JSON line
    r = _run(["--gpus", "2", "--dry-run"], {"ONIRIS_DRY_RUN_FAIL_RANK": "1", "TORCH_DISTRIBUTED_DEBUG": "OFF"}, timeout=600)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_mismatched_launcher_is_refused():
    r = _run(["--gpus", "4", "--dry-run"], {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "disagree" in (r.stderr + r.stdout)


def test_watchdog_names_the_stage_of_a_hung_collective():
    # rank 1 stops answering after the rendezvous: rank 0's watchdog must end it with exit code 124 and the stage name,
    # well before any outer timeout -- the first 8-GPU run must be diagnosable from its stderr alone
    r = _run(["--gpus", "2", "--dry-run", "--watchdog", "5"], {"ONIRIS_DRY_RUN_HANG_RANK": "1", "TORCH_DISTRIBUTED_DEBUG": "OFF"},
             timeout=300)
    assert r.returncode != 0
    assert "watchdog: rank 0 made no progress" in r.stderr and "all_reduce(MAX)" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
