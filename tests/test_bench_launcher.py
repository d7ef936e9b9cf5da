"""bench.py --gpus N must start its N ranks itself when no launcher did: the launch, the
rendezvous, MAX over ranks and the single JSON line from rank 0, exercised on CPU over gloo
(--launcher-dry-run: no GPU work); and the two failure modes - a rank that dies takes the run down with a
non-zero exit code, --gpus that disagrees with an external launcher's WORLD_SIZE is an error."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra)
    return env


def test_self_launch_two_ranks_one_json_line():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--launcher-dry-run"],
                         capture_output=True, text=True, timeout=300, env=_env())
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout                         # rank 0's JSON line and nothing else (no backend banners)
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["max_over_ranks"] == 2.0 and out["steps"] == 3 and out["warmup"] == 1


def test_gpus_must_match_external_world_size():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--launcher-dry-run"], capture_output=True, text=True,
                         timeout=120, env=_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_PORT="29999",
                                               MASTER_ADDR="127.0.0.1"))
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr


def test_failing_rank_fails_the_run():
    # an unknown flag makes every child exit 2 (argparse) before the rendezvous: the parent must report failure
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-dry-run", "--config", "nope"],
                         capture_output=True, text=True, timeout=120, env=_env())
    assert res.returncode != 0
