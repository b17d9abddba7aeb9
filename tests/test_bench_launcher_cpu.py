"""bench.py --gpus N started as a plain `python bench.py --gpus N` must start its own N ranks (the driver may call
it that way): a rehearsal on CPU -- the children stop before any GPU work (PN2_BENCH_DRY_LAUNCH) -- checks that
every rank comes up with the right RANK / WORLD_SIZE and that a failing rank fails the parent."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + list(argv), env=env, capture_output=True,
                          text=True, timeout=240)


def test_plain_invocation_spawns_its_ranks():
    r = _run({"PN2_BENCH_DRY_LAUNCH": "1"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert sorted(l["rank"] for l in lines) == [0, 1]
    assert all(l["world"] == 2 for l in lines)


def test_failing_rank_fails_the_parent():
    # no GPU in this container: every rank exits with the "needs a HIP device" error, and so must the parent
    r = _run({}, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    import torch
    if torch.cuda.is_available():
        return
    assert r.returncode != 0
    assert "HIP device" in (r.stderr + r.stdout)
