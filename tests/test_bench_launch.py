"""bench.py --gpus N must start its own ranks (the driver may call it without a launcher) and print ONE JSON line.
Rehearsed on CPU: --dry skips every kernel, ABUB_BENCH_BACKEND=gloo replaces RCCL; what runs is the real launch /
barrier / max-over-ranks / gather / report protocol of the multi-GPU path."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ, ABUB_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_gpus2_launches_itself_and_prints_one_line():
    p = _run(["--gpus", "2", "--dry", "--steps", "3", "--warmup", "1", "--min-seconds", "0.05"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    t = out["config"]["timing"]
    assert t["blocks"] >= 1 and t["steps_per_block"] == 3
    per_rank = t["ms_per_step_by_rank"]
    assert len(per_rank) == 2 and per_rank[1] > per_rank[0] * 1.5  # the dry straggler (rank 1 sleeps twice as long) shows
    assert out["ms_per_step"] >= per_rank[1] * 0.9                  # a block is as slow as its slowest rank
    assert t["ms_per_step_min"] <= out["ms_per_step"] <= t["ms_per_step_max"]


def test_bench_rejects_a_launcher_with_the_wrong_world_size():
    p = _run(["--gpus", "4", "--dry"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode == 2 and "WORLD_SIZE=2" in p.stderr


def test_bench_single_rank_dry():
    p = _run(["--dry", "--steps", "2", "--warmup", "0", "--min-seconds", "0"])
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["config"]["timing"]["blocks"] == 1


def test_stdout_is_exactly_one_json_line():
    """The driver reads ONE JSON line from stdout: bench.py moves everything else that lands on fd 1 (the host library's
    printf progress lines) to stderr."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 1 and json.loads(lines[0])["metric"]
