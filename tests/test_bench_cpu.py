"""bench.py host logic that needs no GPU: the self-spawn of N ranks (`python bench.py --gpus N` without a launcher)."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=120)


def test_gpus_n_starts_n_ranks_itself():
    r = _run(["--gpus", "3", "--steps", "2"], {"FUSG_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr
    ranks = sorted((json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")), key=lambda d: d["rank"])
    assert [d["rank"] for d in ranks] == [0, 1, 2] and [d["local_rank"] for d in ranks] == [0, 1, 2]
    assert all(d["world"] == 3 and d["gpus"] == 3 for d in ranks)
    assert len({d["master"] for d in ranks}) == 1 and ranks[0]["master"].startswith("127.0.0.1:")


def test_single_gpu_does_not_spawn_and_launcher_env_is_respected():
    r = _run(["--gpus", "1"], {"FUSG_BENCH_DRYRUN": "1"})
    assert r.returncode == 0 and json.loads(r.stdout.strip())["world"] == 1
    # under torchrun (WORLD_SIZE set) the process is a rank, not a launcher
    r = _run(["--gpus", "2"], {"FUSG_BENCH_DRYRUN": "1", "WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"})
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1 and lines[0]["rank"] == 1 and lines[0]["world"] == 2


def test_failed_rank_gives_nonzero_exit():
    """No HIP device here: every started rank exits with an error, and so does the launcher (it must not report
    a number from fewer ranks than asked for)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode != 0
    assert not any(ln.startswith('{"metric"') for ln in r.stdout.splitlines())
