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


def test_one_dead_rank_ends_the_others_promptly():
    """A rank that exits early must not leave the launcher waiting for ranks that sit at a collective (here: sleep
    for a minute): the launcher polls, terminates the survivors and returns the failing status."""
    import time
    t0 = time.time()
    r = _run(["--gpus", "3"], {"FUSG_BENCH_DRYRUN": "1", "FUSG_BENCH_DRYRUN_FAIL_RANK": "1", "FUSG_BENCH_DRYRUN_SLEEP": "60"})
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert time.time() - t0 < 30


def test_cpu_sweep_never_uses_every_cpu_of_a_big_host():
    """Round 2: the 256-thread point of the sweep ran at 0.0115 crops/s and took 260 s of the driver's 361 s."""
    sys.path.insert(0, REPO)
    import bench
    assert bench.cpu_sweep_points(256, 128) == [8, 16, 32, 64]
    assert bench.cpu_sweep_points(16, 16) == [8, 16]
    assert bench.cpu_sweep_points(4, 4) == [4]
    assert bench.CPU_LEG_BUDGET_S <= 60 and bench.CPU_SWEEP_POINT_S <= 10


def test_oracle_crop_pass_deadline():
    import time
    import torch
    sys.path.insert(0, REPO)
    import oracle
    from future_urban_scene_generation_amd.pipeline import load_schema, synth_batch
    from future_urban_scene_generation_amd.synth import synth_state_dict
    sds = {n: synth_state_dict(n, load_schema(n), 0) for n in ("hg", "icn", "vunet")}
    b = synth_batch(1, 128, "cpu")
    with pytest.raises(TimeoutError):
        oracle.crop_pass(sds, b, deadline=time.perf_counter() - 1.0)
    torch.manual_seed(0)
    assert "vunet_u8" in oracle.crop_pass(sds, b, deadline=time.perf_counter() + 600)
