"""`bench.py --gpus N` never degrades to a silent 1-GPU run: without the GPUs it exits non-zero and prints the launch
line; a launcher whose WORLD_SIZE disagrees with --gpus is refused; the rank environments of the self-launch are what torchrun would set."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env_over):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FOCAL_BENCH_TEST_BACKEND")}
    env.update(env_over)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)


def test_more_gpus_than_visible_is_an_error_with_the_launch_line():
    import torch
    n = torch.cuda.device_count() + 2
    r = _run(["--gpus", str(n), "--steps", "2", "--warmup", "1"])
    assert r.returncode != 0
    assert "torch.distributed.run" in r.stderr and f"--nproc-per-node {n}" in r.stderr and not r.stdout.strip()


def test_world_size_must_agree_with_gpus():
    r = _run(["--gpus", "4", "--steps", "2", "--warmup", "1"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and not r.stdout.strip()


def test_rank_environments():
    sys.path.insert(0, ROOT)
    from focal_amd.launch import rank_environments
    old = os.environ.get("HIP_VISIBLE_DEVICES")
    os.environ["HIP_VISIBLE_DEVICES"] = "4,5,6,7"
    try:
        envs = rank_environments([0, 2, 3])
    finally:
        if old is None:
            del os.environ["HIP_VISIBLE_DEVICES"]
        else:
            os.environ["HIP_VISIBLE_DEVICES"] = old
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" and e["HIP_VISIBLE_DEVICES"] == "4,6,7" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1
    keep = rank_environments([0, 1], narrow_visible=False)
    assert all(e.get("HIP_VISIBLE_DEVICES") == old for e in keep)


def test_a_dead_rank_takes_the_job_down(tmp_path):
    sys.path.insert(0, ROOT)
    from focal_amd.launch import spawn_ranks
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(60)\n")
    import time
    t0 = time.time()
    assert spawn_ranks(str(script), [], [0, 1]) == 7
    assert time.time() - t0 < 30
