"""Data-parallel parity: two ranks, each with half of a global batch, must reproduce the single-device result on the
whole batch -- all-gathered embeddings for the global negatives / ranking, SUM all-reduce of the gradients, and for DeepSense
cross-rank BatchNorm statistics.  The two ranks share the box's one GPU over gloo (see tests/dp_worker.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("model,shard", [("SW_Transformer", "1"), ("SW_Transformer", "0"), ("DeepSense", "1")])
def test_two_ranks_equal_single_device(model, shard):
    """shard = FOCAL_LOSS_SHARD: the loss head row-sharded over the two ranks (the default from 6 ranks up) or replicated."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="8", FOCAL_LOSS_SHARD=shard)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533" if model == "DeepSense" else ("29534" if shard == "1" else "29535"), os.path.join(HERE, "dp_worker.py"), model, "16"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "worst gradient error" in r.stdout, tail


@pytest.mark.parametrize("model", ["SW_Transformer", "DeepSense"])
def test_bench_contract_with_two_ranks(model):
    """The driver's N > 1 launch line, verbatim, on the box's one GPU (FOCAL_BENCH_TEST_BACKEND=gloo puts both ranks on cuda:0):
    three hipGraph segments with the two collectives issued eagerly between replays, MAX-over-ranks timing, ONE JSON line from
    rank 0 whose value is the whole-job rate."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="8", FOCAL_BENCH_TEST_BACKEND="gloo",
               FOCAL_LOSS_SHARD="1" if model == "SW_Transformer" else "0")  # both segmentations of the step (sharded / replicated head)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541" if model == "DeepSense" else "29542", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4",
           "--warmup", "2", "--batch", "16", "--model", model]  # no diagnostic flags: rank 0 also measures its roofline (local steps)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, tail
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 32 and out["config"]["parallelism"] == "dp2" and out["config"]["hip_graph"] is True
    assert out["value"] > 0 and abs(out["value"] - 32 / (out["ms_per_step"] * 1e-3)) < 0.02 * out["value"]
    assert out["vs_baseline"] is None and out["cpu_baseline"] is None  # the CPU baseline is timed at N = 1 only
    assert out["roofline"] is not None and out["roofline"]["frac"] > 0
    # day-one diagnostics of the data-parallel step (what the first RCCL runs will be read with)
    dp = out["dp"]
    assert dp["backend"] == "gloo" and dp["world"] == 2 and dp["ranks_seen"] == 2 and dp["rccl_version"] is None
    assert dp["loss_head_sharded"] is (model == "SW_Transformer") and dp["split_backward"] is (model == "SW_Transformer")
    seg = dp["us_segments"]
    assert any(k.startswith("A:") for k in seg) and any(k.startswith("C:") for k in seg) and all(v >= 0 for v in seg.values())
    assert dp["us_exchange"] > 0 and dp["us_allreduce_exposed"] > 0
    # (no bound against ms_per_step here: over gloo the collectives are host round trips of two processes sharing one GPU, and the 5
    # diagnostic steps run after rank 0's peers have gone idle; over RCCL the sum is the step)
    assert dp["us_step_from_events"] >= dp["us_exchange"] + dp["us_allreduce_exposed"]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (no WORLD_SIZE in the environment) spawns two fresh ranks itself (focal_amd/launch.py)
    before any HIP call -- never a silent 1-GPU run.  Both ranks share the box's one GPU over gloo here."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="8", FOCAL_BENCH_TEST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "16", "--no-roofline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, tail
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 32
    assert out["dp"]["world"] == 2 and out["dp"]["ranks_seen"] == 2
    assert out["warmup_effective"] >= 2 and len(out["step_series"]["timed_ms"]) == 4


def test_two_ranks_draw_the_same_views():
    """VERDICT r5 item 3a on the device: both ranks of a job draw identical focal_view_plan records step after step (one broadcast draw
    seed, focal_view_draw_shared), and different dropout masks (tests/dp_views_worker.py)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(HERE, "dp_views_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "plans identical on all ranks: True" in r.stdout, tail
