"""Data-parallel parity: two ranks, each with half of a global batch, must reproduce the single-device result on the
whole batch -- all-gathered embeddings for the global negatives / ranking, SUM all-reduce of the gradients, and for DeepSense
cross-rank BatchNorm statistics.  The two ranks share the box's one GPU over gloo (see tests/dp_worker.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("model", ["SW_Transformer", "DeepSense"])
def test_two_ranks_equal_single_device(model):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533" if model == "DeepSense" else "29534", os.path.join(HERE, "dp_worker.py"), model, "16"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "worst gradient error" in r.stdout, tail
