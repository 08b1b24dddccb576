"""The reference command line end to end on the HIP path (reference: src/train.py + train_utils/pretrain.py): a few synthetic
epochs of `train.py -learn_framework=FOCAL`, including the every-10-epochs branch (KNN estimator on the training features,
validation / test loss + accuracy, latest / best weights) and the build's resume extension."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("model", ["SW_Transformer", "DeepSense"])
def test_train_py_runs_validates_checkpoints_and_resumes(model, tmp_path):
    src = os.path.join(ROOT, "focal_amd", "src")
    base = [sys.executable, os.path.join(src, "train.py"), f"-model={model}", "-dataset=MOD", "-learn_framework=FOCAL",
            "-batch_size=16", "-synthetic_batches=2"]
    r = subprocess.run(base + ["-epochs=2"], capture_output=True, text=True, timeout=900, cwd=src)
    log = r.stdout + r.stderr
    assert r.returncode == 0, log[-3000:]
    for needle in ("Val loss:", "Val acc:", "Test loss:", "Total processing time"):
        assert needle in log, (needle, log[-2000:])
    wdir = os.path.join(ROOT, "weights", f"MOD_{model}")
    for f in ("latest", "best", "train_state"):
        assert os.path.exists(os.path.join(wdir, f"MOD_{model}_pretrain_{f}.pt")), f
    r2 = subprocess.run(base + ["-epochs=3", "-resume"], capture_output=True, text=True, timeout=900, cwd=src)
    log2 = r2.stdout + r2.stderr
    assert r2.returncode == 0, log2[-3000:]
    assert "Total processing time" in log2
