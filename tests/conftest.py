import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "focal_amd", "src")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cfg():
    from oracle.config import load_config
    return load_config()


def make_args(cfg, model, device, compute_dtype=None, tag=None):
    import argparse
    return argparse.Namespace(model=model, dataset="MOD", device=device, train_mode="contrastive", learn_framework="FOCAL",
                              stage="pretrain", task="vehicle_classification", tag=tag, dataset_config=cfg,
                              compute_dtype=compute_dtype)


def no_dropout(cfg):
    import copy
    c = copy.deepcopy(cfg)
    c["DeepSense"]["dropout_ratio"] = 0.0
    c["SW_Transformer"]["dropout_ratio"] = 0.0
    c["SW_Transformer"]["drop_path_rate"] = 0.0
    c["SW_Transformer"]["attn_drop_rate"] = 0.0
    return c


_OBSERVED = {}


def record_observed(key, value):
    """Observed parity errors of this run -> gpurun_out/observed_parity.json (copied to tests/golden/OBSERVED_r5.json (earlier rounds: OBSERVED_r3.json, OBSERVED_r4.json) and
    committed: VERDICT r1 asked for the measured errors behind every tolerance)."""
    import json
    _OBSERVED[key] = float(value)
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "observed_parity.json")
        old = json.load(open(path)) if os.path.exists(path) else {}
        old.update(_OBSERVED)
        json.dump(old, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
