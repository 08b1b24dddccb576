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


def test_pretraining_from_packed_shards_and_from_sample_files(cfg, tmp_path):
    """Real-data paths: MOD-shaped `.pt` samples -> (a) the reference-style DataLoader with the sequence-aware batch sampler,
    (b) a packed shard with the prefetching loader; one pretraining epoch from each on the HIP path, same first-step loss."""
    import copy
    import random
    import torch
    sys.path.insert(0, os.path.join(ROOT, "focal_amd", "src"))
    from conftest import make_args
    from data_augmenter.Augmenter import Augmenter
    from input_utils.multi_modal_dataloader import create_dataloader
    from input_utils.packed_shards import pack_index
    from train_utils.model_selection import init_backbone_model, init_loss_func, init_pretrain_framework
    from train_utils.loss_calc_utils import calc_pretrain_loss
    g = torch.Generator().manual_seed(4)
    files = []
    for s in range(4):
        for k in range(8):
            f = str(tmp_path / f"seq{s}_shake_{k}.pt")
            torch.save({"label": torch.tensor(k % 3), "flag": {"shake": {"audio": True, "seismic": True}},
                        "data": {"shake": {"audio": torch.randn(1, 10, 1600, generator=g), "seismic": torch.randn(1, 10, 20, generator=g)}}}, f)
            files.append(f)
    idx = tmp_path / "index.txt"
    idx.write_text("\n".join(files) + "\n")
    losses = {}
    for kind in ("files", "packed"):
        c = copy.deepcopy(cfg)
        for k in ("dropout_ratio", "drop_path_rate", "attn_drop_rate"):
            c["SW_Transformer"][k] = 0.0
        c["FOCAL"]["random_augmenters"] = {"time_augmenters": ["no"], "freq_augmenters": ["no"]}
        args = make_args(c, "SW_Transformer", torch.device("cuda"), "fp32")
        args.batch_size, args.workers, args.sequence_sampler, args.label_ratio = 16, 0, True, 1.0
        c["pretrain_index_file"] = str(idx) if kind == "files" else pack_index(args, str(idx), str(tmp_path / "pack"))
        torch.manual_seed(0)
        random.seed(9)
        loader = create_dataloader("train", args, batch_size=16, workers=0)
        assert len(loader) == 2
        model = init_pretrain_framework(args, init_backbone_model(args))
        from oracle.weights import fill_state_dict_
        fill_state_dict_(model.backbone.state_dict())
        model.train()
        loss_fn, aug = init_loss_func(args), Augmenter(args)
        vals = []
        for time_loc_inputs, labels in loader:
            assert labels.shape[0] == 16
            vals.append(calc_pretrain_loss(args, model, aug, loss_fn, time_loc_inputs).item())
        losses[kind] = vals
    assert all(abs(a - b) < 1e-4 * max(1.0, abs(a)) for a, b in zip(losses["files"], losses["packed"])), losses


@pytest.mark.parametrize("model", ["SW_Transformer", "DeepSense"])
def test_finetune_stage_runs_on_pretrained_weights(model):
    """`train.py -stage=finetune` (reference: train_utils/finetune.py): loads the pretraining weights, trains the classifier
    head only, validates with the classifier's own loss / accuracy, writes latest / best weights."""
    src = os.path.join(ROOT, "focal_amd", "src")
    base = [sys.executable, os.path.join(src, "train.py"), f"-model={model}", "-dataset=MOD", "-learn_framework=FOCAL",
            "-batch_size=16", "-synthetic_batches=2"]
    r = subprocess.run(base + ["-epochs=1"], capture_output=True, text=True, timeout=900, cwd=src)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    r = subprocess.run(base + ["-stage=finetune", "-epochs=2"], capture_output=True, text=True, timeout=900, cwd=src)
    log = r.stdout + r.stderr
    assert r.returncode == 0, log[-3000:]
    for needle in ("Training loss:", "Val acc:", "Test acc:", "Total processing time"):
        assert needle in log, (needle, log[-2000:])
    wdir = os.path.join(ROOT, "weights", f"MOD_{model}")
    assert any(f.endswith("finetune_latest.pt") for f in os.listdir(wdir))


def test_train_py_sustains_the_benchmarked_step():
    """`train.py` replays the captured step (focal_amd/graph_step.py) with the two random views of every step drawn on the device
    INSIDE the replayed graph (round 5: Augmenter.forward_random_pair): its steady-state windows/s (last epoch, logged by
    train_utils/pretrain.py) must be within 5 % of what bench.py measures on the same box with the batch handed over from the host
    (`--from-host`; fixed views) -- the observed ratio is recorded (`train_py.over_bench_from_host`)."""
    import json
    import re
    src = os.path.join(ROOT, "focal_amd", "src")
    r = subprocess.run([sys.executable, os.path.join(src, "train.py"), "-model=SW_Transformer", "-dataset=MOD", "-learn_framework=FOCAL",
                        "-batch_size=256", "-synthetic_batches=30", "-epochs=3"], capture_output=True, text=True, timeout=1500, cwd=src)
    log = r.stdout + r.stderr
    assert r.returncode == 0, log[-3000:]
    rates = [float(x) for x in re.findall(r"epoch 2: ([0-9.]+) windows/s", log)]
    assert rates, log[-2000:]
    assert "graph replays" in log and int(re.findall(r"(\d+) graph replays", log)[-1]) >= 60, log[-1500:]
    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--from-host", "--steps", "40", "--warmup", "10", "--no-cpu-baseline",
                        "--no-roofline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert b.returncode == 0, (b.stdout + b.stderr)[-3000:]
    bench = json.loads([ln for ln in b.stdout.splitlines() if ln.startswith("{")][-1])["value"]
    from conftest import record_observed
    record_observed("train_py.windows_per_s", rates[-1])
    record_observed("train_py.over_bench_from_host", rates[-1] / bench)
    assert "drawn on the device" in log, log[-1500:]
    assert rates[-1] >= 0.95 * bench, (rates, bench)
