"""Supervised training from scratch (reference: train_utils/supervised_train.py:18-108, `train.py -learn_framework=no`)
against fixtures produced by the reference models (tests/golden/gen_golden_supervised.py): logits, cross-entropy loss and the
gradient of EVERY trained parameter -- the backward runs through the classifier head into the encoders and, unlike pretraining,
into the patch embedding -- plus the Mixup / CutMix kernel of the `fixed` augmentation pipeline against the reference's Mixup class."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import make_args, no_dropout, record_observed

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def build(cfg, model, ct, **over):
    from oracle.weights import fill_state_dict_
    c = no_dropout(cfg)
    for k, v in over.items():
        c[model][k] = v
    args = make_args(c, model, torch.device("cuda"), ct)
    args.train_mode, args.learn_framework = "supervised", "no"
    if model == "SW_Transformer":
        from models.SW_Transformer import SW_Transformer as Net
    else:
        from models.DeepSense import DeepSense as Net
    net = Net(args)
    fill_state_dict_(net.state_dict())
    return args, net.to("cuda")


@pytest.mark.parametrize("model", ["SW_Transformer", "DeepSense"])
@pytest.mark.parametrize("ct", ["fp32", "bf16"])
def test_supervised_step_matches_reference_fixture(cfg, model, ct):
    from models.loss import CrossEntropyLoss
    from oracle.weights import synthetic_freq_input
    fx = np.load(os.path.join(GOLD, f"supervised_{model}_b8.npz"))
    args, net = build(cfg, model, ct)
    net.train()
    x = {l: {m: v.cuda() for m, v in mm.items()} for l, mm in synthetic_freq_input(cfg, 8, seed=505).items()}
    labels = torch.from_numpy(fx["labels"]).cuda()
    net.arena().zero_grad()
    logits = net(x)
    ref = torch.from_numpy(fx["train.logits"])
    e = ((logits.detach().cpu() - ref).abs().max() / ref.abs().max()).item()
    record_observed(f"supervised.{model}.{ct}.logits.max_err_over_max_ref", e)
    assert e < (1e-3 if ct == "fp32" else 3e-2), e
    loss = CrossEntropyLoss()(logits, labels)
    assert abs(loss.item() - float(fx["train.loss"])) < (1e-3 if ct == "fp32" else 2e-2) * max(1.0, float(fx["train.loss"]))
    loss.backward()
    torch.cuda.synchronize()
    params = dict(net.named_parameters())
    names, norms = [str(n) for n in fx["train.grad_names"]], fx["train.grad_norms"]
    bad, worst = [], 0.0
    for n, r in zip(names, norms):
        g = params[n].grad
        assert g is not None, n
        got = g.double().norm().item()
        if n.endswith("conv.bias") and r < 1e-5:
            continue  # analytically zero in front of a train-mode BatchNorm
        rel = abs(got - r) / max(r, 1e-8)
        worst = max(worst, rel)
        if rel > (2e-3 if ct == "fp32" else 6e-2):
            bad.append((n, got, r))
        if ct == "fp32":
            flat = g.detach().reshape(-1).cpu().double()
            mine = flat[::max(1, flat.numel() // 16)][:16]
            sl = torch.from_numpy(fx[f"train.gradslice.{n}"])
            assert (mine - sl).abs().max().item() < 2e-3 * max(sl.abs().max().item(), r / max(flat.numel() ** 0.5, 1), 1e-7) + 1e-7, n
    record_observed(f"supervised.{model}.{ct}.grad_norm.worst_rel_err", worst)
    if ct == "fp32":
        assert not bad, bad[:6]
    else:
        assert len(bad) <= max(1, len(names) * 3 // 100), bad[:6]
    # exactly the reference's dead set has no gradient (torch's optimizers skip `grad is None`: no weight decay on them either)
    assert sorted(n for n, p in params.items() if p.grad is None) == sorted(str(n) for n in fx["train.dead_names"])
    assert any(n.startswith("patch_embed.") for n in names) == (model == "SW_Transformer")


def test_mixup_kernel_and_draw_order_match_reference_fixture(monkeypatch):
    """focal_mixup_fwd against the outputs of the reference's Mixup class (mode "random_batch") with forced draws, and
    `draw_mixup` consuming numpy / torch randomness in the reference's order (same boxes from the same forced stream)."""
    from data_augmenter.Augmenter import draw_mixup
    from focal_amd import ops
    fx = np.load(os.path.join(GOLD, "mixup_b6.npz"))
    g = torch.Generator().manual_seed(int(fx["seed"]))
    base = {"audio": torch.randn(6, 1, 10, 64, generator=g), "seismic": torch.randn(6, 1, 10, 20, generator=g)}
    mc = dict(mixup_alpha=1.0, cutmix_alpha=1.0, cutmix_minmax=None, prob=1.0, switch_prob=0.75, mode="random_batch", label_smoothing=0)
    for tag, cut in (("mixup", False), ("cutmix", True)):
        lam, perm = float(fx[f"{tag}.lam"]), [int(v) for v in fx[f"{tag}.perm"]]
        rands = iter([0.0, 0.0 if cut else 0.99])
        ints = iter([int(v) for m in ("audio", "seismic") for v in fx[f"{tag}.centre.{m}"]] if cut else [])
        monkeypatch.setattr(np.random, "rand", lambda *a: next(rands))
        monkeypatch.setattr(np.random, "beta", lambda a, b, size=None: lam)
        monkeypatch.setattr(np.random, "randint", lambda lo, hi=None, size=None: next(ints))
        monkeypatch.setattr(torch, "randperm", lambda n: torch.tensor(perm))
        d = draw_mixup(mc, {("shake", m): tuple(base[m].shape) for m in base})
        monkeypatch.undo()
        assert d is not None and d["cut"] == cut and abs(d["lam"] - lam) < 1e-12 and d["perm"].tolist() == perm
        for m in base:
            box = d["boxes"].get(("shake", m)) if cut else None
            if cut:
                assert list(box) == [int(v) for v in fx[f"{tag}.box.{m}"]], (m, box)
            got = ops.mixup(base[m].cuda(), d["perm"].to(torch.int32).cuda(), d["lam"], box)
            assert torch.allclose(got.cpu(), torch.from_numpy(fx[f"{tag}.out.{m}"]), rtol=0, atol=1e-6), (tag, m)


def test_nonlinear_class_head_matches_oracle(cfg):
    """`pretrained_head` != "linear": class layer Linear -> GELU -> Linear (models/SW_Transformer.py:175-181) in the finetune stage,
    logits and head gradients against the oracle's restatement (its linear-head path is pinned by the reference fixtures)."""
    from general_utils.weight_utils import set_learnable_params_finetune
    from models.loss import CrossEntropyLoss
    from models.SW_Transformer import SW_Transformer
    from oracle.finetune import finetune_loss_and_grads
    from oracle.weights import fill_state_dict_, synthetic_freq_input
    c = no_dropout(cfg)
    c["SW_Transformer"]["pretrained_head"] = "nonlinear"
    args = make_args(c, "SW_Transformer", torch.device("cuda"), "fp32")
    args.stage = "finetune"
    net = SW_Transformer(args)
    fill_state_dict_(net.state_dict())
    state = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.to("cuda").train()
    assert "class_layer.2.weight" in state
    set_learnable_params_finetune(args, net)
    xs = synthetic_freq_input(c, 8, seed=606)
    labels = torch.randint(0, c[args.task]["num_classes"], (8,), generator=torch.Generator().manual_seed(3))
    net.arena().zero_grad()
    logits = net({l: {m: v.cuda() for m, v in mm.items()} for l, mm in xs.items()}, class_head=True)
    CrossEntropyLoss()(logits, labels.cuda()).backward()
    torch.cuda.synchronize()
    r_logits, r_loss, r_grads = finetune_loss_and_grads("SW_Transformer", state, c, xs, labels, train=True)
    assert ((logits.detach().cpu() - r_logits).abs().max() / r_logits.abs().max()).item() < 1e-3
    params = dict(net.named_parameters())
    for n, g in r_grads.items():
        assert ((params[n].grad.cpu() - g).abs().max() / g.abs().max().clamp_min(1e-8)).item() < 2e-3, n


@pytest.mark.parametrize("model", ["SW_Transformer", "DeepSense"])
def test_train_py_supervised_stage_runs(model):
    src = os.path.join(ROOT, "focal_amd", "src")
    r = subprocess.run([sys.executable, os.path.join(src, "train.py"), f"-model={model}", "-dataset=MOD", "-learn_framework=no",
                        "-batch_size=16", "-synthetic_batches=3", "-epochs=2"], capture_output=True, text=True, timeout=900, cwd=src)
    log = r.stdout + r.stderr
    assert r.returncode == 0, log[-3000:]
    for needle in ("Training loss:", "Val acc:", "Test acc:", "Total processing time"):
        assert needle in log, (needle, log[-2000:])
    wdir = os.path.join(ROOT, "weights", f"MOD_{model}")
    assert os.path.exists(os.path.join(wdir, f"MOD_{model}_vehicle_classification_latest.pt"))
