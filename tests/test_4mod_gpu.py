"""BASELINE.json configs[4]: 4-modality synthetic config (focal_amd/src/data/HAR4.yaml) through SW_Transformer + FOCAL,
against the reference fixture tests/golden/SW_Transformer_4mod_b8.npz (gen_golden.py ran the reference on this YAML)."""
import os

import numpy as np
import pytest
import torch

from conftest import no_dropout

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("ct", ["fp32", "bf16"])
def test_four_modality_step(ct):
    import argparse

    from models.FOCALModules import FOCAL
    from models.loss import FOCALLoss
    from models.SW_Transformer import SW_Transformer
    from oracle.config import load_config
    from oracle.weights import fill_state_dict_, synthetic_freq_input
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cfg = no_dropout(load_config(os.path.join(root, "focal_amd", "src", "data", "HAR4.yaml")))
    fx = np.load(os.path.join(GOLD, "SW_Transformer_4mod_b8.npz"))
    args = argparse.Namespace(model="SW_Transformer", dataset="HAR4", device=torch.device("cuda"), train_mode="contrastive",
                              learn_framework="FOCAL", stage="pretrain", task="activity_classification", tag=None,
                              dataset_config=cfg, compute_dtype=ct)
    net = SW_Transformer(args)
    fill_state_dict_(net.state_dict())
    net = net.to("cuda").train()
    focal, loss_fn = FOCAL(args, net), FOCALLoss(args)
    dev = lambda d: {l: {m: v.cuda() for m, v in mm.items()} for l, mm in d.items()}
    f1, f2 = focal(dev(synthetic_freq_input(cfg, 8, seed=303)), dev(synthetic_freq_input(cfg, 8, seed=404)), proj_head=True)
    assert list(f1.keys()) == ["acc", "gyr", "mag", "lig"]
    tol = 1e-3 if ct == "fp32" else 1.2e-2  # observed 7.5e-3 - 9.6e-3 (tests/golden/OBSERVED_r2.json)
    from conftest import record_observed
    for m in f1:
        ref = torch.from_numpy(fx[f"train.emb1.{m}"])
        e = ((f1[m].detach().cpu() - ref).abs().max() / ref.abs().max()).item()
        record_observed(f"swt4mod.train.emb.{m}.{ct}.max_err_over_max_ref", e)
        assert e < tol, m
    net.arena().zero_grad()
    loss = loss_fn(f1, f2)
    loss.backward()
    terms = loss_fn.last_terms.cpu().numpy()
    for i, k in enumerate(("shared", "private", "orth", "rank", "total")):
        ref = float(fx[f"train.loss.{k}"])
        record_observed(f"swt4mod.train.loss.{k}.{ct}.abs_err_over_max1", abs(terms[i] - ref) / max(1.0, abs(ref)))
        assert abs(terms[i] - ref) < (1e-3 if ct == "fp32" else 1e-2) * max(1.0, abs(ref)), (k, terms[i], ref)  # observed <= 2.6e-3
    params = dict(net.named_parameters())
    bad = []
    for n, ref in zip(fx["train.grad_names"], fx["train.grad_norms"]):
        got = params[str(n)].grad.double().norm().item()
        if abs(got - ref) > (2e-3 if ct == "fp32" else 6e-2) * max(ref, 1e-6) + 1e-6:
            bad.append((str(n), got, ref))
    if ct == "fp32":
        assert not bad, bad[:6]
    else:
        assert len(bad) <= max(1, len(fx["train.grad_names"]) * 3 // 100), bad[:6]
