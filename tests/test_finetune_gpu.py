"""Classifier / finetune path against fixtures produced by the reference models
(tests/golden/finetune_*_b8.npz, tests/golden/gen_golden_finetune.py): `backbone(freq_x, class_head=True)` logits, the
cross-entropy loss, and the gradients of exactly the parameters finetuning trains."""
import os

import numpy as np
import pytest
import torch

from conftest import make_args, no_dropout

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def build(cfg, model, ct):
    from oracle.weights import fill_state_dict_
    args = make_args(no_dropout(cfg), model, torch.device("cuda"), ct)
    args.stage = "finetune"
    if model == "SW_Transformer":
        from models.SW_Transformer import SW_Transformer as Net
    else:
        from models.DeepSense import DeepSense as Net
    net = Net(args)
    fill_state_dict_(net.state_dict())
    return args, net.to("cuda")


@pytest.mark.parametrize("model", ["SW_Transformer", "DeepSense"])
@pytest.mark.parametrize("ct", ["fp32", "bf16"])
def test_classifier_logits_loss_and_head_gradients(cfg, model, ct):
    from general_utils.weight_utils import set_learnable_params_finetune
    from models.loss import CrossEntropyLoss
    from oracle.weights import synthetic_freq_input
    fx = np.load(os.path.join(GOLD, f"finetune_{model}_b8.npz"))
    args, net = build(cfg, model, ct)
    x = {l: {m: v.cuda() for m, v in mm.items()} for l, mm in synthetic_freq_input(cfg, 8, seed=303).items()}
    labels = torch.from_numpy(fx["labels"]).cuda()
    tol = 1e-3 if ct == "fp32" else 3e-2
    net.eval()
    with torch.no_grad():
        logits = net(x, class_head=True)
    ref = torch.from_numpy(fx["eval.logits"])
    assert ((logits.cpu() - ref).abs().max() / ref.abs().max()).item() < tol
    net.train()
    learnable = set_learnable_params_finetune(args, net)
    names = [str(n) for n in fx["train.grad_names"]]
    assert sorted(n for n, p in net.named_parameters() if p.requires_grad) == sorted(names)
    net.arena().zero_grad()
    logits = net(x, class_head=True)
    ref = torch.from_numpy(fx["train.logits"])
    assert ((logits.detach().cpu() - ref).abs().max() / ref.abs().max()).item() < tol
    loss = CrossEntropyLoss()(logits, labels)
    assert abs(loss.item() - float(fx["train.loss"])) < tol * max(1.0, float(fx["train.loss"]))
    loss.backward()
    torch.cuda.synchronize()
    params = dict(net.named_parameters())
    for n in names:
        g, r = params[n].grad.cpu(), torch.from_numpy(fx[f"train.grad.{n}"])
        assert ((g - r).abs().max() / r.abs().max().clamp_min(1e-8)).item() < (2e-3 if ct == "fp32" else 6e-2), n
    # one optimizer step moves the head and nothing else (the frozen encoder must not even see weight decay)
    from train_utils.optimizer import define_optimizer
    before = {k: v.detach().clone() for k, v in net.state_dict().items()}
    opt = define_optimizer(args, learnable)
    opt.step()
    torch.cuda.synchronize()
    after = net.state_dict()
    for k in before:
        moved = not torch.equal(before[k], after[k])
        assert moved == (k in names), k
