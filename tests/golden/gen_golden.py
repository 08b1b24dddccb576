#!/usr/bin/env python3
"""Generate the committed parity fixtures by IMPORTING THE REFERENCE in the build container.

Run here only (needs /root/reference; never on the GPU box):   python tests/golden/gen_golden.py

What it does
  1. Puts minimal stand-ins for the two third-party packages the image lacks on sys.path, in a temp dir
     (nothing from the reference is copied, and the stand-ins never enter the repo's runtime path):
       timm.models.layers.trunc_normal_  -> torch.nn.init.trunc_normal_   (init only; fixtures overwrite weights)
       timm.models.layers.DropPath       -> canonical per-sample stochastic depth (identity at p=0 / eval)
       timm.scheduler.{cosine_lr,step_lr} -> not exercised (we only build models, the loss and AdamW)
       tsai.data.{transforms,core}       -> identity placeholders so `data_augmenter.Augmenter` imports;
                                            the parity boundary is after augmentation
  2. Builds the reference DeepSense / SW_Transformer / FOCAL / FOCALLoss from the reference's own MOD.yaml with all
     dropout / drop-path rates overridden to 0, fills every floating-point state-dict entry with the name-seeded
     values of oracle/weights.py, runs seeded synthetic inputs, and records outputs as small .npz fixtures.
  3. Cross-checks the oracle against the reference on the same inputs and refuses to write fixtures on mismatch.
"""
import argparse
import copy
import json
import os
import sys
import tempfile

import numpy as np
import torch
import yaml

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))

STUBS = {
    "timm/__init__.py": "",
    "timm/models/__init__.py": "",
    "timm/models/layers.py": (
        "import torch\nimport torch.nn as nn\n"
        "def trunc_normal_(t, mean=0., std=1., a=-2., b=2.):\n    return nn.init.trunc_normal_(t, mean, std, a, b)\n"
        "class DropPath(nn.Module):\n"
        "    def __init__(self, drop_prob=0.):\n        super().__init__(); self.drop_prob = drop_prob\n"
        "    def forward(self, x):\n"
        "        if self.drop_prob == 0. or not self.training: return x\n"
        "        keep = 1 - self.drop_prob\n"
        "        m = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)\n"
        "        return x * m / keep\n"),
    "timm/scheduler/__init__.py": "",
    "timm/scheduler/cosine_lr.py": "class CosineLRScheduler:\n    def __init__(self, *a, **k): pass\n    def step(self, e): pass\n",
    "timm/scheduler/step_lr.py": "class StepLRScheduler:\n    def __init__(self, *a, **k): pass\n    def step(self, e): pass\n",
    "tsai/__init__.py": "",
    "tsai/data/__init__.py": "",
    "tsai/data/transforms.py": ("class _Id:\n    def __init__(self, *a, **k): pass\n    def __call__(self, x, **k): return x\n"
                                "TSTimeWarp = _Id\nTSMagWarp = _Id\n"),
    "tsai/data/core.py": "def TSTensor(x):\n    return x\n",
}


def install_reference():
    d = tempfile.mkdtemp(prefix="focal_ref_stubs_")
    for rel, body in STUBS.items():
        p = os.path.join(d, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(body)
    sys.path.insert(0, REF)
    sys.path.insert(0, d)
    sys.path.insert(0, REPO)


def ref_args(model, cfg):
    return argparse.Namespace(model=model, dataset="MOD", device=torch.device("cpu"), train_mode="contrastive",
                              learn_framework="FOCAL", stage="pretrain", task="vehicle_classification", tag=None,
                              dataset_config=cfg)


def no_dropout(cfg):
    cfg = copy.deepcopy(cfg)
    cfg["DeepSense"]["dropout_ratio"] = 0.0
    cfg["SW_Transformer"]["dropout_ratio"] = 0.0
    cfg["SW_Transformer"]["drop_path_rate"] = 0.0
    cfg["SW_Transformer"]["attn_drop_rate"] = 0.0
    return cfg


def four_mod_cfg(cfg):
    """The 4-modality synthetic config of BASELINE.json configs[4] (authored by this build)."""
    with open(os.path.join(REPO, "focal_amd", "src", "data", "HAR4.yaml")) as f:
        return yaml.safe_load(f)


def sub(t, n=64):
    """Strided sub-sample of a tensor as float64 numpy (keeps fixtures small)."""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].double().numpy()


def main():
    install_reference()
    from oracle import weights as ow
    from oracle.config import load_config
    from oracle.deepsense import deepsense_forward
    from oracle.loss import focal_loss_terms
    from oracle.step import OracleTrainer, fft_realpack
    from oracle.swt import swt_forward

    from models.DeepSense import DeepSense
    from models.FOCALModules import FOCAL
    from models.loss import FOCALLoss
    from models.SW_Transformer import SW_Transformer
    from data_augmenter.Augmenter import Augmenter
    from general_utils.weight_utils import freeze_patch_embedding

    torch.manual_seed(0)
    ref_cfg = yaml.safe_load(open(os.path.join(REF, "data", "MOD.yaml")))
    cfg = no_dropout(ref_cfg)
    my_cfg = load_config()
    summary = {}

    # ---------------------------------------------------------------- manifests (key names + shapes)
    for model, cls, spec_fn in (("SW_Transformer", SW_Transformer, ow.swt_state_spec),
                                ("DeepSense", DeepSense, ow.deepsense_state_spec)):
        net = cls(ref_args(model, cfg))
        sd = net.state_dict()
        manifest = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]
        spec = spec_fn(my_cfg)
        assert [m[0] for m in manifest] == list(spec.keys()), f"{model}: oracle key list differs from reference"
        for k, shp, _ in manifest:
            assert tuple(shp) == tuple(spec[k]), (k, shp, spec[k])
        with open(os.path.join(OUT, f"manifest_{model}.json"), "w") as f:
            json.dump(manifest, f, indent=0)
        summary[f"{model}_keys"] = len(manifest)
        summary[f"{model}_params"] = int(sum(p.numel() for p in net.parameters()))

    # ---------------------------------------------------------------- FFT pack (row 3)
    aug = Augmenter(ref_args("SW_Transformer", ref_cfg))
    tx = ow.synthetic_time_input(my_cfg, 4, seed=11)
    ref_f = aug.forward("no", {l: {m: v.clone() for m, v in d.items()} for l, d in tx.items()})
    my_f = fft_realpack(tx)
    fft_fix = {}
    for l in ref_f:
        for m in ref_f[l]:
            assert torch.equal(ref_f[l][m], my_f[l][m]), "fft_realpack differs from reference"
            fft_fix[f"{l}.{m}"] = ref_f[l][m].numpy()
    np.savez_compressed(os.path.join(OUT, "fft_b4_seed11.npz"), **fft_fix)

    # ---------------------------------------------------------------- backbones + loss + grads
    for model, cls, fwd in (("SW_Transformer", SW_Transformer, swt_forward), ("DeepSense", DeepSense, deepsense_forward)):
        args = ref_args(model, cfg)
        B = 8
        net = cls(args)
        ow.fill_state_dict_(net.state_dict())
        state0 = {k: v.clone() for k, v in net.state_dict().items()}
        x1 = ow.synthetic_freq_input(my_cfg, B, seed=101)
        x2 = ow.synthetic_freq_input(my_cfg, B, seed=202)
        fix = {}

        # eval mode (deterministic; BN running stats)
        net.eval()
        with torch.no_grad():
            ref_eval = net(x1, class_head=False, proj_head=True)
            ref_feat = net(x1, class_head=False, proj_head=False)
        taps = {}
        with torch.no_grad():
            if model == "DeepSense":
                mine = fwd(state0, my_cfg, x1, proj_head=True, train=False, taps=taps)
            else:
                mine = fwd(state0, my_cfg, x1, proj_head=True, taps=taps)
        for m in ref_eval:
            err = (ref_eval[m] - mine[m]).abs().max().item()
            assert err < 2e-5 * max(1.0, ref_eval[m].abs().max().item()), (model, m, "eval", err)
            fix[f"eval.emb.{m}"] = ref_eval[m].numpy()
            fix[f"eval.feat.{m}"] = ref_feat[m].numpy()
        for k, v in taps.items():
            fix[f"eval.tap.{k}"] = sub(v, 256)
            fix[f"eval.tapnorm.{k}"] = np.array([v.double().norm().item(), v.double().abs().max().item()])

        # train mode, dropout 0: FOCAL(view1, view2) -> loss -> backward (pretrain.py:62-70)
        net.train()
        focal = FOCAL(args, net)
        focal = freeze_patch_embedding(args, focal)
        loss_fn = FOCALLoss(args)
        f1, f2 = focal(x1, x2, proj_head=True)
        for d in (f1, f2):
            for m in d:
                d[m].retain_grad()
        loss = loss_fn(f1, f2)
        loss.backward()
        tr = OracleTrainer(model, my_cfg, state0)
        terms, o1, o2, grads = tr.loss_and_grads(x1, x2)
        assert abs(float(terms["total"]) - float(loss)) < 1e-4 * max(1.0, abs(float(loss))), (float(terms["total"]), float(loss))
        for m in f1:
            assert (f1[m] - o1[m]).abs().max().item() < 1e-4, (model, m, "train view1")
            fix[f"train.emb1.{m}"] = f1[m].detach().numpy()
            fix[f"train.emb2.{m}"] = f2[m].detach().numpy()
            fix[f"train.demb1.{m}"] = f1[m].grad.numpy()
            fix[f"train.demb2.{m}"] = f2[m].grad.numpy()
        for k in ("shared", "private", "orth", "rank", "total"):
            fix[f"train.loss.{k}"] = np.array(float(terms[k]))
        fix["train.loss.reference_total"] = np.array(float(loss))
        names, norms = [], []
        for k, p in net.named_parameters():
            if p.grad is None:
                continue
            assert k in grads, f"reference has a gradient for {k} that the oracle treats as dead"
            # conv biases feeding a train-mode BatchNorm have an analytically zero gradient (pure rounding noise)
            gerr = (p.grad - grads[k]).norm().item()
            assert gerr < 5e-4 * p.grad.norm().item() + 1e-5, (model, k, gerr, p.grad.norm().item())
            names.append(k)
            norms.append(p.grad.double().norm().item())
            fix[f"train.gradslice.{k}"] = sub(p.grad, 16)
        dead = [k for k, p in net.named_parameters() if p.grad is None]
        assert sorted(dead) == sorted(k for k, _ in net.named_parameters() if k not in tr.train_keys), "dead-param set differs"
        fix["train.grad_names"] = np.array(names)
        fix["train.grad_norms"] = np.array(norms)
        if model == "DeepSense":
            for k, v in net.state_dict().items():
                if k.endswith(("running_mean", "running_var")) and k.startswith("loc_mod_extractors"):
                    assert (v - tr.P[k]).abs().max().item() < 1e-5 * max(1.0, v.abs().max().item()), k
                    fix[f"train.buf.{k}"] = v.numpy()

        # three AdamW steps on a fixed batch (rows 1, 2, 14)
        net2 = cls(args)
        net2.load_state_dict(state0)
        net2.train()
        focal2 = FOCAL(args, net2)
        from torch import optim
        oc = cfg["FOCAL"]["pretrain_optimizer"]
        opt = optim.AdamW(focal2.parameters(), lr=oc["start_lr"], weight_decay=oc["weight_decay"])
        focal2 = freeze_patch_embedding(args, focal2)
        tr2 = OracleTrainer(model, my_cfg, state0)
        traj = []
        for it in range(3):
            opt.zero_grad()
            a, b = focal2(x1, x2, proj_head=True)
            l = loss_fn(a, b)
            l.backward()
            opt.step()
            mine_terms = tr2.step(freq_pair=(x1, x2))
            assert abs(mine_terms["total"] - float(l)) < 2e-3 * max(1.0, abs(float(l))), (it, mine_terms["total"], float(l))
            traj.append(float(l))
        fix["adamw.loss_traj"] = np.array(traj)
        probe = "mod_projectors.audio.2.weight"
        fix["adamw.probe_after3"] = sub(dict(net2.named_parameters())[probe], 32)
        np.savez_compressed(os.path.join(OUT, f"{model}_b8.npz"), **fix)
        summary[f"{model}_loss"] = float(loss)
        summary[f"{model}_traj"] = traj

    # ---------------------------------------------------------------- loss head alone (rows 11-13)
    def loss_fixture(name, lcfg, model, B, seed, scale):
        args = ref_args(model, lcfg)
        mods = lcfg["modality_names"]
        g = torch.Generator().manual_seed(seed)
        f1 = {m: (torch.randn(B, 256, generator=g) * scale).requires_grad_(True) for m in mods}
        f2 = {m: (torch.randn(B, 256, generator=g) * scale).requires_grad_(True) for m in mods}
        # make the views correlated so that positives matter
        with torch.no_grad():
            for m in mods:
                f2[m].mul_(0.5).add_(0.5 * f1[m])
        loss = FOCALLoss(args)(f1, f2)
        loss.backward()
        mine = focal_loss_terms({m: v.detach() for m, v in f1.items()}, {m: v.detach() for m, v in f2.items()}, lcfg, model)
        assert abs(float(mine["total"]) - float(loss)) < 2e-5 * max(1.0, abs(float(loss))), (name, float(mine["total"]), float(loss))
        fix = {"B": np.array(B), "seed": np.array(seed), "scale": np.array(scale), "mods": np.array(mods)}
        for k in ("shared", "private", "orth", "rank", "total"):
            fix[f"loss.{k}"] = np.array(float(mine[k]))
        fix["loss.reference_total"] = np.array(float(loss))
        for m in mods:
            fix[f"demb1.{m}"] = f1[m].grad.numpy().astype(np.float32)
            fix[f"demb2.{m}"] = f2[m].grad.numpy().astype(np.float32)
        np.savez_compressed(os.path.join(OUT, f"loss_{name}.npz"), **fix)
        summary[f"loss_{name}"] = float(loss)

    loss_fixture("swt_b32", cfg, "SW_Transformer", 32, 7, 1.5)
    loss_fixture("ds_b32", cfg, "DeepSense", 32, 8, 0.25)
    loss_fixture("swt_b256", cfg, "SW_Transformer", 256, 9, 1.5)
    try:
        c4 = no_dropout(four_mod_cfg(cfg))
        loss_fixture("swt_4mod_b32", c4, "SW_Transformer", 32, 10, 1.5)
    except FileNotFoundError:
        print("HAR4.yaml not authored yet; skipping 4-modality loss fixture")

    # ---------------------------------------------------------------- 4-modality SW_Transformer (BASELINE configs[4])
    c4_ref = four_mod_cfg(cfg)
    c4 = no_dropout(c4_ref)
    args4 = ref_args("SW_Transformer", c4)
    args4.dataset, args4.task = "HAR4", "activity_classification"
    net4 = SW_Transformer(args4)
    ow.fill_state_dict_(net4.state_dict())
    st4 = {k: v.clone() for k, v in net4.state_dict().items()}
    spec4 = ow.swt_state_spec(c4, task="activity_classification")
    assert list(spec4.keys()) == list(st4.keys())
    y1, y2 = ow.synthetic_freq_input(c4, 8, seed=303), ow.synthetic_freq_input(c4, 8, seed=404)
    net4.train()
    focal4 = freeze_patch_embedding(args4, FOCAL(args4, net4))
    g1, g2 = focal4(y1, y2, proj_head=True)
    loss4 = FOCALLoss(args4)(g1, g2)
    loss4.backward()
    tr4 = OracleTrainer("SW_Transformer", c4, st4)
    t4, o1, o2, gr4 = tr4.loss_and_grads(y1, y2)
    assert abs(float(t4["total"]) - float(loss4)) < 1e-4 * abs(float(loss4)), (float(t4["total"]), float(loss4))
    fix4 = {"train.loss.reference_total": np.array(float(loss4))}
    for k in ("shared", "private", "orth", "rank", "total"):
        fix4[f"train.loss.{k}"] = np.array(float(t4[k]))
    for m in g1:
        assert (g1[m] - o1[m]).abs().max().item() < 1e-4
        fix4[f"train.emb1.{m}"] = g1[m].detach().numpy()
        fix4[f"train.emb2.{m}"] = g2[m].detach().numpy()
    names4, norms4 = [], []
    for k, p_ in net4.named_parameters():
        if p_.grad is None:
            continue
        gerr = (p_.grad - gr4[k]).norm().item()
        assert gerr < 5e-4 * p_.grad.norm().item() + 1e-5, (k, gerr)
        names4.append(k)
        norms4.append(p_.grad.double().norm().item())
    fix4["train.grad_names"] = np.array(names4)
    fix4["train.grad_norms"] = np.array(norms4)
    np.savez_compressed(os.path.join(OUT, "SW_Transformer_4mod_b8.npz"), **fix4)
    summary["SW_Transformer_4mod_loss"] = float(loss4)
    summary["SW_Transformer_4mod_params"] = int(sum(p_.numel() for p_ in net4.parameters()))

    with open(os.path.join(OUT, "SUMMARY.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
