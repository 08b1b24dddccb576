"""One FOCAL pretraining step on the CPU (test infrastructure; see oracle/__init__.py).

Restates train_utils/pretrain.py:62-74 with the parity-boundary conventions of SURVEY 8c: "no" augmentation
(FFT only), dropout rates 0.  Also the `cpu_baseline` ("port") leg of bench.py.
"""
import math

import torch

from .deepsense import deepsense_forward
from .loss import focal_loss_terms
from .swt import swt_forward


def fft_realpack(time_x):
    """Augmenter.fft_preprocess, data_augmenter/Augmenter.py:141-158: full two-sided complex FFT along the
    last axis of a real [b, c, i, s] tensor, packed as channels [c0.re, c0.im, c1.re, ...] -> [b, 2c, i, s]."""
    out = {}
    for loc in time_x:
        out[loc] = {}
        for mod, x in time_x[loc].items():
            f = torch.view_as_real(torch.fft.fft(x, dim=-1))  # [b, c, i, s, 2]
            b, c, i, s, _ = f.shape
            out[loc][mod] = f.permute(0, 1, 4, 2, 3).reshape(b, 2 * c, i, s)
    return out


def backbone_forward(model, P, cfg, freq_x, train=True, new_buffers=None):
    if model == "SW_Transformer":
        return swt_forward(P, cfg, freq_x, proj_head=True)
    if model == "DeepSense":
        return deepsense_forward(P, cfg, freq_x, proj_head=True, train=train, new_buffers=new_buffers)
    raise Exception(f"Invalid model provided: {model}")  # train_utils/model_selection.py:21


def pretrain_param_filter(model, key):
    """True for parameters that receive a gradient in FOCAL pretraining: everything except
    patch_embed.* (frozen, general_utils/weight_utils.py:85-94), class_layer, mod_fusion_layers,
    absolute_pos_embed (APE off) and DeepSense's mod_extractors (1 location)."""
    dead = ("patch_embed.", "class_layer.", "mod_fusion_layers.", "absolute_pos_embed.", "mod_extractors.")
    return not key.startswith(dead)


def cosine_lr(epoch, lr0, lr_min, t_initial):
    """timm CosineLRScheduler with cycle_limit=1, no warm-up (train_utils/lr_scheduler.py:21-34): value
    applied AFTER `.step(epoch)`; epochs beyond t_initial sit at lr_min."""
    if epoch >= t_initial:
        return lr_min
    return lr_min + 0.5 * (lr0 - lr_min) * (1 + math.cos(math.pi * epoch / t_initial))


def adamw_update(p, g, m, v, step, lr, wd=0.05, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.AdamW single-tensor semantics (train_utils/optimizer.py:27-32), in place; `step` is 1-based."""
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = (v.sqrt() / math.sqrt(1 - b2 ** step)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))


class OracleTrainer:
    """State-dict-in train loop equivalent to FOCAL + FOCALLoss + AdamW on the CPU."""

    def __init__(self, model, cfg, state, dtype=torch.float32):
        self.model, self.cfg = model, cfg
        self.P = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in state.items()}
        self.train_keys = [k for k, v in self.P.items() if v.is_floating_point() and pretrain_param_filter(model, k)
                           and not k.endswith(("running_mean", "running_var", "attn_mask"))]
        self.m = {k: torch.zeros_like(self.P[k]) for k in self.train_keys}
        self.v = {k: torch.zeros_like(self.P[k]) for k in self.train_keys}
        self.steps = 0
        opt = cfg["FOCAL"]["pretrain_optimizer"]
        self.lr, self.wd = opt["start_lr"], opt["weight_decay"]

    def loss_and_grads(self, freq1, freq2):
        for k in self.train_keys:
            self.P[k].requires_grad_(True)
        nb = {}
        f1 = backbone_forward(self.model, self.P, self.cfg, freq1, True, nb)
        self._apply_buffers(nb)
        nb = {}
        f2 = backbone_forward(self.model, self.P, self.cfg, freq2, True, nb)  # second call: second BN update
        self._apply_buffers(nb)
        terms = focal_loss_terms(f1, f2, self.cfg, self.model)
        grads = torch.autograd.grad(terms["total"], [self.P[k] for k in self.train_keys], allow_unused=True)
        for k in self.train_keys:
            self.P[k].requires_grad_(False)
        return terms, f1, f2, dict(zip(self.train_keys, grads))

    def _apply_buffers(self, nb):
        for k, v in nb.items():
            self.P[k] = v.detach()

    def step(self, time_x=None, freq_pair=None):
        """One optimiser step.  Either a time-domain batch ("no" augmentation twice -> identical views) or an
        explicit pair of frequency-domain views."""
        if freq_pair is None:
            f = fft_realpack(time_x)
            freq_pair = (f, f)
        terms, _, _, grads = self.loss_and_grads(*freq_pair)
        self.steps += 1
        with torch.no_grad():
            for k in self.train_keys:
                if grads[k] is None:
                    continue  # grad-less params are skipped entirely, as torch does
                adamw_update(self.P[k], grads[k], self.m[k], self.v[k], self.steps, self.lr, self.wd)
        return {k: float(v) for k, v in terms.items()}
