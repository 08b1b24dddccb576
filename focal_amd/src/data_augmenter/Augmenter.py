"""View generation in front of the encoders (reference: data_augmenter/Augmenter.py:25-227).

`forward("random", time_loc_inputs)` draws ONE augmenter from the configured pool per call, flips a coin per
(location, modality) as the reference's augmenter classes do, and always ends with the time->frequency transform,
which runs on the HIP DFT kernel (`focal_fft_realpack_fwd`).  Augmentations are host-driven torch tensor edits on
device tensors -- they sit BEFORE the parity boundary and are stochastic / unseeded in the reference.
Deviation: `time_warp` / `mag_warp` wrap tsai's random-spline transforms, whose source is not available in this
build environment; they are applied as identity (their coin flip is still drawn) and logged once.
"""
import logging
import math
import os
import sys
from random import random

import numpy as np
import torch

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from focal_amd import ops  # noqa: E402


def _negation(x, cfg):  # NegationAugmenter.py:34
    return -x


def _scaling(x, cfg):  # ScalingAugmenter.py:35-36: one N(1, std) factor per (loc, mod)
    return x * float(np.random.normal(1.0, cfg["std"]))


def _horizontal_flip(x, cfg):  # HorizontalFlipAugmenter.py:34: flip intervals and samples
    return torch.flip(x, dims=[2, 3])


def _permutation(x, cfg):  # PermutationAugmenter.py:35-36: one random interval order for the whole batch
    return x[:, :, torch.randperm(x.shape[2], device=x.device), :]


_WARNED = set()


def _spline_warp_unavailable(name):
    def fn(x, cfg):
        if name not in _WARNED:
            _WARNED.add(name)
            logging.warning(f"{name}: tsai spline warp unavailable in this build; applied as identity")
        return x
    return fn


def _phase_shift(x, cfg):  # PhaseShiftAugmenter.py:39-54: rotate every complex bin by one random angle
    b, c, i, s = x.shape
    ang = (random() - 0.5) * 2 * math.pi
    z = x.reshape(b, c // 2, 2, i, s)
    re, im = z[:, :, 0], z[:, :, 1]
    ca, sa = math.cos(ang), math.sin(ang)
    return torch.stack([re * ca - im * sa, re * sa + im * ca], 2).reshape(b, c, i, s)


TIME_AUGMENTERS = {"no": None, "negation": _negation, "scaling": _scaling, "horizontal_flip": _horizontal_flip,
                   "permutation": _permutation, "time_warp": _spline_warp_unavailable("time_warp"),
                   "mag_warp": _spline_warp_unavailable("mag_warp")}
FREQ_AUGMENTERS = {"no": None, "phase_shift": _phase_shift}


class Augmenter:
    def __init__(self, args) -> None:
        self.args = args
        self.modalities = args.dataset_config["modality_names"]
        self.locations = args.dataset_config["location_names"]
        if args.train_mode != "supervised" and args.stage == "pretrain":
            pool = args.dataset_config[args.learn_framework]["random_augmenters"]
        else:
            pool = args.dataset_config[args.model]["fixed_augmenters"]
        self.time_aug_names, self.freq_aug_names = list(pool["time_augmenters"]), list(pool["freq_augmenters"])
        for n in self.time_aug_names:
            if n not in TIME_AUGMENTERS:
                raise Exception(f"Invalid augmenter provided: {n}")
        for n in self.freq_aug_names:
            if n not in FREQ_AUGMENTERS:
                raise Exception(f"Invalid augmenter provided: {n}")
        self.aug_names = self.time_aug_names + self.freq_aug_names

    def to(self, device):
        return self

    def forward(self, option, time_loc_inputs, labels=None, return_aug_id=False, return_aug_mods=False):
        time_loc_inputs, labels = self.move_to_target_device(time_loc_inputs, labels)
        if option == "random":
            out = self.forward_random(time_loc_inputs)
        elif option == "no":
            out = self.fft_preprocess(time_loc_inputs)
        else:
            raise Exception(f"Invalid augmentation option: {option}")
        return out if labels is None else (out, labels)

    def _apply(self, fn, name, inputs):
        if fn is None:
            return inputs
        out = {}
        for loc in self.locations:
            out[loc] = {}
            for mod in self.modalities:
                x = inputs[loc][mod]
                out[loc][mod] = fn(x, self.args.dataset_config.get(name, {})) if random() < self.args.dataset_config[name]["prob"] else x
        return out

    def forward_random(self, time_loc_inputs):
        k = np.random.randint(len(self.aug_names))
        name = self.aug_names[k]
        x = time_loc_inputs
        if k < len(self.time_aug_names):
            x = self._apply(TIME_AUGMENTERS[name], name, x)
        f = self.fft_preprocess(x)
        if k >= len(self.time_aug_names):
            f = self._apply(FREQ_AUGMENTERS[name], name, f)
        return f

    def move_to_target_device(self, time_loc_inputs, labels):
        dev = self.args.device
        out = {loc: {mod: t.float().to(dev) for mod, t in mods.items()} for loc, mods in time_loc_inputs.items()}
        return out, (labels.to(dev) if labels is not None else None)

    def fft_preprocess(self, time_loc_inputs):
        """[b, c, i, s] real -> [b, 2c, i, s] packed spectrum (reference :141-158), on the HIP DFT kernel."""
        return {loc: {mod: ops.fft_realpack(x.contiguous()) for mod, x in mods.items()}
                for loc, mods in time_loc_inputs.items()}
