"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the FOCAL view augmentations that the MI355X path folds into the DFT (include/focal_hip.h:
focal_augment_fft_fwd).  The reference draws its randomness inside each augmenter (`random()`, `torch.normal`,
`torch.randperm`); here every draw is an explicit argument so that the arithmetic can be pinned
(tests/golden/gen_golden.py forces the same draws inside the reference classes and stores both sides)."""
import math

import torch

from .step import fft_realpack as _fft_dict


def fft_realpack(x):
    """Single-tensor form of oracle.step.fft_realpack (Augmenter.fft_preprocess, data_augmenter/Augmenter.py:141-158)."""
    return _fft_dict({"_": {"_": x}})["_"]["_"]


def negation(x):
    """NegationAugmenter.forward, data_augmenter/NegationAugmenter.py:34."""
    return -x


def scaling(x, factor):
    """ScalingAugmenter.forward, data_augmenter/ScalingAugmenter.py:35-36: one N(1, std) factor per (loc, mod)."""
    return x * factor


def horizontal_flip(x):
    """HorizontalFlipAugmenter.forward, data_augmenter/HorizontalFlipAugmenter.py:34: intervals and samples reversed."""
    return torch.flip(x, dims=[2, 3])


def permutation(x, order):
    """PermutationAugmenter.forward, data_augmenter/PermutationAugmenter.py:35-36: one interval order for the whole batch."""
    return x[:, :, torch.as_tensor(order, dtype=torch.long), :]


def phase_shift(f, angle):
    """PhaseShiftAugmenter.forward, data_augmenter/PhaseShiftAugmenter.py:39-54 on the packed [b, 2c, i, s] spectrum:
    |z| (cos(arg z + angle), sin(arg z + angle)) = z * e^{i angle}."""
    b, c, i, s = f.shape
    z = f.reshape(b, c // 2, 2, i, s)
    re, im = z[:, :, 0], z[:, :, 1]
    ca, sa = math.cos(angle), math.sin(angle)
    return torch.stack([re * ca - im * sa, re * sa + im * ca], 2).reshape(b, c, i, s)


def _tsai_curve(L, knots, order):
    """tsai 0.3.7 `random_curve_generator` (tsai/data/transforms.py) with the Gaussian knot values passed in: a cubic spline
    (scipy default end condition) through 3 (ord - 1) + 1 knots at linspace(-L, 2L - 1, ., dtype=int), evaluated on arange(L).
    tsai is a pinned dependency of the reference (requirements.txt:91) that is absent from /root/reference and from this image:
    restated from its published source -- PARITY UNPINNED for the two warps."""
    import numpy as np
    from scipy.interpolate import CubicSpline
    xs = np.linspace(-L, 2 * L - 1, 3 * (order - 1) + 1, dtype=int)
    return CubicSpline(xs, np.asarray(knots, np.float64), axis=-1)(np.arange(L))


def mag_warp(x, knots, order=4):
    """MagWarpAugmenter.forward (data_augmenter/MagWarpAugmenter.py:40-44) -> tsai TSMagWarp.encodes: o * curve on the [b, c, i*s]
    reshape, one curve for the whole tensor."""
    b, c, i, s = x.shape
    curve = torch.from_numpy(_tsai_curve(i * s, knots, order)).to(x.dtype)
    return (x.reshape(b, c, i * s) * curve).reshape(b, c, i, s)


def time_warp(x, knots, order=6):
    """TimeWarpAugmenter.forward (data_augmenter/TimeWarpAugmenter.py:40-44) -> tsai TSTimeWarp.encodes:
    CubicSpline(arange(L), o)(random_cum_curve_generator(o)) -- the exact (banded, not-a-knot) signal spline."""
    import numpy as np
    from scipy.interpolate import CubicSpline
    b, c, i, s = x.shape
    L = i * s
    cum = _tsai_curve(L, knots, order).cumsum()
    cum -= cum[0]
    cum /= cum[-1]
    pos = np.clip(cum, 0, 1) * (L - 1)
    f = CubicSpline(np.arange(L), x.reshape(b, c, L).double().numpy(), axis=-1)
    return torch.from_numpy(f(pos)).to(x.dtype).reshape(b, c, i, s)


def mixup_bbox(I, S, lam, cy, cx):
    """rand_bbox, input_utils/mixup_utils.py:32-54 (margin 0) with the two centre draws passed in: the box is cut from the LAST TWO
    dims of the [b, c, i, s] tensor, i.e. img_h = i (intervals), img_w = s (samples)."""
    import numpy as np
    ratio = np.sqrt(1 - lam)
    cut_h, cut_w = int(I * ratio), int(S * ratio)
    yl, yh = int(np.clip(cy - cut_h // 2, 0, I)), int(np.clip(cy + cut_h // 2, 0, I))
    xl, xh = int(np.clip(cx - cut_w // 2, 0, S)), int(np.clip(cx + cut_w // 2, 0, S))
    return yl, yh, xl, xh


def mixup_batch_random(x, perm, lam, box=None):
    """Mixup._mix_batch_random on one (loc, mod) tensor, input_utils/mixup_utils.py:252-281: `perm` is the call's single
    torch.randperm(b); box = (yl, yh, xl, xh) selects the CutMix branch (a paste from the permuted batch), else the lam blend."""
    perm = torch.as_tensor(perm, dtype=torch.long)
    if box is not None:
        yl, yh, xl, xh = box
        y = x.clone()
        y[:, :, yl:yh, xl:xh] = x[perm][:, :, yl:yh, xl:xh]
        return y
    return x * lam + x[perm] * (1.0 - lam)


def augmented_view(x, name, draw=None):
    """Augmenter.forward_random for ONE chosen augmenter applied to one (loc, mod) tensor (data_augmenter/Augmenter.py:76-113):
    time-domain augmenters act before the DFT, `phase_shift` after it.  `draw` = factor / order / angle where one is needed."""
    if name == "negation":
        x = negation(x)
    elif name == "scaling":
        x = scaling(x, draw)
    elif name == "horizontal_flip":
        x = horizontal_flip(x)
    elif name == "permutation":
        x = permutation(x, draw)
    elif name == "mag_warp":
        x = mag_warp(x, draw)
    elif name == "time_warp":
        x = time_warp(x, draw)
    elif name not in ("no", "phase_shift"):
        raise ValueError(name)
    f = fft_realpack(x)
    return phase_shift(f, draw) if name == "phase_shift" else f
