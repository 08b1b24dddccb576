"""Host logic of the TimeWarp / MagWarp augmenters (focal_amd/warp.py) against the oracle's restatement of tsai's algorithm with
scipy (oracle/augment.py): the random curve, the warped positions and the cardinal-spline tables the device kernel consumes."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from focal_amd import warp  # noqa: E402
from oracle import augment  # noqa: E402


def _apply_tables(x, k0, w):
    L = x.shape[-1]
    idx = np.clip(k0[:, None] + np.arange(w.shape[1])[None, :], 0, L - 1)
    return (x[..., idx] * w).sum(-1)


@pytest.mark.parametrize("L,order,mag", [(16000, 4, 0.05), (200, 4, 0.05), (16000, 6, 0.2), (200, 6, 0.2)])
def test_random_curve_is_scipys_cubic_spline_through_the_knots(L, order, mag):
    kn = warp.draw_knots(order, mag, np.random.RandomState(L + order))
    assert len(kn) == 3 * (order - 1) + 1
    assert np.abs(warp.random_curve(L, kn, order) - augment._tsai_curve(L, kn, order)).max() < 1e-12


@pytest.mark.parametrize("I,S", [(10, 1600), (10, 20)])
def test_time_warp_tables_evaluate_the_signal_spline(I, S):
    L, order = I * S, 6
    kn = warp.draw_knots(order, 0.2, np.random.RandomState(5))
    pos = warp.warp_positions(L, kn, order)
    assert pos[0] == 0 and abs(pos[-1] - (L - 1)) < 1e-9 and (np.diff(pos) > 0).all()
    assert np.abs(pos - np.arange(L)).max() > 0.01 * L          # a real warp, not the identity
    k0, w = warp.time_warp_tables(pos)
    assert k0.dtype == np.int32 and w.shape == (L, warp.TAPS) and w.dtype == np.float32
    assert np.abs(w.sum(1) - 1).max() < 1e-5                    # partition of unity: constants are reproduced
    x = torch.randn(3, 2, I, S, generator=torch.Generator().manual_seed(1))
    ref = augment.time_warp(x, kn, order).reshape(3, 2, L).numpy()
    got = _apply_tables(x.reshape(3, 2, L).numpy(), k0, w)
    scale = np.abs(ref).max()
    err = np.abs(got - ref)
    assert err.max() < 2e-5 * scale                             # 24-tap filter == scipy's banded not-a-knot solve, ends included


def test_unit_knots_are_the_identity():
    L, order = 200, 6
    ones = np.ones(3 * (order - 1) + 1)
    assert np.abs(warp.random_curve(L, ones, 4 if False else order) - 1).max() < 1e-12
    pos = warp.warp_positions(L, ones, order)
    assert np.abs(pos - np.arange(L)).max() < 1e-9
    k0, w = warp.time_warp_tables(pos)
    x = np.random.RandomState(0).randn(4, L)
    assert np.abs(_apply_tables(x, k0, w) - x).max() < 1e-5
    xt = torch.from_numpy(x).reshape(4, 1, 10, 20)
    assert (augment.mag_warp(xt, np.ones(10), 4) - xt).abs().max() < 1e-12
    assert (augment.time_warp(xt, ones, order) - xt).abs().max() < 1e-9
