#!/usr/bin/env python3
"""The [1, k] convolutions of the DeepSense step (B = 256, both views as one batch: 102 400 token rows of 64 channels, intervals of 20 tokens;
k = 5 audio, 3 seismic) and the BatchNorm launches around them, HIP-event timed, cold operands (rotated through > 600 MB).
python tools/mb_conv.py"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from focal_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
ROWS, S, C = 102400, 20, 64


def timeit(fn, n):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = max(40, 2 * n)
    e0.record()
    for i in range(it):
        fn(i)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def main():
    cc = ops.code(BF)
    n = (600 << 20) // (ROWS * C * (2 + 4 + 4)) + 2
    xs = [torch.randn(ROWS, C, device=DEV).to(BF) for _ in range(n)]
    gs = [torch.randn(ROWS, C, device=DEV) for _ in range(n)]
    outs = [torch.empty(ROWS, C, device=DEV) for _ in range(n)]
    for k in (5, 3):
        d = ops.conv_desc(cc, ROWS, S, C, C, k)
        w = torch.randn(C, C, 1, k, device=DEV) * (C * k) ** -0.5
        w_fwd = ops.permute_pack(w, C, C, k, BF)
        w_bwd = ops.conv_pack_bwd(d, w, BF)
        bias = torch.randn(C, device=DEV) * 0.1
        d_bn = ops.bn_desc(cc, ROWS, C, 10 * S, groups=2)
        rm, rv = torch.zeros(2, C, device=DEV), torch.ones(2, C, device=DEV)
        us = timeit(lambda i: ops.conv_fwd(d, xs[i % n], w_fwd, bias), n)
        print(f"k={k} conv_fwd              {us:7.1f} us  {ROWS * C * 6 / us / 1e3:6.0f} GB/s")
        us = timeit(lambda i: ops.conv_fwd_bn(d, xs[i % n], w_fwd, bias, d_bn, rm, rv), n)
        print(f"k={k} conv_fwd_bn (2 groups) {us:7.1f} us  {ROWS * C * 6 / us / 1e3:6.0f} GB/s")
        us = timeit(lambda i: ops.conv_bwd_data(d, xs[i % n], w_bwd, gs[i % n], outs[i % n]), n)
        print(f"k={k} conv_bwd_data         {us:7.1f} us  {ROWS * C * 10 / us / 1e3:6.0f} GB/s")
        dw, db = torch.zeros(C, k * C, device=DEV), torch.zeros(C, device=DEV)
        us = timeit(lambda i: ops.conv_bwd_weight(d, xs[i % n], xs[(i + 1) % n], dw, db), n)
        print(f"k={k} conv_bwd_weight       {us:7.1f} us  {ROWS * C * 4 / us / 1e3:6.0f} GB/s")


if __name__ == "__main__":
    main()
