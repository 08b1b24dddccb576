#!/usr/bin/env python3
"""Microbenchmark: the one-launch MLP forward at 128 / 256 channels (focal_mlp_wide_fwd) against the two launches it replaces, at the four
stage-1 / stage-2 shapes of the B = 256 MOD step; operand sets rotate through > 512 MB so every launch reads cold memory, as in the step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from focal_amd import ops  # noqa: E402
from focal_amd._lib import ACT_GELU, ACT_NONE, EPI_GELU, EPI_RESIDUAL  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def timeit(fn, n):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    rng = ops.new_rng_state(7, DEV)
    cc, f32 = ops.code(BF), ops.code(torch.float32)
    for C, M in ((128, 73728), (128, 36864), (256, 18432), (256, 9216)):
        H = 4 * C
        per_set = M * (C * 2 + C * 4 * 2 + H * 2 * 2)
        nsets = max(2, (640 << 20) // per_set + 1)
        sets = [dict(a=torch.randn(M, C, device=DEV).to(BF), r=torch.randn(M, C, device=DEV), y=torch.empty(M, C, device=DEV),
                     h=torch.empty(M, H, dtype=BF, device=DEV), hg=torch.empty(M, H, dtype=BF, device=DEV)) for _ in range(nsets)]
        w1, b1 = (torch.randn(H, C, device=DEV) * C ** -0.5).to(BF), torch.randn(H, device=DEV) * 0.1
        w2, b2 = (torch.randn(C, H, device=DEV) * H ** -0.5).to(BF), torch.randn(C, device=DEV) * 0.1
        gam, bet = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        dh, do = ops.drop_desc(rng, 17, 0.2, 21, 0.0, 1), ops.drop_desc(rng, 18, 0.2, 22, 0.1, 9)
        d1 = ops.linear_desc(cc, M, H, C, cc, cc, ACT_NONE, EPI_GELU, out_drop=dh)
        d2 = ops.linear_desc(cc, M, C, H, cc, f32, ACT_GELU, EPI_RESIDUAL, out_drop=do)
        dw = ops.mlp_desc(cc, M, C, H, dh, do)
        ln = C == 128
        k = [0]

        def two():
            s = sets[k[0] % nsets]
            k[0] += 1
            ops.linear_fwd(d1, s["a"], w1, b1, None, s["h"], s["hg"])
            if ln:
                ops.linear_resid_ln_fwd(d2, s["h"], w2, b2, s["r"], s["y"], gam, bet, BF)
            else:
                ops.linear_fwd(d2, s["h"], w2, b2, s["r"], s["y"])

        def one():
            s = sets[k[0] % nsets]
            k[0] += 1
            ops.mlp_wide_fwd(dw, s["a"], s["r"], w1, b1, w2, b2, s["y"], s["h"], s["hg"], next_ln=(gam, bet) if ln else None)

        def fc1():
            s = sets[k[0] % nsets]
            k[0] += 1
            ops.linear_fwd(d1, s["a"], w1, b1, None, s["h"], s["hg"])
        n = max(20, 2 * nsets)
        t2, t1, tf = timeit(two, n), timeit(one, n), timeit(fc1, n)
        wr = M * H * 2 * 2 + M * C * 4
        print(f"C {C:3d} M {M:6d}: two launches {t2:7.1f} us (fc1 alone {tf:6.1f})   one launch {t1:7.1f} us   x{t2 / t1:.2f}   "
              f"one launch: {wr / t1 / 1e6:.2f} TB/s of h + hg + y writes, {4.0 * M * C * H / t1 / 1e6:.0f} TFLOP/s")


if __name__ == "__main__":
    main()
