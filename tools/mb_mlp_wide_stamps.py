#!/usr/bin/env python3
"""Lab: where the waves of focal_mlp_wide_fwd spend their cycles (a -DWIDE_STAMPS build: bash tools/build_variant.sh stamps "-DWIDE_STAMPS"
mlp_wide.hip; run with FOCAL_HIP_LIB=focal_amd/lab/libfocal_hip_stamps.so).  s_memtime ticks (100 MHz) summed per wave over the launch."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from focal_amd import _lib, ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def main():
    lib = _lib.load()
    rng = ops.new_rng_state(7, DEV)
    cc = ops.code(BF)
    for Cc, M in ((128, 73728), (128, 36864), (256, 18432), (256, 9216)):
        H = 4 * Cc
        a, r, y = torch.randn(M, Cc, device=DEV).to(BF), torch.randn(M, Cc, device=DEV), torch.empty(M, Cc, device=DEV)
        h, hg = torch.empty(M, H, dtype=BF, device=DEV), torch.empty(M, H, dtype=BF, device=DEV)
        w1, b1 = (torch.randn(H, Cc, device=DEV) * Cc ** -0.5).to(BF), torch.randn(H, device=DEV) * 0.1
        w2, b2 = (torch.randn(Cc, H, device=DEV) * H ** -0.5).to(BF), torch.randn(Cc, device=DEV) * 0.1
        dh, do = ops.drop_desc(rng, 17, 0.2, 21, 0.0, 1), ops.drop_desc(rng, 18, 0.2, 22, 0.1, 9)
        d = ops.mlp_desc(cc, M, Cc, H, dh, do)
        stamps = torch.zeros(512 * 16 * 8, dtype=torch.int64, device=DEV)
        p = ops._p
        for _ in range(3):
            _lib.check(lib.focal_mlp_wide_fwd(C.byref(d), p(a), p(r), p(w1), p(b1), p(w2), p(b2), p(y), p(h), p(hg), None, None, None, p(stamps), ops._stream()))
        torch.cuda.synchronize()
        s = stamps.view(512, 16, 8).cpu()
        cons = s[s[:, :, 4] == 2].float()
        load = s[s[:, :, 4] == 1].float()
        f = lambda t, i: f"{t[:, i].mean().item() / 100:.1f}"
        print(f"C {Cc} M {M}: {cons.shape[0]} consumer waves: us per wave: barrier {f(cons, 0)} fc1 {f(cons, 1)} gelu+stores {f(cons, 2)} fc2 {f(cons, 3)} epilogue {f(cons, 5)} all {f(cons, 6)}"
              f" | {load.shape[0]} loader waves: land-wait {f(load, 0)} barrier {f(load, 1)} issue {f(load, 2)} all {f(load, 3)}")


if __name__ == "__main__":
    main()
