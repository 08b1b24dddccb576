#!/usr/bin/env python3
"""LayerNorm backward launches of the SW_Transformer step that are their own kernel (256 channels; the PatchMerging norms), HIP-event timed,
cold operands (rotated through > 600 MB).  python tools/mb_ln_bwd.py"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from focal_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
SHAPES = [(9216, 256), (4608, 256), (18432, 256), (36864, 128), (73728, 128), (18432, 512), (9216, 512)]


def main():
    rng = ops.new_rng_state(7, DEV)
    for M, C in SHAPES:
        per = M * C * (2 + 4 + 4 + 2)
        n = max(2, (600 << 20) // per + 1)
        sets = []
        for i in range(n):
            x = torch.randn(M, C, device=DEV)
            st = torch.stack([x.mean(1), (x.var(1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
            sets.append((torch.randn(M, C, device=DEV).to(BF), x, st, torch.randn(M, C, device=DEV), torch.empty(M, C, dtype=BF, device=DEV)))
        gam, dg, dbt = torch.randn(C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        mask = ops.drop_desc(rng, 5, 0.2, 9, 0.1, 64)

        def run(i):
            dy, x, st, g, gm = sets[i % n]
            ops.layernorm_bwd(dy, x, st, gam, g, True, dg, dbt, dx_masked=gm, mask=mask)

        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = max(40, 2 * n)
        e0.record()
        for i in range(it):
            run(i)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) / it * 1e3
        print(f"ln_bwd [{M:6d} x {C:3d}] {us:7.1f} us  {M * C * (2 + 4 + 8 + 2) / us / 1e3:6.0f} GB/s")
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
