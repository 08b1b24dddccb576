#!/usr/bin/env python3
"""dX of a linear layer + the backward of the LayerNorm in front of it at 256 channels: one launch (focal_linear_bwd_data_ln) against the
two launches the step uses (ring GEMM + ln_bwd_kernel), cold operands.  Run under rocprofv3 for kernel durations (tools/prof_ln_fused.sh):
the Python loop is host-bound below ~20 us per call."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from focal_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
SHAPES = [(9216, 768, 256), (9216, 1024, 256), (4608, 768, 256), (4608, 1024, 256)]


def main():
    cc, f32 = ops.code(BF), ops.code(torch.float32)
    rng = ops.new_rng_state(11, DEV)
    mask = ops.drop_desc(rng, 5, 0.2, 9, 0.1, 64)
    for M, N, K in SHAPES:
        n = max(2, (500 << 20) // (M * (N * 2 + K * 14)) + 1)
        sets = []
        for i in range(n):
            x = torch.randn(M, K, device=DEV)
            st = torch.stack([x.mean(1), (x.var(1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
            sets.append((torch.randn(M, N, device=DEV).to(BF), x, st, torch.randn(M, K, device=DEV), torch.empty(M, K, dtype=BF, device=DEV), torch.empty(M, K, dtype=BF, device=DEV)))
        w = (torch.randn(N, K, device=DEV) * K ** -0.5).to(BF)
        gam, dg, dbt = torch.randn(K, device=DEV), torch.zeros(K, device=DEV), torch.zeros(K, device=DEV)
        d = ops.linear_desc(cc, M, N, K, cc, cc)
        for it in range(24):
            dy, x, st, g, gm, da = sets[it % n]
            ops.linear_bwd_data_ln(d, dy, w, x, st, gam, g, dg, dbt, g_masked=gm, mask=mask)
        for it in range(24):
            dy, x, st, g, gm, da = sets[it % n]
            ops.linear_bwd_data(d, dy, w, None, da)
            ops.layernorm_bwd(da, x, st, gam, g, True, dg, dbt, dx_masked=gm, mask=mask)
        torch.cuda.synchronize()
        print("done", M, N, K)
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
