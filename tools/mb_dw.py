#!/usr/bin/env python3
"""Weight-gradient GEMM instances of the SW_Transformer step (B = 256, both views in one pass), HIP-event timed, cold operands
(rotated through > 600 MB).  python tools/mb_dw.py   (launch-plan knobs: FOCAL_DW_WIDE_MIN, FOCAL_DW_WGS, FOCAL_DW_MIN_ROWS)"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from focal_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
SHAPES = [(18432, 1024, 256), (18432, 256, 1024), (9216, 1024, 256), (9216, 256, 1024), (73728, 512, 128), (73728, 128, 512),
          (18432, 256, 256), (18432, 768, 256), (73728, 384, 128), (73728, 128, 128), (294912, 192, 64), (294912, 64, 64)]


if os.environ.get("FOCAL_MB_DW_SHAPES"):  # "rows,N,K;rows,N,K;..."
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["FOCAL_MB_DW_SHAPES"].split(";")]


def main():
    cc = ops.code(BF)
    tot = 0.0
    for M, N, K in SHAPES:
        n = max(2, (600 << 20) // (M * (N + K) * 2) + 1)
        sets = [(torch.randn(M, N, device=DEV).to(BF), torch.randn(M, K, device=DEV).to(BF)) for _ in range(n)]
        dw, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
        d = ops.linear_desc(cc, M, N, K, cc, cc)
        for i in range(3):
            ops.linear_bwd_weight(d, sets[i % n][0], sets[i % n][1], dw, db)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = max(20, 2 * n)
        e0.record()
        for i in range(it):
            ops.linear_bwd_weight(d, sets[i % n][0], sets[i % n][1], dw, db)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) / it * 1e3
        tot += us
        print(f"dW[{N:4d},{K:4d}] over {M:6d} rows: {us:7.1f} us  {(M * (N + K) * 2 + N * K * 4) / us / 1e3:6.0f} GB/s  {2.0 * M * N * K / us / 1e6:5.0f} TFLOP/s")
        del sets
        torch.cuda.empty_cache()
    print(f"sum {tot:.1f} us")


if __name__ == "__main__":
    main()
