#!/usr/bin/env python3
"""Weight-gradient GEMM instances of the SW_Transformer step (B = 256, both views in one pass), HIP-event timed, cold operands
(rotated through > 600 MB).  python tools/mb_dw.py"""
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


if __name__ == "__main__" and (len(sys.argv) < 2 or sys.argv[1] != "group"):
    main()


def group_main():
    """python tools/mb_dw.py group: the four weight gradients of a Swin block, four launches (ring kernel) vs one (group kernel), cold."""
    cc = ops.code(BF)
    blocks = [(73728, 128), (36864, 128), (18432, 256), (9216, 256)]
    if os.environ.get("FOCAL_MB_DW_BLOCKS"):  # "rows,C;rows,C"
        blocks = [tuple(int(v) for v in t.split(",")) for t in os.environ["FOCAL_MB_DW_BLOCKS"].split(";")]
    for rows, C in blocks:
        shapes = [(C, 4 * C), (4 * C, C), (C, C), (3 * C, C)]
        per_set = sum(rows * (n + k) * 2 for n, k in shapes)
        nset = max(2, (700 << 20) // per_set + 1)
        sets = [[(torch.randn(rows, n, device=DEV).to(BF), torch.randn(rows, k, device=DEV).to(BF)) for n, k in shapes] for _ in range(nset)]
        outs = [(torch.zeros(n, k, device=DEV), torch.zeros(n, device=DEV)) for n, k in shapes]
        descs = [ops.linear_desc(cc, rows, n, k, cc, cc) for n, k in shapes]

        def four(i):
            for (dy, x), (dw, db), d in zip(sets[i % nset], outs, descs):
                ops.linear_bwd_weight(d, dy, x, dw, db)

        def one(i):
            ops.linear_bwd_weight_group(cc, [(dy, x, dw, db) for (dy, x), (dw, db) in zip(sets[i % nset], outs)])

        res = {}
        for name, fn in (("four launches", four), ("one group launch", one)):
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            it = max(20, 2 * nset)
            e0.record()
            for i in range(it):
                fn(i)
            e1.record()
            e1.synchronize()
            res[name] = e0.elapsed_time(e1) / it * 1e3
        flops = sum(2.0 * rows * n * k for n, k in shapes)
        wgs = ops.linear_bwd_weight_group_workgroups(cc, [(dy, x, dw, db) for (dy, x), (dw, db) in zip(sets[0], outs)])
        print(f"block rows {rows:6d} C {C:3d}: " + "  ".join(f"{k} {v:7.1f} us ({per_set / v / 1e3:5.0f} GB/s, {flops / v / 1e6:4.0f} TFLOP/s)" for k, v in res.items())
              + f"  [{wgs} workgroups]")
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "group":
    group_main()
