#!/usr/bin/env python3
"""Microbenchmark of the fused Swin MLP kernels at the stage-0 shapes of the SW_Transformer step (B = 256 windows, both views in one
pass): audio M = 294 912 tokens, seismic M = 147 456.  HIP-event timed; `cold` rotates through enough operand sets that nothing is
cache-resident (> 512 MB), which is how the step meets these kernels.
  python tools/mb_mlp.py [iters]            (FOCAL_MB_ONLY=fwd|bwd, FOCAL_MB_DROP=0 to switch the dropout masks off)"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from focal_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ONLY = os.environ.get("FOCAL_MB_ONLY", "")
P = 0.2 if os.environ.get("FOCAL_MB_DROP", "1") == "1" else 0.0


def sets(M, n):
    out = []
    for i in range(n):
        g = torch.Generator(device=DEV).manual_seed(i)
        out.append(dict(a=torch.randn(M, 64, device=DEV, generator=g).to(BF), r=torch.randn(M, 64, device=DEV, generator=g),
                        gm=torch.randn(M, 64, device=DEV, generator=g).to(BF), y=torch.empty(M, 64, device=DEV),
                        da=torch.empty(M, 64, device=DEV, dtype=BF), bits=torch.empty(M, 8, device=DEV, dtype=torch.int32)))
    return out


def timed(fn, n):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    rng = ops.new_rng_state(7, DEV)
    w1, b1 = (torch.randn(256, 64, device=DEV) / 8).to(BF), torch.randn(256, device=DEV) * 0.1
    w2, b2 = (torch.randn(64, 256, device=DEV) / 16).to(BF), torch.randn(64, device=DEV) * 0.1
    gamma, beta = torch.ones(64, device=DEV), torch.zeros(64, device=DEV)
    dw1, db1, dw2, db2 = (torch.zeros(256, 64, device=DEV), torch.zeros(256, device=DEV), torch.zeros(64, 256, device=DEV),
                          torch.zeros(64, device=DEV))
    for name, M in (("audio", 294912), ("seismic", 147456)):
        d = ops.mlp_desc(ops.code(BF), M, 64, 256, ops.drop_desc(rng, 1, P, 5, 0.0, 576), ops.drop_desc(rng, 2, P, 6, 0.1 if P else 0.0, 576))
        per_set = M * 64 * (2 + 4 + 2 + 4 + 2)
        for mode, n in (("warm", 1), ("cold", max(2, (600 << 20) // per_set + 1))):
            S = sets(M, n)
            if ONLY in ("", "fwd"):
                us = timed(lambda i: ops.mlp_fwd(d, S[i % n]["a"], S[i % n]["r"], w1, b1, w2, b2, S[i % n]["y"], mask_bits=S[i % n]["bits"] if P else None), ITERS)
                byt = M * 64 * (2 + 4 + 4)
                print(f"mlp_fwd      {name:8s} {mode}: {us:8.1f} us  {byt / us / 1e3:7.0f} GB/s  {2 * 2 * M * 64 * 256 / us / 1e6:6.0f} TFLOP/s")
                us = timed(lambda i: ops.mlp_fwd(d, S[i % n]["a"], S[i % n]["r"], w1, b1, w2, b2, S[i % n]["y"], next_ln=(gamma, beta), mask_bits=S[i % n]["bits"] if P else None), ITERS)
                print(f"mlp_fwd + LN {name:8s} {mode}: {us:8.1f} us")
            if ONLY in ("", "bwd"):
                us = timed(lambda i: ops.mlp_bwd(d, S[i % n]["gm"], S[i % n]["a"], w1, b1, w2, S[i % n]["da"], dw1, db1, dw2, db2, mask_bits=S[i % n]["bits"] if P else None), ITERS)
                byt = M * 64 * (2 + 2 + 2)
                print(f"mlp_bwd      {name:8s} {mode}: {us:8.1f} us  {byt / us / 1e3:7.0f} GB/s  {5 * 2 * M * 64 * 256 / us / 1e6:6.0f} TFLOP/s")
            del S
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
