#!/usr/bin/env python3
"""Stage-0 window-attention backward (q / k / v and dO formed in the kernel: focal_window_attn_qkv_bwd) at the step's shapes
(both views of 256 windows; audio 12 x 48 and seismic 12 x 24 tokens, 3 x 3 windows, plain and shifted), cold operands (rotated
through > 600 MB).      python tools/mb_attn_bwd.py [iters]     (FOCAL_ATTN_BWD_NW=16|8|4 in lab builds selects the kernel form)"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from focal_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 30


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
    C, heads = 64, 4
    rng = ops.new_rng_state(7, DEV)
    wqkv, bqkv = (torch.randn(3 * C, C, device=DEV) / 8).to(BF), torch.randn(3 * C, device=DEV) * 0.1
    wproj = (torch.randn(C, C, device=DEV) / 8).to(BF)
    table = torch.randn(25, heads, device=DEV) * 0.3
    dt = torch.zeros(25, heads, device=DEV)
    for name, (B, H, W) in (("audio", (512, 12, 48)), ("seismic", (512, 12, 24))):
        M = B * H * W
        n = max(2, (640 << 20) // (M * C * (2 + 2 + 6)) + 1)
        S = [dict(a1=torch.randn(M, C, device=DEV).to(BF), gm=torch.randn(M, C, device=DEV).to(BF), dqkv=torch.empty(M, 3 * C, device=DEV, dtype=BF))
             for _ in range(n)]
        for shift in (0, 1):
            d = ops.attn_desc(ops.code(BF), B, H, W, C, heads, 3, 3, shift, shift, 0.2, rng, 77)

            def run(i):
                s = S[i % n]
                ops.window_attn_qkv_bwd(d, s["a1"], wqkv, bqkv, table, s["gm"], s["dqkv"], dt, wproj=wproj)
            us = timed(run, ITERS)
            print(f"{name:8s} shift {shift}  M = {M:7d}  {us:7.1f} us   {M * C * 10 / us / 1e6:6.2f} TB/s of a1 + gm + dqkv")


if __name__ == "__main__":
    main()
