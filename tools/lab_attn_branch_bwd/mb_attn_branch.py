#!/usr/bin/env python3
"""Stage-0 attention-branch backward at the step's shapes (B = 256 windows, both views): the two launches the engine uses --
focal_window_attn_qkv_bwd (q / k / v and dO formed in the kernel) + focal_linear_bwd_data_ln (dX of qkv + norm1's backward) -- against
the one-launch form focal_window_attn_branch_bwd (the window's four heads on four waves, dL/da1 partials summed through LDS).  Cold operands (rotated through > 600 MB).
  python tools/mb_attn_branch.py [iters]      FOCAL_MB_ONLY=two|one for rocprofv3 --pmc passes"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from focal_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ONLY = os.environ.get("FOCAL_MB_ONLY", "")


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
    table, gamma = torch.randn(25, heads, device=DEV) * 0.3, torch.ones(C, device=DEV)
    dt, dg, db = torch.zeros(25, heads, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    for name, (B, H, W) in (("audio", (512, 12, 48)), ("seismic", (512, 12, 24))):
        M = B * H * W
        cc = ops.code(BF)
        d = ops.attn_desc(cc, B, H, W, C, heads, 3, 3, 1, 1, 0.2, rng, 77)
        d_qkv = ops.linear_desc(cc, M, 3 * C, C, cc, cc)
        mask = ops.drop_desc(rng, 5, 0.2, 9, 0.1, H * W)
        n = max(2, (640 << 20) // (M * C * (2 + 2 + 4 + 4 + 2 + 6)) + 1)
        S = [dict(a1=torch.randn(M, C, device=DEV).to(BF), gm=torch.randn(M, C, device=DEV).to(BF), x=torch.randn(M, C, device=DEV),
                  st=torch.cat([torch.zeros(M, 1, device=DEV), torch.ones(M, 1, device=DEV)], 1).contiguous(), g=torch.randn(M, C, device=DEV),
                  gmn=torch.empty(M, C, device=DEV, dtype=BF), dqkv=torch.empty(M, 3 * C, device=DEV, dtype=BF)) for _ in range(n)]

        def two(i):
            s = S[i % n]
            ops.window_attn_qkv_bwd(d, s["a1"], wqkv, bqkv, table, s["gm"], s["dqkv"], dt, wproj=wproj)
            ops.linear_bwd_data_ln(d_qkv, s["dqkv"], wqkv, s["x"], s["st"], gamma, s["g"], dg, db, g_masked=s["gmn"], mask=mask)

        def one(i):
            s = S[i % n]
            ops.window_attn_branch_bwd(d, s["a1"], s["gm"], wqkv, bqkv, wproj, table, s["dqkv"], dt, s["x"], s["st"], gamma, s["g"], dg, db,
                                       g_masked=s["gmn"], mask=mask)
        if ONLY in ("", "two"):
            print(f"{name:8s} M = {M}: attention backward + (dX qkv + norm1 backward), two launches  {timed(two, ITERS):8.1f} us")
        if ONLY in ("", "one"):
            print(f"{name:8s} M = {M}: focal_window_attn_branch_bwd, one launch                    {timed(one, ITERS):8.1f} us")
        del S
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
