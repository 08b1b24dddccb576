"""Time the FOCAL loss head (forward terms + dL/dz) at the global batch of N-way data parallelism (256 x N windows):
  replicated  focal_loss_head on the whole global batch -- what every rank ran in round 1;
  sharded     focal_loss_head_shard_a + _b of rank 0 -- the rows of the rank's own 256 windows (round 2); the ~70 KB all-gather between
              the two phases is not in the number (one latency-bound collective).
Usage: python tools/bench_loss_head.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from focal_amd import ops  # noqa: E402

dim = int(os.environ.get("DIM", "256"))
w = (1.0, 1.0, 3.0, 1.0)


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for world in (1, 2, 4, 8):
    B = 256 * world
    f1 = [torch.randn(B, dim, device="cuda") for _ in range(2)]
    f2 = [torch.randn(B, dim, device="cuda") for _ in range(2)]
    rep = timed(lambda: ops.loss_head(f1, f2, 0.07, 1.0, w, 4))
    line = f"global batch {B:5d} (dim {dim}, {world} rank{'s' if world > 1 else ' '}): replicated {rep:8.1f} us"
    if world > 1:
        head = ops.ShardedLossHead(0, world)

        def sharded():
            head.phase_a(f1, f2, 0.07, 1.0, w, 4)
            head.chunks[0].copy_(head.send)  # stands in for the all-gather (the other ranks' chunks keep stale but finite values)
            head.phase_b()
        sharded()
        head.chunks[1:] = head.chunks[0]
        line += f"   sharded (rank 0 of {world}) {timed(sharded):8.1f} us"
    print(line)
