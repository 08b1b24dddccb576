"""Time the FOCAL loss head (focal_loss_head: forward terms + dL/dz) at several global batch sizes -- under N-way data
parallelism every rank evaluates it on the GLOBAL batch (256 x N windows).  Usage: python tools/bench_loss_head.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from focal_amd import ops  # noqa: E402

dim = int(os.environ.get("DIM", "256"))
for B in (256, 512, 1024, 2048):
    f1 = [torch.randn(B, dim, device="cuda") for _ in range(2)]
    f2 = [torch.randn(B, dim, device="cuda") for _ in range(2)]
    w = (1.0, 1.0, 3.0, 1.0)
    for _ in range(3):
        ops.loss_head(f1, f2, 0.07, 1.0, w, 4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.loss_head(f1, f2, 0.07, 1.0, w, 4)
    e1.record()
    e1.synchronize()
    print(f"global batch {B:5d} (dim {dim}): loss head {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us")
