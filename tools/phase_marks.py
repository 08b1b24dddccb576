#!/usr/bin/env python3
"""The REAL timeline of the hipGraph-replayed step at branch granularity: one-thread marker kernels (focal_mark: device wall clock) at the
start and end of every encoder's forward pass, backward blocks and backward tail, inside the captured graph.  rocprofv3 serialises the
branches and events cannot be read inside a graph; this is what runs beside what, unperturbed (2 x ~12 launches of ~2 us).
  python3 tools/phase_marks.py [--model ... --dataset ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "focal_amd", "src")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from focal_amd import ops, swin_engine  # noqa: E402

NAMES = []


def slot(name):
    if name not in NAMES:
        NAMES.append(name)
    return NAMES.index(name)


def main():
    a = bench.parse()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    marks = torch.zeros(256, dtype=torch.int64, device=dev)
    E = swin_engine.SwinModEncoder
    f0, b0, bb0, bt0 = E.forward, E.backward, E._backward_blocks, E._backward_tail

    def fwd(self, x, view, training):
        ops.mark(marks, slot(f"{self.mod}.fwd.begin"))
        out = f0(self, x, view, training)
        ops.mark(marks, slot(f"{self.mod}.fwd.end"))
        return out

    def bwd(self, saved, dfeat):
        ops.mark(marks, slot(f"{self.mod}.bwd.begin"))
        out = b0(self, saved, dfeat)
        ops.mark(marks, slot(f"{self.mod}.bwd.end"))
        return out

    def bblocks(self, saved, state, stop):
        nxt = state["next"]
        out = bb0(self, saved, state, stop)
        return out
    E.forward, E.backward = fwd, bwd
    step = bench.Step(a, dev)
    seg = step.seg
    a0, c0 = seg.seg_a, seg.seg_c

    def seg_a():
        ops.mark(marks, slot("step.begin"))
        a0()

    def seg_c():
        ops.mark(marks, slot("adamw.begin"))
        c0()
        ops.mark(marks, slot("step.end"))
    seg.seg_a, seg.seg_c = seg_a, seg_c
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(2):
            step.run()
        torch.cuda.synchronize()
        run = step.capture(side)
        for _ in range(10):
            run()
            step.loss.item()
        rec = []
        for _ in range(20):
            t0 = time.perf_counter()
            run()
            step.loss.item()
            dt = (time.perf_counter() - t0) * 1e3
            torch.cuda.synchronize()
            m = marks[:len(NAMES)].cpu().tolist()
            rec.append((dt, m))
    med = lambda v: sorted(v)[len(v) // 2]
    base = slot("step.begin")
    print(f"replayed step (host clock, with the markers in the graph): median {med([r[0] for r in rec]):.3f} ms")
    order = sorted(range(len(NAMES)), key=lambda i: med([(r[1][i] - r[1][base]) for r in rec]))
    for i in order:
        us = med([(r[1][i] - r[1][base]) / 100.0 for r in rec])
        print(f"  {us:9.1f} us  {NAMES[i]}")


if __name__ == "__main__":
    main()
