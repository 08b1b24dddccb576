#!/usr/bin/env python3
"""How long does the HOST spend inside hipGraphLaunch for the captured step (259 kernel nodes on two branches)?  ROCm's graph executor
enqueues the nodes one AQL packet at a time in topological (= capture) order: if that takes a millisecond, the second branch's first kernel
cannot start before the host has written the first branch's packets.  Prints per-step: host time in replay(), total step time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "focal_amd", "src")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    sys.argv = [sys.argv[0]] + sys.argv[1:]
    a = bench.parse()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    step = bench.Step(a, dev)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(2):
            step.run()
        torch.cuda.synchronize()
        run = step.capture(side)
        for _ in range(10):
            run()
            step.loss.item()
        host, total = [], []
        for _ in range(30):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run()
            t1 = time.perf_counter()
            step.loss.item()
            t2 = time.perf_counter()
            host.append((t1 - t0) * 1e3)
            total.append((t2 - t0) * 1e3)
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"graph replay: host time inside replay() median {med(host):.3f} ms (min {min(host):.3f}), step {med(total):.3f} ms")


if __name__ == "__main__":
    main()
