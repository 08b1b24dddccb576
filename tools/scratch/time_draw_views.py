"""How long does the product augmenter's per-step draw (two random views) take on the host and on the GPU?  (bench.py --views random)"""
import sys, os, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "focal_amd", "src"))
import torch
import bench
sys.argv = [sys.argv[0], "--views", "random", "--no-cpu-baseline", "--no-roofline", "--no-secondary"] + sys.argv[1:]
a = bench.parse()
import cProfile, pstats
def main():
    dev = torch.device("cuda:0")
    step = bench.Step(a, dev) if hasattr(bench, "Step") else None
    for _ in range(3):
        step.draw_views(); step.run()
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        step.draw_views()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"draw_views: host {1e3 * (t1 - t0) / n:.3f} ms per step enqueue, {1e3 * (t2 - t0) / n:.3f} ms per step including the GPU")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(n):
        step.draw_views()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
main()
