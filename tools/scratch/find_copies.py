"""Which host lines launch the small copy / fill kernels of a step (41 per SW_Transformer step in r2_z_swt_instances.txt)?
One eager step under torch.profiler with stacks; prints aten::copy_ / fill_ / zero_ / cat / stack calls grouped by source line."""
import collections, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "focal_amd", "src"))
import torch
import bench
a = bench.parse()
dev = torch.device("cuda", 0)
step = bench.Step(a, dev)
for _ in range(3):
    step.run()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step.run()
    torch.cuda.synchronize()
by = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::cat", "aten::stack", "aten::zeros", "aten::contiguous", "aten::clone", "aten::add_", "aten::mul", "aten::sum"):
        st = [s for s in (e.stack or []) if "focal_amd" in s or "bench.py" in s or "/src/" in s]
        by[(e.name, st[0] if st else "?")] += 1
for (n, s), c in sorted(by.items(), key=lambda kv: -kv[1]):
    print(f"{c:4d}  {n:18s} {s}")
