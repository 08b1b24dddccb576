"""Where do the small device-to-device copies of one eager step come from?  (torch.profiler, python stacks of aten::copy_ / clone / cat)"""
import os, sys, collections
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--no-graph", "--no-cpu-baseline", "--no-roofline"]
import torch
import bench
a = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
step = bench.Step(a, dev)
for _ in range(3):
    step.run()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step.run()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::cat", "aten::stack", "aten::fill_", "aten::zero_", "aten::contiguous", "aten::_to_copy", "aten::zeros", "aten::mul_"):
        st = [s for s in (e.stack or []) if "focal_amd" in s or "bench.py" in s or "/src/" in s]
        cnt[(e.name, str(e.input_shapes)[:60], st[0][-90:] if st else "?")] += 1
for k, v in cnt.most_common(40):
    print(v, k)
print("---- ops that launched a Memcpy / Memset")
cnt = collections.Counter()
for e in prof.events():
    ks = [k for k in (getattr(e, "kernels", None) or []) if "emcpy" in k.name or "emset" in k.name or "copyBuffer" in k.name]
    if ks:
        st = [s for s in (e.stack or []) if "focal_amd" in s or "bench.py" in s or "/src/" in s]
        cnt[(e.name, ks[0].name[:40], str(e.input_shapes)[:50], " <- ".join(x[-70:] for x in st[:2]) if st else "?")] += len(ks)
for k, v in cnt.most_common(60):
    print(v, k)
print("---- device-side events named like copies")
c2 = collections.Counter()
for e in prof.events():
    if ("emcpy" in e.name or "emset" in e.name or "copyBuffer" in e.name):
        c2[(e.name[:60], str(e.device_type))] += 1
for k, v in c2.most_common(20):
    print(v, k)
