import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from focal_amd import ops
B = 256
x = torch.randn(B, 2, 10, 1600, device="cuda")
w = torch.randn(64, 2, 1, 80, device="cuda") * 0.1
b = torch.randn(64, device="cuda")
d = ops.conv_in_desc(B, 2, 10, 1600, 20, 80, 80, 0, 64)
for _ in range(3): ops.conv_in_fwd(d, x, w, b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(24): ops.conv_in_fwd(d, x, w, b)
e1.record(); e1.synchronize()
print(os.environ.get("FOCAL_CONVIN_BLOCKS", "768"), f"{e0.elapsed_time(e1) / 24 * 1e3:.1f} us")
