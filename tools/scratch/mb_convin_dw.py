import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from focal_amd import ops
B = 256
x = torch.randn(B, 2, 10, 20, device="cuda")
dz = torch.randn(B * 10 * 20, 64, device="cuda").bfloat16()
dw = torch.zeros(64, 2, 1, 3, device="cuda"); db = torch.zeros(64, device="cuda")
d = ops.conv_in_desc(B, 2, 10, 20, 20, 3, 1, 1, 64)
for _ in range(3): ops.conv_in_bwd_weight(d, x, dz, dw, db)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(24): ops.conv_in_bwd_weight(d, x, dz, dw, db)
e1.record(); e1.synchronize()
print(os.environ.get("FOCAL_CONVIN_DW_BLOCKS", "512"), f"{e0.elapsed_time(e1) / 24 * 1e3:.1f} us")
