import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from focal_amd import ops
B = 512
x = torch.randn(B, 2, 10, 1600, device="cuda")
w = torch.randn(64, 2, 1, 40, device="cuda") * 0.1
b = torch.randn(64, device="cuda"); g = torch.ones(64, device="cuda"); be = torch.zeros(64, device="cuda")
xs = [torch.randn_like(x) for _ in range(6)]
for cold in (False, True):
    for _ in range(3): ops.pad_patch_embed_ln(x, w, b, g, be, 12, 48, 40)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(24): ops.pad_patch_embed_ln(xs[i % 6] if cold else x, w, b, g, be, 12, 48, 40)
    e1.record(); e1.synchronize()
    print("cold" if cold else "warm", os.environ.get("FOCAL_EMBED_BLOCKS", "1024"), f"{e0.elapsed_time(e1) / 24 * 1e3:.1f} us")
