import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from focal_amd import ops
x = torch.randn(256, 2, 10, 1600, device="cuda")
out = torch.empty(256, 4, 10, 1600, device="cuda")
for _ in range(3): ops.fft_realpack(x, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(24): ops.fft_realpack(x, out=out)
e1.record(); e1.synchronize()
print(os.environ.get("FOCAL_FFT_BLOCKS", "1024"), f"{e0.elapsed_time(e1) / 24 * 1e3:.1f} us")
