import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from focal_amd import ops
DEV, BF = "cuda", torch.bfloat16
P = 0.2 if os.environ.get("FOCAL_MB_DROP", "1") == "1" else 0.0
rng = ops.new_rng_state(7, DEV)
w1, b1 = (torch.randn(256, 64, device=DEV) / 8).to(BF), torch.randn(256, device=DEV) * 0.1
w2 = (torch.randn(64, 256, device=DEV) / 16).to(BF)
dw1, db1, dw2, db2 = torch.zeros(256, 64, device=DEV), torch.zeros(256, device=DEV), torch.zeros(64, 256, device=DEV), torch.zeros(64, device=DEV)
for tiles_per_wg in (1, 2, 3, 4, 6, 9, 12):
    M = 128 * 256 * tiles_per_wg
    a = torch.randn(M, 64, device=DEV).to(BF); gm = torch.randn(M, 64, device=DEV).to(BF); da = torch.empty(M, 64, device=DEV, dtype=BF)
    bits = torch.randint(-2**31, 2**31 - 1, (M, 8), device=DEV, dtype=torch.int32)
    d = ops.mlp_desc(ops.code(BF), M, 64, 256, ops.drop_desc(rng, 1, P, 5, 0.0, 576), ops.drop_desc(rng, 2, P, 6, 0.0, 576))
    f = lambda: ops.mlp_bwd(d, gm, a, w1, b1, w2, da, dw1, db1, dw2, db2, mask_bits=bits if P else None)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); e1.synchronize()
    print(f"tiles/WG {tiles_per_wg:3d}  M {M:7d}  {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us")
