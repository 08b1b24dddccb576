"""Is LayerNorm backward's gamma / beta reduction (and dx) reproducible run to run?  Many launches on the same inputs."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from focal_amd import ops
torch.manual_seed(0)
for ct in (torch.float32, torch.bfloat16):
    for rows, C in ((2304, 64), (1152, 64), (576, 128), (147456, 64), (18432, 256)):
        for acc in (True, False):
            dy = torch.randn(rows, C, device="cuda").to(ct)
            x = torch.randn(rows, C, device="cuda") * 2 + 0.3
            mean = x.mean(1, keepdim=True); rstd = (x.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
            stats = torch.cat([mean, rstd], 1).contiguous()
            gamma = torch.randn(C, device="cuda")
            g0 = torch.randn(rows, C, device="cuda")
            ref = None; worst = [0.0, 0.0, 0.0]; bad = 0
            n = 300 if rows < 100000 else 60
            for it in range(n):
                dx = g0.clone()
                dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
                ops.layernorm_bwd(dy, x, stats, gamma, dx, acc, dg, db)
                cur = (dg.clone(), db.clone(), dx.clone())
                if ref is None:
                    ref = cur; continue
                e = [((a - b).abs().max() / b.abs().max()).item() for a, b in zip(cur, ref)]
                worst = [max(w, v) for w, v in zip(worst, e)]
                if max(e) > 1e-4: bad += 1
            print(f"{str(ct)[6:]:9s} rows {rows:6d} C {C:3d} accumulate {acc!s:5s}: worst rel dev dgamma {worst[0]:.2e} dbeta {worst[1]:.2e} dx {worst[2]:.2e}  launches off by > 1e-4: {bad}/{n - 1}")
