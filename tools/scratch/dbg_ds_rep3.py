"""Which samples of a replicated B = 256 DeepSense batch (32 copies of an 8-window block) deviate from their replicas, forward and backward?"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "focal_amd", "src"))
import torch
from oracle.config import load_config
from test_deepsense_parity_gpu import build
cfg = load_config()
args, net, _, _ = build(cfg, "bf16")
net.train()
g = torch.Generator().manual_seed(7)
x8 = {"shake": {"audio": torch.randn(8, 2, 10, 1600, generator=g).cuda(), "seismic": torch.randn(8, 2, 10, 20, generator=g).cuda()}}
r8 = {m: torch.randn(8, 256, generator=g).cuda() for m in cfg["modality_names"]}
rep = {"shake": {m: v.repeat(32, 1, 1, 1) for m, v in x8["shake"].items()}}
r256 = {m: v.repeat(32, 1) for m, v in r8.items()}
for it in range(6):
    net.arena().zero_grad()
    xin = {"shake": {m: v.clone().requires_grad_(True) for m, v in rep["shake"].items()}}
    out = net(xin, class_head=False, proj_head=True)
    sum((out[m] * r256[m]).sum() for m in out).backward()
    torch.cuda.synchronize()
    msg = [f"run {it}:"]
    for m in out:
        o = out[m].detach().float().view(32, 8, -1)
        dev = (o - o[0:1]).abs().amax(dim=(1, 2)) / o.abs().max()
        bad = [(i, round(v, 4)) for i, v in enumerate(dev.tolist()) if v > 2e-2]
        msg.append(f"{m}: fwd max replica deviation {dev.max().item():.4f} bad blocks {bad[:8]}")
        gx = xin["shake"][m].grad
        if gx is not None:
            gx = gx.float().view(32, 8, -1)
            gd = (gx - gx[0:1]).abs().amax(dim=(1, 2)) / gx.abs().max()
            badg = [(i, round(v, 3)) for i, v in enumerate(gd.tolist()) if v > 5e-2]
            msg.append(f"   dx max replica deviation {gd.max().item():.4f} bad blocks {badg[:10]}")
    print("\n  ".join(msg))
