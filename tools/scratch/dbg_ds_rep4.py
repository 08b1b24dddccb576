"""Bimodal B = 256 DeepSense bf16 gradients: which tensors inside the backward differ between runs, and for which samples?"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "focal_amd", "src"))
import torch
from oracle.config import load_config
from test_deepsense_parity_gpu import build
from focal_amd import ops
cfg = load_config()
args, net, _, _ = build(cfg, "bf16")
net.train()
g = torch.Generator().manual_seed(7)
x8 = {"shake": {"audio": torch.randn(8, 2, 10, 1600, generator=g).cuda(), "seismic": torch.randn(8, 2, 10, 20, generator=g).cuda()}}
r8 = {m: torch.randn(8, 256, generator=g).cuda() for m in cfg["modality_names"]}
rep = {"shake": {m: v.repeat(32, 1, 1, 1) for m, v in x8["shake"].items()}}
r256 = {m: v.repeat(32, 1) for m, v in r8.items()}
print("modality order", cfg["modality_names"])
stash = []
orig = ops.gru_seq_bwd
def spy(gd, dout, ld_b, ld_t, scale, whh_t, hs, save, dgi, dgh):
    d_in = dout.clone()
    orig(gd, dout, ld_b, ld_t, scale, whh_t, hs, save, dgi, dgh)
    stash.append(dict(dout=d_in, dgi=[t.clone() for t in dgi], dgh=[t.clone() for t in dgh], stream=torch.cuda.current_stream().cuda_stream))
ops.gru_seq_bwd = spy
import focal_amd.deepsense_engine as de
de.ops.gru_seq_bwd = spy
runs = []
for it in range(6):
    stash.clear()
    net.arena().zero_grad()
    out = net(rep, class_head=False, proj_head=True)
    sum((out[m] * r256[m]).sum() for m in out).backward()
    torch.cuda.synchronize()
    runs.append((net.arena().grad.clone(), [dict(s) for s in stash]))
ref_g, ref_s = runs[0]
def rel(a, b): return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item()
for it in range(1, 6):
    gi, si = runs[it]
    line = [f"run {it} vs 0: arena {rel(gi, ref_g):.4f}; gru_seq_bwd calls {len(si)}"]
    for c, (a, b) in enumerate(zip(si, ref_s)):
        line.append(f"   call {c} (stream {a['stream']}): dout {rel(a['dout'], b['dout']):.5f} dgi {[round(rel(x, y), 5) for x, y in zip(a['dgi'], b['dgi'])]} dgh {[round(rel(x, y), 5) for x, y in zip(a['dgh'], b['dgh'])]}")
        d = (a['dout'].float() - b['dout'].float()).abs().amax(dim=-1) if a['dout'].dim() == 2 else None
        if d is not None and d.max() > 0:
            bad = torch.nonzero(d > 0.01 * b['dout'].abs().max()).flatten().tolist()
            line.append(f"      dout rows that differ: {bad[:40]} ({len(bad)} rows)")
    print("\n".join(line))
