"""Repeat the same training-mode forward + backward (fixed dropout seed) and look for runs whose gradients deviate: which parameters?"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "focal_amd", "src"))
import torch
from oracle.config import load_config
from conftest import make_args
from models.SW_Transformer import SW_Transformer
from oracle.weights import fill_state_dict_, synthetic_freq_input
cfg = load_config()
ct, B, N = sys.argv[1] if len(sys.argv) > 1 else "fp32", int(sys.argv[2]) if len(sys.argv) > 2 else 4, int(sys.argv[3]) if len(sys.argv) > 3 else 300
args = make_args(cfg, "SW_Transformer", torch.device("cuda"), ct)
net = SW_Transformer(args)
fill_state_dict_(net.state_dict())
net = net.to("cuda").train()
x = synthetic_freq_input(cfg, B, seed=101)
x = {l: {m: v.cuda() for m, v in mm.items()} for l, mm in x.items()}
r = {m: torch.randn(B, 256, device="cuda", generator=torch.Generator("cuda").manual_seed(i)) for i, m in enumerate(cfg["modality_names"])}
ref = None
events = 0
for it in range(N):
    net._fwd_calls = 0
    net.arena().zero_grad()
    out = net(x, class_head=False, proj_head=False)
    sum((out[m] * r[m]).sum() for m in out).backward()
    torch.cuda.synchronize()
    g = net.arena().grad.clone()
    if ref is None:
        ref = g; continue
    ar = net.arena()
    bad = []
    for name, (off, n, shape) in ar.index.items():
        a, b = g[off:off + n], ref[off:off + n]
        d = ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()
        if d > 1e-3: bad.append((name, round(d, 4)))
    if bad:
        events += 1
        print(f"run {it}: {len(bad)} parameters deviate; first {bad[:3]} ... last {bad[-3:]}")
print(f"{ct} B={B}: {events} deviating runs of {N - 1}")
