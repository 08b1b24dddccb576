"""Run-to-run distribution of a training-mode forward value (same weights, same dropout seed): are there outliers beyond the fp32
summation-order spread of the split-K atomics?"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "focal_amd", "src"))
import numpy as np, torch
from oracle.config import load_config
from conftest import make_args, no_dropout
from models.SW_Transformer import SW_Transformer
from oracle.weights import fill_state_dict_, synthetic_freq_input
cfg = load_config()
for label, c, ct in (("fp32 dropout on", cfg, "fp32"), ("fp32 dropout off", no_dropout(cfg), "fp32"), ("bf16 dropout on", cfg, "bf16")):
    args = make_args(c, "SW_Transformer", torch.device("cuda"), ct)
    net = SW_Transformer(args)
    fill_state_dict_(net.state_dict())
    net = net.to("cuda").train()
    x = synthetic_freq_input(c, 4, seed=101)
    x = {l: {m: v.cuda() for m, v in mm.items()} for l, mm in x.items()}
    r = {m: torch.randn(4, 256, device="cuda", generator=torch.Generator("cuda").manual_seed(i)) for i, m in enumerate(c["modality_names"])}
    vals, per_mod = [], {m: [] for m in c["modality_names"]}
    with torch.no_grad():
        for it in range(400):
            net._fwd_calls = 0
            out = net(x, class_head=False, proj_head=False)
            tot = 0.0
            for m in out:
                v = (out[m] * r[m]).sum().item()
                per_mod[m].append(v)
                tot += v
            vals.append(tot)
    v = np.array(vals); med = np.median(v); dev = np.abs(v - med)
    print(f"{label}: median {med:.6f}  unique values {len(set(vals))}  p50 |dev| {np.median(dev):.2e}  p99 {np.percentile(dev, 99):.2e}  max {dev.max():.2e}  "
          f"count > 10 x p50: {(dev > 10 * max(np.median(dev), 1e-12)).sum()}")
    for m, pv in per_mod.items():
        pv = np.array(pv); d = np.abs(pv - np.median(pv))
        print(f"    {m}: unique {len(set(pv.tolist()))} max dev {d.max():.2e} (p50 {np.median(d):.2e})")
