#!/usr/bin/env python3
"""Per-parameter gradient error of the bf16 mode against the fp32 mode of the HIP path itself, at the BASELINE batch (B = 256), same
weights, same windows, dropout off (the worst gradient-norm errors, 5.4e-2 against a 6e-2 bound, had no owner).

  python tools/grad_error_table.py [SW_Transformer|DeepSense] [B]   -> gpurun_out/grad_error_<model>.json + a table on stdout

Per tensor: rel = ||g_bf16 - g_fp32|| / ||g_fp32||, norm = | ||g_bf16|| / ||g_fp32|| - 1 | (what the parity tests bound).  The table
groups tensors by (stage, layer kind) and lists the ten worst tensors."""
import collections
import json
import os
import re
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "focal_amd", "src"))
import torch  # noqa: E402
from conftest import make_args, no_dropout  # noqa: E402
from oracle.config import load_config  # noqa: E402
from oracle.weights import fill_state_dict_, synthetic_freq_input  # noqa: E402


def run(model, ct, B, cfg):
    if model == "DeepSense":
        from models.DeepSense import DeepSense as Net
    else:
        from models.SW_Transformer import SW_Transformer as Net
    from models.FOCALModules import FOCAL
    from models.loss import FOCALLoss
    args = make_args(cfg, model, torch.device("cuda"), ct)
    net = Net(args)
    fill_state_dict_(net.state_dict())
    net = net.to("cuda").train()
    focal, loss_fn = FOCAL(args, net), FOCALLoss(args)
    dev = lambda d: {l: {m: v.cuda() for m, v in mm.items()} for l, mm in d.items()}
    x1, x2 = dev(synthetic_freq_input(cfg, B, seed=101)), dev(synthetic_freq_input(cfg, B, seed=202))
    net.arena().zero_grad()
    f1, f2 = focal(x1, x2, proj_head=True)
    loss = loss_fn(f1, f2)
    loss.backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().double().cpu() for n, p in net.named_parameters() if p.grad is not None}, loss.item()


def kind(name):
    m = re.search(r"\.(\d)\.blocks\.(\d)\.(.+)$", name)
    if m:
        return f"stage {m.group(1)}", re.sub(r"\.(weight|bias)$", "", m.group(3)) + (" (bias)" if name.endswith("bias") else "")
    m = re.search(r"\.(\d)\.downsample\.(.+)$", name)
    if m:
        return f"stage {m.group(1)}", "merge." + m.group(2)
    return "tail / other", re.sub(r"^(loc_mod_extractors|recurrent_layers|mod_in_layers|mod_projectors|freq_interval_layers)\.", "", name)


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "SW_Transformer"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    cfg = no_dropout(load_config())
    g32, l32 = run(model, "fp32", B, cfg)
    g16, l16 = run(model, "bf16", B, cfg)
    rows, zero = [], []
    gmax = max(a.abs().max().item() for a in g32.values())
    for n, a in g32.items():
        b = g16[n]
        na = a.norm().item()
        if a.abs().max().item() < 1e-5 * gmax:  # analytically zero (a bias in front of a BatchNorm): only summation noise, no relative error
            zero.append(dict(name=n, norm_fp32=na, norm_bf16=b.norm().item()))
            continue
        rows.append(dict(name=n, numel=a.numel(), norm_fp32=na, rel=(b - a).norm().item() / max(na, 1e-30), norm_err=abs(b.norm().item() / max(na, 1e-30) - 1.0)))
    grp = collections.defaultdict(list)
    for r in rows:
        st, k = kind(r["name"])
        grp[(st, k if model == "SW_Transformer" else r["name"].split(".")[-3] + "." + r["name"].split(".")[-1])].append(r)
    print(f"{model} B={B}: loss fp32 {l32:.6f} bf16 {l16:.6f}; {len(rows)} tensors; arena rel L2 "
          f"{(sum((g16[n] - g32[n]).pow(2).sum() for n in g32) / sum(g32[n].pow(2).sum() for n in g32)).sqrt().item():.3e}")
    print(f"{'stage':14s} {'layer':44s} {'tensors':>7s} {'median rel':>11s} {'max rel':>9s} {'max norm err':>13s}")
    table = []
    for (st, k), rs in sorted(grp.items()):
        rel = sorted(r["rel"] for r in rs)
        e = dict(stage=st, layer=k, tensors=len(rs), median_rel=rel[len(rel) // 2], max_rel=rel[-1], max_norm_err=max(r["norm_err"] for r in rs))
        table.append(e)
        print(f"{st:14s} {k:44s} {len(rs):7d} {e['median_rel']:11.2e} {e['max_rel']:9.2e} {e['max_norm_err']:13.2e}")
    if zero:
        print(f"{len(zero)} tensors with an analytically zero gradient left out (|g| fp32 <= {max(z['norm_fp32'] for z in zero):.1e}, bf16 <= {max(z['norm_bf16'] for z in zero):.1e})")
    worst = sorted(rows, key=lambda r: -r["norm_err"])[:10]
    print("worst gradient-norm errors:")
    for r in worst:
        print(f"  {r['norm_err']:.2e} (rel {r['rel']:.2e}, |g| {r['norm_fp32']:.2e}, {r['numel']} elements)  {r['name']}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(dict(model=model, B=B, loss_fp32=l32, loss_bf16=l16, groups=table, worst_norm_err=worst, analytically_zero=zero),
              open(os.path.join(ROOT, "gpurun_out", f"grad_error_{model}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
