#!/usr/bin/env python3
"""BUILD CONTAINER ONLY: the loss head at config 4's GLOBAL batch (B = 2048 windows = 512 subsequences) from the REFERENCE's
FOCALLoss (models/loss.py:139-218, fp32; it materialises [4, 1024, 1024, 128] broadcasts: ~2 GB per InfoNCE call, which fits here)
and from the oracle in fp64.  Inputs are the seeded views of tests/test_loss_gpu.py::_views, so only outputs are stored:
the four terms, the total, and every 16th row of dL/dz per (view, modality).  Output: tests/golden/loss_swt_b2048.npz."""
import os
import sys

import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402


def views(B, mods, seed, scale):
    g = torch.Generator().manual_seed(seed)
    f1 = {m: torch.randn(B, 256, generator=g) * scale for m in mods}
    f2 = {m: torch.randn(B, 256, generator=g) * scale for m in mods}
    for m in mods:
        f2[m] = 0.5 * f2[m] + 0.5 * f1[m]
    return f1, f2


def main():
    gg.install_reference()
    from models.loss import FOCALLoss
    from oracle.loss import focal_loss_terms
    torch.set_num_threads(8)
    cfg = yaml.safe_load(open("/root/reference/src/data/MOD.yaml"))
    B, seed, scale = 2048, 2048, 1.2
    mods = cfg["modality_names"]
    f1, f2 = views(B, mods, seed, scale)
    a1 = {m: f1[m].clone().requires_grad_(True) for m in mods}
    a2 = {m: f2[m].clone().requires_grad_(True) for m in mods}
    loss_fn = FOCALLoss(gg.ref_args("SW_Transformer", cfg))
    total = loss_fn(a1, a2)
    total.backward()
    d1 = {m: f1[m].double().requires_grad_(True) for m in mods}
    d2 = {m: f2[m].double().requires_grad_(True) for m in mods}
    terms = focal_loss_terms(d1, d2, cfg, "SW_Transformer")
    terms["total"].backward()
    out = {"B": B, "seed": seed, "scale": scale, "mods": np.array(mods), "loss.reference_total": float(total)}
    for k in ("shared", "private", "orth", "rank", "total"):
        out[f"loss.{k}"] = float(terms[k])
    rel = abs(float(total) - float(terms["total"])) / abs(float(terms["total"]))
    print("reference total", float(total), "oracle fp64 total", float(terms["total"]), "rel", rel)
    assert rel < 2e-5
    for m in mods:
        for tag, ref, ora in (("demb1", a1[m].grad, d1[m].grad), ("demb2", a2[m].grad, d2[m].grad)):
            e = ((ref.double() - ora).norm() / ora.norm()).item()
            print(tag, m, "reference fp32 vs oracle fp64 gradient", e)
            assert e < 2e-3  # the reference's fp32 cdist / broadcast path is the noisier of the two
            out[f"{tag}.{m}"] = ora[::16].float().numpy()
            out[f"{tag}_norm.{m}"] = float(ora.norm())
    np.savez_compressed(os.path.join(gg.OUT, "loss_swt_b2048.npz"), **out)
    print("written")


if __name__ == "__main__":
    main()
