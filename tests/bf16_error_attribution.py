#!/usr/bin/env python3
"""Where the bf16 path's 1e-2-of-scale embedding error comes from, on the CPU: the oracle in fp32 arithmetic with the
operands the HIP path keeps in bf16 rounded (oracle/swt.py: emulate_bf16), against the reference fixture SW_Transformer_b8.npz -- once
with every site rounded, then with one GROUP of sites left in fp32 at a time.  The drop of the error is what giving that group fp32 (or
split hi + lo bf16) operands would buy.  Not a test (no test_ prefix): python tests/bf16_error_attribution.py [--b256]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "focal_amd", "src"), os.path.dirname(__file__)):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import swt  # noqa: E402
from oracle.config import load_config  # noqa: E402
from oracle.weights import fill_state_dict_, swt_state_spec, synthetic_freq_input  # noqa: E402

GROUPS = [
    ("nothing (all sites rounded)", []),
    ("mod_in_layers: x and weight", [r"^mod_in_layers\."]),
    ("mod_in_layers: weight only", [r"^mod_in_layers\..*weight$"]),
    ("mod_in_layers: x only", [r"^mod_in_layers\..*\.x$"]),
    ("projector (both layers, x and weight)", [r"^mod_projectors\."]),
    ("mod_in + projector", [r"^mod_in_layers\.", r"^mod_projectors\."]),
    ("stage 2: everything", [r"\.2\.blocks\."]),
    ("stage 2: proj + fc2 weights", [r"\.2\.blocks\.\d\.(attn\.proj|mlp\.fc2)\.weight$"]),
    ("stage 2: proj + fc2 weights and inputs", [r"\.2\.blocks\.\d\.(attn\.proj\.weight|mlp\.fc2\.weight|attn\.out|mlp\.gelu\.out)$"]),
    ("stage 2: LayerNorm outputs", [r"\.2\.blocks\.\d\.norm\d\.out$"]),
    ("stage 2: qkv out + probs", [r"\.2\.blocks\.\d\.attn\.(qkv\.out|probs)$"]),
    ("stage 1: everything", [r"\.1\.blocks\.", r"\.1\.downsample"]),
    ("stage 0: everything", [r"\.0\.blocks\.", r"\.0\.downsample"]),
    ("all LayerNorm outputs", [r"norm\d?\.out$"]),
    ("all weights", [r"weight$"]),
    ("all activations", [r"(out|probs|\.x)$"]),
    ("tail: mod_in + projector + stage 2 proj/fc2 (w + inputs)", [r"^mod_in_layers\.", r"^mod_projectors\.", r"\.2\.blocks\.\d\.(attn\.proj\.weight|mlp\.fc2\.weight|attn\.out|mlp\.gelu\.out)$"]),
]


def main():
    cfg = load_config()
    for k in ("dropout_ratio", "drop_path_rate", "attn_drop_rate"):
        cfg["SW_Transformer"][k] = 0.0
    fx = np.load(os.path.join(ROOT, "tests", "golden", "SW_Transformer_b8.npz"))
    spec = swt_state_spec(cfg, "vehicle_classification")
    state = {k: (torch.zeros(shp, dtype=torch.long) if k.endswith(("relative_position_index", "num_batches_tracked")) else torch.zeros(shp)) for k, shp in spec.items()}
    fill_state_dict_(state)
    x = synthetic_freq_input(cfg, 8, seed=101)
    x2 = synthetic_freq_input(cfg, 8, seed=202)
    from oracle.loss import focal_loss_terms
    torch.set_num_threads(8)
    with torch.no_grad():
        exact = swt.swt_forward(state, cfg, x)
        for m in exact:
            ref = torch.from_numpy(fx[f"eval.emb.{m}"])
            print(f"fp32 oracle vs fixture, {m}: {((exact[m] - ref).abs().max() / ref.abs().max()).item():.2e}")
        print(f"{'sites left in fp32':62s} " + "  ".join(f"{m:>9s}" for m in exact) + "   |d rank| / max(1, |rank|)   (other terms)")
        swt._TAIL_FP32[0] = False   # the baseline row is the round-5 path: every site rounded, mod_in and the projector included
        for name, keep in GROUPS:
            swt._KEEP_FP32[:] = keep
            emu = swt.swt_forward(state, cfg, x, emulate_bf16=True)
            errs = []
            for m in emu:
                ref = torch.from_numpy(fx[f"eval.emb.{m}"])
                errs.append(((emu[m] - ref).abs().max() / ref.abs().max()).item())
            emu2 = swt.swt_forward(state, cfg, x2, emulate_bf16=True)
            terms = focal_loss_terms(emu, emu2, cfg, "SW_Transformer")
            te = {k: abs(float(terms[k]) - float(fx[f"train.loss.{k}"])) / max(1.0, abs(float(fx[f"train.loss.{k}"]))) for k in ("rank", "shared", "private", "orth")}
            print(f"{name:62s} " + "  ".join(f"{e * 1e2:8.3f}e-2" for e in errs) + f"   {te['rank'] * 1e2:7.3f}e-2   ({te['shared']:.1e} {te['private']:.1e} {te['orth']:.1e})", flush=True)
        swt._KEEP_FP32[:] = []
        swt._TAIL_FP32[0] = True


if __name__ == "__main__":
    main()
