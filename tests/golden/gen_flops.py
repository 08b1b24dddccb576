#!/usr/bin/env python3
"""BUILD CONTAINER ONLY: algorithmic FLOPs of one FOCAL pretraining step per window, counted on the REFERENCE models with
torch.utils.flop_counter (SURVEY 8d's method): forward of one view at batch 8, then step = 2 views x (1 fwd + 2 bwd).  Datasets: the
reference's MOD.yaml and this build's 4-modality HAR4.yaml (BASELINE.json configs[4]).  Output: tests/golden/flops.json (data that
bench.py reads for `model_flops_frac_of_bf16_mfma_peak`)."""
import copy
import json
import os
import sys

import torch
import yaml
from torch.utils.flop_counter import FlopCounterMode

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402


def main():
    gg.install_reference()
    from models.DeepSense import DeepSense
    from models.SW_Transformer import SW_Transformer
    from oracle.weights import synthetic_freq_input
    out = {}
    for ds, path in (("MOD", "/root/reference/src/data/MOD.yaml"), ("HAR4", os.path.join(gg.REPO, "focal_amd", "src", "data", "HAR4.yaml"))):
        cfg = gg.no_dropout(yaml.safe_load(open(path)))
        for model, Net in (("SW_Transformer", SW_Transformer), ("DeepSense", DeepSense)):
            args = gg.ref_args(model, copy.deepcopy(cfg))
            args.dataset = ds
            args.task = "vehicle_classification" if ds == "MOD" else "activity_classification"
            net = Net(args).eval()
            B = 8
            x = synthetic_freq_input(cfg, B, 3)
            with FlopCounterMode(display=False) as fc, torch.no_grad():
                net(x, class_head=False, proj_head=True)
            fwd = fc.get_total_flops() / B
            out[f"{model}/{ds}"] = {"fwd_one_view_per_window": fwd, "step_per_window": 6 * fwd,
                                    "params": sum(p.numel() for p in net.parameters())}
            print(model, ds, f"{fwd / 1e9:.4f} GF fwd/view/window, step {6 * fwd / 1e9:.3f} GF/window")
    json.dump(out, open(os.path.join(gg.OUT, "flops.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
