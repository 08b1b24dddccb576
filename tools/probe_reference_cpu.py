#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference; never runs on the GPU box): how fast is the oracle -- the "port" that bench.py's
`cpu_baseline` times on the GPU box's host cores -- relative to the reference's own train loop body on the same cores?

SURVEY 8d-i asks for the two to be within +-15 %; they are not (the oracle is a functional restatement without the reference's
~270 small ATen ops / permute-contiguous copies per view), so the measured RATIO is recorded and bench.py carries it next to its
number as `reference_ratio_probed`: reference-equivalent windows/s = cpu_baseline.value x ratio.

Both legs: same threads, batch 32, fp32, MOD-shaped N(0,1) windows, two DISTINCT views (view 2 = -1.1 x view 1, as the GPU leg), the
loop body of train_utils/pretrain.py:62-74 (zero_grad, FFT of both views, backbone x2, FOCALLoss, backward, AdamW step, loss.item()).
The reference runs once in train() mode with its configured dropout (what its users run) and once with every dropout rate 0 (what
the oracle computes).  Output: profiles/cpu_equivalence.json."""
import copy
import json
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_golden as gg  # noqa: E402  (stand-ins for timm / tsai + sys.path for /root/reference/src)


def timed(fn, warm=1, steps=4, budget=60.0):
    for _ in range(warm):
        fn()
    t0, n = time.time(), 0
    while n < steps and time.time() - t0 < budget:
        fn()
        n += 1
    return (time.time() - t0) / max(n, 1), n


def main():
    gg.install_reference()
    import yaml
    threads = int(os.environ.get("FOCAL_PROBE_THREADS", "8"))
    torch.set_num_threads(threads)
    B = 32
    cfg = yaml.safe_load(open("/root/reference/src/data/MOD.yaml"))
    from oracle.step import OracleTrainer, fft_realpack
    from oracle.weights import fill_state_dict_, synthetic_time_input
    out = {"threads": threads, "batch": B, "torch": torch.__version__, "views": "x and -1.1 x (distinct)", "models": {}}
    for model in ("SW_Transformer", "DeepSense"):
        x = synthetic_time_input(cfg, B, 5)
        x2 = {l: {m: -1.1 * v for m, v in mm.items()} for l, mm in x.items()}
        res = {}
        for tag, c in (("reference_train_mode_dropout_on", cfg), ("reference_dropout_off", gg.no_dropout(cfg))):
            args = gg.ref_args(model, copy.deepcopy(c))
            from models.FOCALModules import FOCAL
            from models.loss import FOCALLoss
            if model == "DeepSense":
                from models.DeepSense import DeepSense as Net
            else:
                from models.SW_Transformer import SW_Transformer as Net
            from input_utils.time_input_utils import fft_preprocess
            net = Net(args)
            fill_state_dict_(net.state_dict())
            focal, loss_fn = FOCAL(args, net), FOCALLoss(args)
            opt = torch.optim.AdamW(focal.parameters(), lr=1e-3, weight_decay=0.05)
            for n_, p_ in focal.backbone.named_parameters():
                if "patch_embed" in n_:
                    p_.requires_grad = False
            focal.train()

            def ref_step():
                opt.zero_grad()
                f1, f2 = fft_preprocess(x, args), fft_preprocess(x2, args)
                a, b = focal(f1, f2, proj_head=True)
                loss = loss_fn(a, b)
                loss.backward()
                opt.step()
                return loss.item()
            dt, n = timed(ref_step)
            res[tag] = {"s_per_step": round(dt, 4), "windows_per_s": round(B / dt, 2), "steps": n}
        net = Net(gg.ref_args(model, gg.no_dropout(cfg)))
        state = {k: v.detach().clone() for k, v in net.state_dict().items()}
        fill_state_dict_(state)
        tr = OracleTrainer(model, gg.no_dropout(cfg), state)
        dt, n = timed(lambda: tr.step(freq_pair=(fft_realpack(x), fft_realpack(x2))))
        res["oracle_port"] = {"s_per_step": round(dt, 4), "windows_per_s": round(B / dt, 2), "steps": n}
        res["reference_over_oracle"] = round(res["reference_train_mode_dropout_on"]["windows_per_s"] / res["oracle_port"]["windows_per_s"], 3)
        res["reference_dropout_off_over_oracle"] = round(res["reference_dropout_off"]["windows_per_s"] / res["oracle_port"]["windows_per_s"], 3)
        out["models"][model] = res
        print(model, json.dumps(res))
    with open(os.path.join(ROOT, "profiles", "cpu_equivalence.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
