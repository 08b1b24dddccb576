"""Optimizer selection (reference: train_utils/optimizer.py:3-35): same config keys, AdamW runs as the fused
arena kernel."""
import os
import sys

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from focal_amd.optim import FocalAdamW  # noqa: E402


def define_optimizer(args, parameters):
    if args.train_mode in {"supervised"}:
        cfg = args.dataset_config[args.model]["optimizer"]
    elif args.stage == "pretrain":
        cfg = args.dataset_config[args.learn_framework]["pretrain_optimizer"]
    elif args.stage == "finetune":
        cfg = args.dataset_config[args.learn_framework]["finetune_optimizer"]
    else:
        raise Exception("Optimizer not defined.")
    wd = cfg["weight_decay"][args.model] if isinstance(cfg["weight_decay"], dict) else cfg["weight_decay"]
    if cfg["name"] == "AdamW":
        return FocalAdamW(parameters, lr=cfg["start_lr"], weight_decay=wd)
    if cfg["name"] == "Adam":  # torch.optim.Adam: L2 weight decay folded into the gradient (the finetune optimizer)
        return FocalAdamW(parameters, lr=cfg["start_lr"], weight_decay=wd, l2_decay=True)
    raise NotImplementedError(f"Optimizer {cfg['name']} not implemented.")
