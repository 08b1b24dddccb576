"""Two random augmentations -> FOCAL forward -> loss (reference: train_utils/loss_calc_utils.py:1-22)."""
import os
import sys

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from focal_amd.distributed import gather_features  # noqa: E402



def calc_contrastive_loss(args, default_model, augmenter, loss_func, time_loc_inputs):
    if args.learn_framework == "FOCAL":
        aug_freq_loc_inputs_1 = augmenter.forward("random", time_loc_inputs)
        aug_freq_loc_inputs_2 = augmenter.forward("random", time_loc_inputs)
        feature1, feature2 = default_model(aug_freq_loc_inputs_1, aug_freq_loc_inputs_2, proj_head=True)
        # data parallel (torchrun): every rank sees the global batch of embeddings -> global negatives
        feature1, feature2 = gather_features([feature1, feature2])
        return loss_func(feature1, feature2)
    raise Exception(f"Invalid framework provided: {args.learn_framework}")


def calc_pretrain_loss(args, default_model, augmenter, loss_func, time_loc_inputs):
    if args.train_mode == "contrastive":
        return calc_contrastive_loss(args, default_model, augmenter, loss_func, time_loc_inputs)
    raise Exception(f"Invalid train mode: {args.train_mode}")
