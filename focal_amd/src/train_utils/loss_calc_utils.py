"""Two random augmentations -> FOCAL forward -> loss (reference: train_utils/loss_calc_utils.py:1-22)."""


def calc_contrastive_loss(args, default_model, augmenter, loss_func, time_loc_inputs):
    if args.learn_framework == "FOCAL":
        aug_freq_loc_inputs_1 = augmenter.forward("random", time_loc_inputs)
        aug_freq_loc_inputs_2 = augmenter.forward("random", time_loc_inputs)
        feature1, feature2 = default_model(aug_freq_loc_inputs_1, aug_freq_loc_inputs_2, proj_head=True)
        return loss_func(feature1, feature2)
    raise Exception(f"Invalid framework provided: {args.learn_framework}")


def calc_pretrain_loss(args, default_model, augmenter, loss_func, time_loc_inputs):
    if args.train_mode == "contrastive":
        return calc_contrastive_loss(args, default_model, augmenter, loss_func, time_loc_inputs)
    raise Exception(f"Invalid train mode: {args.train_mode}")
