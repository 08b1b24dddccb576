"""Weight helpers on the pretraining path (reference: general_utils/weight_utils.py:9-25, 85-94)."""
import torch


def load_model_weight(args, model, weight_file, load_class_layer=True):
    trained = torch.load(weight_file, map_location=args.device)
    own = model.state_dict()
    picked = {k: v for k, v in trained.items() if k in own and (load_class_layer or "class_layer" not in k)}
    own.update(picked)
    model.load_state_dict(own)
    return model


def freeze_patch_embedding(args, default_model):
    """Called AFTER the optimizer exists (reference pretrain.py:42): the patch embedding simply never gets grads."""
    if "Fusion" not in args.learn_framework:
        for name, param in default_model.backbone.named_parameters():
            if "patch_embed" in name:
                param.requires_grad = False
    return default_model
