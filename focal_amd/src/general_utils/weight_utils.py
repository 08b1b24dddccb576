"""Weight helpers (reference: general_utils/weight_utils.py:9-25, 61-94)."""
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


def set_learnable_params_finetune(args, classifier):
    """reference :61-80: FOCAL finetuning trains the class layer and the modality fusion layer, everything else is frozen."""
    learnable = []
    for name, param in classifier.named_parameters():
        hit = ("class_layer" in name or "mod_fusion_layer" in name) if args.learn_framework in {"FOCAL"} else "class_layer" in name
        param.requires_grad = hit
        if hit:
            learnable.append(param)
    return learnable
