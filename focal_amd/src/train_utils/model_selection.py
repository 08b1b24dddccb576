"""Plugin registry (reference: train_utils/model_selection.py:14-59): model / framework / loss by name."""
from models.FOCALModules import FOCAL
from models.loss import FOCALLoss
from models.SW_Transformer import SW_Transformer


def init_backbone_model(args):
    if args.model == "DeepSense":
        from models.DeepSense import DeepSense
        classifier = DeepSense(args)
    elif args.model == "SW_Transformer":
        classifier = SW_Transformer(args)
    else:
        raise Exception(f"Invalid model provided: {args.model}")
    return classifier.to(args.device)


def init_contrastive_framework(args, backbone_model):
    if args.learn_framework == "FOCAL":
        default_model = FOCAL(args, backbone_model)
    else:
        raise NotImplementedError(f"Invalid {args.train_mode} framework {args.learn_framework} provided")
    return default_model.to(args.device)


def init_pretrain_framework(args, backbone_model):
    return init_contrastive_framework(args, backbone_model)


def init_loss_func(args):
    if args.train_mode == "contrastive" and args.stage == "pretrain":
        if args.learn_framework in {"FOCAL"}:
            return FOCALLoss(args).to(args.device)
        raise NotImplementedError(f"Invalid {args.train_mode} framework {args.learn_framework} provided")
    if args.stage == "finetune" or args.train_mode == "supervised":
        from models.loss import CrossEntropyLoss
        return CrossEntropyLoss()
    raise Exception(f"Train mode {args.train_mode} / stage {args.stage} is outside the MI355X path (FOCAL pretraining and finetuning)")
