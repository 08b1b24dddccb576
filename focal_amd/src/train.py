"""Entry point: `python train.py -model=SW_Transformer -dataset=MOD -learn_framework=FOCAL [-batch_size=N] [-gpu=0]`
(reference: src/train.py:25-94).  FOCAL pretraining and finetuning (`-stage=finetune`) are implemented in this build."""
import logging
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from input_utils.multi_modal_dataloader import create_dataloader  # noqa: E402
from params.train_params import parse_train_params  # noqa: E402
from train_utils.model_selection import init_backbone_model, init_loss_func  # noqa: E402
from train_utils.pretrain import pretrain  # noqa: E402


def train(args):
    train_dataloader = create_dataloader("train", args, batch_size=args.batch_size, workers=args.workers)
    val_dataloader = create_dataloader("val", args, batch_size=args.batch_size, workers=args.workers)
    test_dataloader = create_dataloader("test", args, batch_size=args.batch_size, workers=args.workers)
    from data_augmenter.Augmenter import Augmenter
    augmenter = Augmenter(args)
    args.augmenter = augmenter
    classifier = init_backbone_model(args)
    args.classifier = classifier
    loss_func = init_loss_func(args)
    if args.train_mode == "contrastive" and args.stage == "pretrain":
        return pretrain(args, classifier, augmenter, train_dataloader, val_dataloader, test_dataloader, loss_func,
                        len(train_dataloader))
    if args.train_mode == "contrastive" and args.stage == "finetune":
        from train_utils.finetune import finetune
        return finetune(args, classifier, augmenter, train_dataloader, val_dataloader, test_dataloader, loss_func, len(train_dataloader))
    raise Exception(f"Invalid stage ({args.stage}) provided: FOCAL pretraining and finetuning are implemented on the HIP path.")


def main_train():
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    train(parse_train_params())


if __name__ == "__main__":
    main_train()
