"""Entry point: `python train.py -model=SW_Transformer -dataset=MOD -learn_framework=FOCAL [-batch_size=N] [-gpu=0]`
(reference: src/train.py:25-94).  FOCAL pretraining, finetuning (`-stage=finetune`) and supervised training from scratch
(`-learn_framework=no`) are implemented in this build."""
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
    if args.train_mode == "supervised":
        from train_utils.supervised_train import supervised_train
        return supervised_train(args, classifier, augmenter, train_dataloader, val_dataloader, test_dataloader, loss_func, len(train_dataloader))
    if args.train_mode == "contrastive" and args.stage == "pretrain":
        return pretrain(args, classifier, augmenter, train_dataloader, val_dataloader, test_dataloader, loss_func,
                        len(train_dataloader))
    if args.train_mode == "contrastive" and args.stage == "finetune":
        from train_utils.finetune import finetune
        return finetune(args, classifier, augmenter, train_dataloader, val_dataloader, test_dataloader, loss_func, len(train_dataloader))
    raise Exception(f"Invalid stage ({args.stage}) provided: FOCAL pretraining and finetuning are implemented on the HIP path.")


def spawn_data_parallel(devices):
    """`-gpu=0,1,...,7` (the reference already parses device lists, params/params_util.py:34-38, but trains on the first one):
    one child process per listed GPU, each a rank of an RCCL data-parallel job (focal_amd/launch.py, shared with `bench.py --gpus N`).
    Runs before anything touches the GPU in this process."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    from focal_amd.launch import spawn_ranks
    sys.exit(spawn_ranks(__file__, sys.argv[1:], devices))


def wants_data_parallel(base):
    """Only FOCAL pretraining is data-parallel (train_utils/pretrain.py: broadcast weights, rank shares of the loader, gathered
    embeddings, summed gradients, rank-0 validation).  The supervised and finetune stages run on the first listed device, which is
    what the reference does with a device list in every stage (params/params_util.py:43)."""
    return getattr(base, "learn_framework", "FOCAL") == "FOCAL" and getattr(base, "stage", "pretrain") == "pretrain"


def main_train():
    from params.base_params import parse_base_args
    from params.params_util import parse_device_list
    base = parse_base_args("train")
    if int(os.environ.get("WORLD_SIZE", "1")) == 1:
        devices = parse_device_list(base.gpu)
        if len(devices) > 1 and wants_data_parallel(base):
            spawn_data_parallel(devices)
    elif not wants_data_parallel(base):
        raise SystemExit("only FOCAL pretraining runs data-parallel: launch the supervised / finetune stages as one process "
                         "(ranks of such a job would each train their own model and write the same checkpoints)")
    rank = int(os.environ.get("RANK", "0"))
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING, format="%(message)s")
    try:
        train(parse_train_params())
    finally:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main_train()
