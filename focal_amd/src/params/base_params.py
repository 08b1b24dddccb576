"""Command-line surface (reference: params/base_params.py:10-85): the same single-dash flags."""
import argparse


def parse_base_args(option="train"):
    p = argparse.ArgumentParser()
    p.add_argument("-dataset", type=str, default="MOD", help="Dataset to evaluate.")
    p.add_argument("-task", type=str, default=None, help="The downstream task to evaluate.")
    p.add_argument("-learn_framework", type=str, default="no", help="No/Contrastive/Predictive/Reconstruction learning framework to use")
    p.add_argument("-stage", type=str, default="pretrain", help="The pretrain/finetune, used for foundation model only.")
    p.add_argument("-model", type=str, default="DeepSense", help="The backbone classification model to use.")
    p.add_argument("-model_weight", type=str, default=None, help="Specify the model weight path to evaluate.")
    p.add_argument("-batch_size", type=int, default=None, help="Specify the batch size for training.")
    p.add_argument("-label_ratio", type=float, default=1.0, help="Only used in supervised training or finetune stage.")
    p.add_argument("-gpu", type=str, default="0", help="Specify which GPU to use ('cpu' is rejected: HIP path only).")
    p.add_argument("-tag", type=str, default=None, help="The tag of execution, for record only.")
    p.add_argument("-compute_dtype", type=str, default=None, help="[build extension] bf16 (default) or fp32 matrix-core operands.")
    p.add_argument("-epochs", type=int, default=None, help="[build extension] override train_epochs.")
    p.add_argument("-synthetic_batches", type=int, default=8, help="[build extension] batches per synthetic epoch.")
    p.add_argument("-resume", action="store_true", help="[build extension] continue from the latest weights + optimizer state.")
    p.add_argument("-no_graph", action="store_true", help="[build extension] launch every kernel eagerly instead of replaying the captured step.")
    p.add_argument("-host_draws", action="store_true", help="[build extension] draw the random views on the host (the reference's form) instead of on the device inside the step.")
    p.add_argument("-init_weight", type=str, default=None, help="[build extension] backbone state dict to start pretraining from.")
    p.add_argument("-config", type=str, default=None, help="[build extension] dataset YAML to use instead of ./data/{dataset}.yaml.")
    p.add_argument("-sync_bn", action="store_true", help="[build extension] DeepSense under torchrun: cross-rank BatchNorm statistics.")
    args = p.parse_args()
    args.option = option
    return args
