"""Derived run parameters (reference: params/params_util.py:20-138)."""
import os

import torch

from input_utils.yaml_utils import load_yaml

_FRAMEWORK_MODE = {"FOCAL": "contrastive", "no": "supervised"}
_DEFAULT_TASK = {"ACIDS": "vehicle_classification", "MOD": "vehicle_classification",
                 "RealWorld_HAR": "activity_classification", "PAMAP2": "activity_classification",
                 "HAR4": "activity_classification"}


def select_device(device=""):
    device = str(device).strip().lower().replace("cuda:", "").replace("none", "")
    if device in ("cpu", "mps"):
        raise RuntimeError("this build is the MI355X HIP path of FOCAL pretraining; -gpu=cpu has no implementation here "
                           "(the CPU oracle lives under oracle/ and is test infrastructure only)")
    assert torch.cuda.is_available(), "no ROCm device visible"
    local = int(os.environ.get("LOCAL_RANK", device.split(",")[0] if device else 0))
    torch.cuda.set_device(local)
    return torch.device("cuda", local)


def get_train_mode(learn_framework):
    if learn_framework not in _FRAMEWORK_MODE:
        raise ValueError(f"Invalid learn_framework provided: {learn_framework}")
    return _FRAMEWORK_MODE[learn_framework]


def set_auto_params(args):
    args.device = select_device(str(args.gpu if args.gpu is not None else 0))
    args.half = False
    args.task = _DEFAULT_TASK[args.dataset] if args.task is None else args.task
    here = os.path.dirname(os.path.abspath(__file__))
    args.dataset_config = load_yaml(os.path.join(here, "..", "data", f"{args.dataset}.yaml"))
    args.sequence_sampler = args.learn_framework in {"FOCAL"}
    args.workers = 10
    args.train_mode = get_train_mode(args.learn_framework)
    if args.batch_size is None:
        args.batch_size = 256 if args.stage == "pretrain" else 128
    args.weight_folder = os.path.join(here, "..", "..", "..", "weights", f"{args.dataset}_{args.model}")
    os.makedirs(args.weight_folder, exist_ok=True)
    return args
