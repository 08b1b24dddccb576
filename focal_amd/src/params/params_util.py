"""Derived run parameters (reference: params/params_util.py:20-138)."""
import datetime
import os

import torch

from input_utils.yaml_utils import load_yaml

_FRAMEWORK_MODE = {"FOCAL": "contrastive", "no": "supervised"}
_DEFAULT_TASK = {"ACIDS": "vehicle_classification", "MOD": "vehicle_classification",
                 "RealWorld_HAR": "activity_classification", "PAMAP2": "activity_classification",
                 "HAR4": "activity_classification"}


def parse_device_list(device=""):
    """`-gpu` as the reference parses it (params/params_util.py:20-55): "0", "cuda:0", "0,1,2,3" -> list of device indices."""
    device = str(device).strip().lower().replace("cuda:", "").replace("none", "")
    if device in ("cpu", "mps"):
        raise RuntimeError("this build is the MI355X HIP path of FOCAL pretraining; -gpu=cpu has no implementation here "
                           "(the CPU oracle lives under oracle/ and is test infrastructure only)")
    return [int(x) for x in device.split(",") if x != ""] or [0]


def local_device_index(device=""):
    """Device index of THIS process: LOCAL_RANK under a data-parallel launch (torchrun, or train.py's own `-gpu=0,1,...`
    spawn, which narrows HIP_VISIBLE_DEVICES to the listed devices), else the first listed device.  FOCAL_DIST_ONE_DEVICE=1
    (tests on a 1-GPU box) puts every rank on device 0."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return 0 if os.environ.get("FOCAL_DIST_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    return parse_device_list(device)[0]


def init_distributed(device=""):
    """One process per GPU: under WORLD_SIZE > 1 create the process group BEFORE the first HIP call, bound to this
    rank's device (backend "nccl" = RCCL over xGMI; FOCAL_DIST_BACKEND=gloo for CPU-side / single-GPU tests)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("FOCAL_DIST_BACKEND", "nccl")
    if backend == "nccl":
        # (rank 0 validates / fits the KNN estimator / writes checkpoints alone every 10 epochs while the others wait at a barrier)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_device_index(device)), timeout=datetime.timedelta(minutes=60))
    else:
        dist.init_process_group(backend, timeout=datetime.timedelta(minutes=60))


def select_device(device=""):
    local = local_device_index(device)
    init_distributed(device)
    assert torch.cuda.is_available(), "no ROCm device visible"
    torch.cuda.set_device(local)
    return torch.device("cuda", local)


def get_train_mode(learn_framework):
    if learn_framework not in _FRAMEWORK_MODE:
        raise ValueError(f"Invalid learn_framework provided: {learn_framework}")
    return _FRAMEWORK_MODE[learn_framework]


def set_auto_params(args):
    args.device = select_device(str(args.gpu if args.gpu is not None else 0))
    args.world_size = int(os.environ.get("WORLD_SIZE", "1"))
    args.rank = int(os.environ.get("RANK", "0")) if args.world_size > 1 else 0
    args.half = False
    args.task = _DEFAULT_TASK[args.dataset] if args.task is None else args.task
    here = os.path.dirname(os.path.abspath(__file__))
    args.dataset_config = load_yaml(getattr(args, "config", None) or os.path.join(here, "..", "data", f"{args.dataset}.yaml"))
    args.sequence_sampler = args.learn_framework in {"FOCAL"}
    args.workers = 10
    args.train_mode = get_train_mode(args.learn_framework)
    if args.batch_size is None:
        args.batch_size = 256 if args.stage == "pretrain" else 128
    args.weight_folder = os.path.join(here, "..", "..", "..", "weights", f"{args.dataset}_{args.model}")
    os.makedirs(args.weight_folder, exist_ok=True)
    return args
