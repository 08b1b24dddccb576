"""Epoch-level LR schedules (reference: train_utils/lr_scheduler.py:4-48, which uses timm's CosineLRScheduler /
StepLRScheduler).  timm is not a dependency here; the two schedules are closed forms of the epoch index with the
same `.step(epoch)` call convention (the value set after `.step(e)` is the LR of epoch e + 1's batches... exactly
like timm: `step(epoch)` evaluates the schedule AT `epoch` and writes it to the param groups)."""
import math


class _EpochScheduler:
    def __init__(self, optimizer):
        self.optimizer = optimizer
        self.base = [g["lr"] for g in optimizer.param_groups]

    def value(self, base_lr, t):
        raise NotImplementedError

    def _init_warmup(self):
        """timm writes `warmup_lr_init` to the param groups at construction when warm-up is on: epoch 0 runs at it."""
        if getattr(self, "warmup_t", 0) > 0:
            for g in self.optimizer.param_groups:
                g["lr"] = self.warmup_lr_init

    def step(self, epoch):
        for g, b in zip(self.optimizer.param_groups, self.base):
            g["lr"] = self.value(b, epoch)


class CosineLRScheduler(_EpochScheduler):
    """lr_min + (lr0 - lr_min)/2 (1 + cos(pi t / t_initial)), one cycle (cycle_limit=1), optional linear warm-up."""

    def __init__(self, optimizer, t_initial, lr_min=0.0, warmup_t=0, warmup_lr_init=0.0, warmup_prefix=False, **_):
        super().__init__(optimizer)
        self.t_initial, self.lr_min = t_initial, lr_min
        self.warmup_t, self.warmup_lr_init, self.warmup_prefix = warmup_t, warmup_lr_init, warmup_prefix
        self._init_warmup()

    def value(self, base_lr, t):
        if t < self.warmup_t:
            return self.warmup_lr_init + t * (base_lr - self.warmup_lr_init) / self.warmup_t
        if self.warmup_prefix:
            t = t - self.warmup_t
        if t >= self.t_initial:
            return self.lr_min
        return self.lr_min + 0.5 * (base_lr - self.lr_min) * (1 + math.cos(math.pi * t / self.t_initial))


class StepLRScheduler(_EpochScheduler):
    def __init__(self, optimizer, decay_t, decay_rate=1.0, warmup_t=0, warmup_lr_init=0.0, **_):
        super().__init__(optimizer)
        self.decay_t, self.decay_rate, self.warmup_t, self.warmup_lr_init = decay_t, decay_rate, warmup_t, warmup_lr_init
        self._init_warmup()

    def value(self, base_lr, t):
        if t < self.warmup_t:
            return self.warmup_lr_init + t * (base_lr - self.warmup_lr_init) / self.warmup_t
        return base_lr * self.decay_rate ** (t // self.decay_t)


def define_lr_scheduler(args, optimizer):
    if args.train_mode in {"supervised"}:
        oc, sc = args.dataset_config[args.model]["optimizer"], args.dataset_config[args.model]["lr_scheduler"]
    elif args.stage == "pretrain":
        oc = args.dataset_config[args.learn_framework]["pretrain_optimizer"]
        sc = args.dataset_config[args.learn_framework]["pretrain_lr_scheduler"]
    elif args.stage == "finetune":
        oc = args.dataset_config[args.learn_framework]["finetune_optimizer"]
        sc = args.dataset_config[args.learn_framework]["finetune_lr_scheduler"]
    else:
        raise Exception(f"Mode: {args.train_mode} and stage: {args.stage} not defined.")
    if sc["name"] == "cosine":
        t_initial = sc["train_epochs"] - sc["warmup_epochs"] if sc["warmup_prefix"] else sc["train_epochs"]
        return CosineLRScheduler(optimizer, t_initial=t_initial, lr_min=oc["min_lr"], warmup_lr_init=oc["warmup_lr"],
                                 warmup_t=sc["warmup_epochs"], warmup_prefix=sc["warmup_prefix"])
    if sc["name"] == "step":
        return StepLRScheduler(optimizer, decay_t=sc["decay_epochs"], decay_rate=sc["decay_rate"],
                               warmup_lr_init=oc["warmup_lr"], warmup_t=sc["warmup_epochs"])
    raise Exception(f"Unknown LR scheduler: {sc['name']}")
