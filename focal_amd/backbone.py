"""Shared plumbing of the two backbones: arena ownership, operand dtype, autograd glue."""
import torch
import torch.nn as nn

from . import runtime
from .arena import ParamArena

DEAD_PREFIXES = ("patch_embed.", "class_layer.", "mod_fusion_layers.", "absolute_pos_embed.", "mod_extractors.",
                 "loc_fusion_layers.", "loc_context_layers.", "loc_fusion_layer.")


HEAD_PREFIXES = ("class_layer.", "mod_fusion_layers.")


def is_hot(name):
    """Parameters that receive a gradient in FOCAL pretraining."""
    return not name.startswith(DEAD_PREFIXES)


def is_hot_with_head(name):
    """Classifier / finetune stage: the head (class layer, modality fusion) joins the arena; it is what finetuning trains
    (general_utils/weight_utils.py:61-80), the encoder weights stay there as the frozen operands of the forward pass."""
    return is_hot(name) or name.startswith(HEAD_PREFIXES)


SUPERVISED_DEAD = ("absolute_pos_embed.", "mod_extractors.", "loc_fusion_layers.", "loc_context_layers.", "loc_fusion_layer.", "mod_projectors.")


def is_hot_supervised(name):
    """Supervised training from scratch (train_utils/supervised_train.py: every parameter is handed to the optimizer): everything
    `backbone(freq_x, class_head=True)` touches gets a gradient -- the patch embedding included (it is frozen only in FOCAL
    pretraining) -- and the projection heads of the contrastive path do not (torch skips their `grad is None`)."""
    return not name.startswith(SUPERVISED_DEAD)


class HipBackbone(nn.Module):
    def _init_hip(self, args):
        self.compute_dtype = runtime.compute_dtype_from(args)
        self.supervised = getattr(args, "train_mode", "") == "supervised"
        self._hot = is_hot_supervised if self.supervised else (is_hot_with_head if getattr(args, "stage", "pretrain") == "finetune" else is_hot)
        self._arena = None
        self._named = None
        self._fwd_calls = 0
        self.register_load_state_dict_post_hook(lambda module, keys: module._after_load())

    def _after_load(self):
        if self._arena is not None and self._arena.intact():
            self._arena.sync_shadow(force=True)

    def arena(self):
        if self._arena is None or not self._arena.intact():
            self._arena = ParamArena(self, self._hot, self.compute_dtype)
            self._named = None
        self._arena.sync_shadow()
        return self._arena

    def param(self, name):
        if self._named is None:
            self._named = dict(self.named_parameters())
        return self._named[name]

    def rng_state(self):
        return runtime.rng_state(next(self.parameters()).device)

    def _anchor(self, device):
        # torch.autograd only schedules a node whose inputs need grad; parameter gradients are written straight into
        # the arena by the HIP kernels, so a dummy differentiable input keeps each encoder node alive.
        return torch.zeros(1, device=device, requires_grad=True)


class StageFn(torch.autograd.Function):
    """One autograd node around an engine object exposing forward(x, ...)->(y, saved) / backward(saved, dy)->dx|None."""

    @staticmethod
    def forward(ctx, anchor, x, engine, args):
        # x may have been produced on another stream (encoders run on side streams): keep its block alive for us
        x.record_stream(torch.cuda.current_stream(x.device))
        y, saved = engine.forward(x, *args)
        ctx.engine, ctx.saved = engine, saved
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dy.record_stream(torch.cuda.current_stream(dy.device))  # produced by the loss head on the caller's stream
        dx = ctx.engine.backward(ctx.saved, dy)
        ctx.saved = None
        return None, dx, None, None


def run_stage(backbone, engine, x, *args):
    if torch.is_grad_enabled():
        return StageFn.apply(backbone._anchor(x.device), x, engine, args)
    x.record_stream(torch.cuda.current_stream(x.device))
    y, _ = engine.forward(x, *args)
    return y
