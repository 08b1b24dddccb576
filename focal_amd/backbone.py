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
        # data parallel: encoders stop their backward pass where most gradient bytes are final (focal_amd/graph_step.py issues the first
        # all-reduce bucket there) and park the rest here for backward_continue()
        self.split_backward = False
        self.pending_backward = []
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

    def backward_continue(self):
        """The parked second halves of the encoders' backward passes (split_backward), each on the stream its first half ran on,
        all started from one fork point and joined back before returning."""
        pend, self.pending_backward = self.pending_backward, []
        if not pend:
            return
        dev = pend[0][2].device
        cur = torch.cuda.current_stream(dev)
        point = runtime.fork_point(dev)
        with torch.no_grad():
            for engine, saved, st in pend:
                if st != cur:
                    st.wait_event(point)
                with torch.cuda.stream(st):
                    engine.backward_rest(saved)
        runtime.join_all(dev)

    def final_after_first_phase(self):
        """Names of the arena parameters whose gradients are final when the first phase of a split backward pass ends: everything but
        the encoder stages in front of the last one (their blocks and the PatchMerging that follows them)."""
        import re
        ar = self.arena()
        stage_of = {n: re.match(r"(freq_interval_layers\.[^.]+\.[^.]+)\.(\d+)\.", n) for n in ar.index}
        last = {}
        for m in stage_of.values():
            if m:
                last[m.group(1)] = max(last.get(m.group(1), 0), int(m.group(2)))
        return [n for n, m in stage_of.items() if m is None or int(m.group(2)) == last[m.group(1)]]

    def rng_state(self):
        return runtime.rng_state(next(self.parameters()).device)

    def _anchor(self, device):
        # torch.autograd only schedules a node whose inputs need grad; parameter gradients are written straight into
        # the arena by the HIP kernels, so a dummy differentiable input keeps each encoder node alive.
        return runtime.anchor(device)  # (one cached leaf per device: a fresh torch.zeros per stage call was a fill launch in the step)


def _join_after_backward(device):
    """When the autograd engine has run every node of this backward pass, the thread that called backward() makes its current stream
    wait for all encoder side streams (runtime.join_all: event waits).  Queued by every stage node: a few redundant waits per pass, no
    state that an exception inside a backward pass could leave behind."""
    key = torch.device(device)
    torch.autograd.Variable._execution_engine.queue_callback(lambda: runtime.join_all(key))


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
        # The parameter gradients were written straight into the arena on THIS stream (the stage's forward stream).  Autograd orders the
        # caller's stream behind the streams of the leaves that received a gradient through it -- none here (the anchor gets None) -- so a
        # caller that reads p.grad right after backward() raced the last kernels of a side-stream encoder (round 5).  The optimizer joins
        # every side stream itself; a callback at the END of the backward pass does it for everybody else: an event wait, no kernel.  (Not
        # a wait per stage: the caller's stream carries the first modality's own backward pass, which would then queue behind every other
        # modality's -- measured -7 % on the four-modality HAR4 step.)
        _join_after_backward(dy.device)
        return None, dx, None, None


def run_stage(backbone, engine, x, *args):
    if torch.is_grad_enabled():
        return StageFn.apply(backbone._anchor(x.device), x, engine, args)
    x.record_stream(torch.cuda.current_stream(x.device))
    y, _ = engine.forward(x, *args)
    return y
