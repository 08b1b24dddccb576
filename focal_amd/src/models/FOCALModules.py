"""FOCAL framework wrapper (reference: models/FOCALModules.py): the backbone is run on both augmented views."""
import inspect
import os
import sys

import torch
import torch.nn as nn

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)


class FOCAL(nn.Module):
    def __init__(self, args, backbone):
        super().__init__()
        self.args = args
        self.config = args.dataset_config["FOCAL"]
        self.backbone_config = args.dataset_config[args.model]
        self.modalities = args.dataset_config["modality_names"]
        self.backbone = backbone

    def forward(self, aug_freq_input1, aug_freq_input2, proj_head=False):
        # two backbone calls in program order, as in the reference (:21-34): BatchNorm statistics are per view and the
        # running buffers see view 1 before view 2.  HIP backbones run each modality encoder on its own stream; the
        # streams are joined once here, after both views, so late (small) stages of the two views overlap.
        if getattr(self.backbone, "views_share_pass", False):
            # A backbone without batch statistics (SW_Transformer: LayerNorm only) gives the same features whether the two
            # views are two batches or one batch of 2B; one pass halves the launch count and doubles every kernel's size.
            # DeepSense's BatchNorms compute per-view statistics inside the one batch (views_in_batch).
            both = {loc: {mod: _as_one_batch(x, aug_freq_input2[loc][mod]) for mod, x in mods.items()}
                    for loc, mods in aug_freq_input1.items()}
            # (a backbone with batch statistics -- DeepSense -- is told that the batch is two views: its BatchNorms keep them apart)
            kw = {"views_in_batch": 2} if "views_in_batch" in inspect.signature(self.backbone.forward).parameters else {}
            feats = self.backbone(both, class_head=False, proj_head=proj_head, **kw)
            halves = {m: _SplitHalves.apply(f) for m, f in feats.items()}
            return {m: h[0] for m, h in halves.items()}, {m: h[1] for m, h in halves.items()}
        kw = {}
        if "defer_join" in inspect.signature(self.backbone.forward).parameters:
            kw["defer_join"] = True
        views = "view_index" in inspect.signature(self.backbone.forward).parameters  # the two passes may overlap on separate streams
        mod_features1 = self.backbone(aug_freq_input1, class_head=False, proj_head=proj_head, **kw, **({"view_index": 0} if views else {}))
        mod_features2 = self.backbone(aug_freq_input2, class_head=False, proj_head=proj_head, **kw, **({"view_index": 1} if views else {}))
        if kw:
            from focal_amd import runtime
            runtime.join_all(next(self.backbone.parameters()).device)
        if hasattr(self.backbone, "finish_views"):
            self.backbone.finish_views()
        return mod_features1, mod_features2


def _as_one_batch(x1, x2):
    """The two views as one batch of 2B.  When they already are the two halves of one tensor (the augmenter / DFT wrote them
    there: `ops.fft_realpack(..., out=both[:B])`), that tensor is used as is; otherwise they are concatenated."""
    base = x1._base
    if (base is not None and base is x2._base and base.is_contiguous() and x1.is_contiguous() and x2.is_contiguous()
            and base.shape[0] == 2 * x1.shape[0] and base.shape[1:] == x1.shape[1:] and x1.shape == x2.shape
            and x1.data_ptr() == base.data_ptr() and x2.data_ptr() == base.data_ptr() + x1.numel() * x1.element_size()
            and not (x1.requires_grad or x2.requires_grad)):
        return base
    return torch.cat([x1, x2], dim=0)


class _SplitHalves(torch.autograd.Function):
    """f [2B, E] -> (f[:B], f[B:]).  Backward hands the encoder ONE [2B, E] gradient: when the two incoming gradients already
    sit next to each other in memory (the loss head lays a modality's two views out that way) it is a view of them, no kernel;
    plain slicing would cost two zero-fills, two copies and an add per modality."""

    @staticmethod
    def forward(ctx, f):
        b = f.shape[0] // 2
        return f[:b], f[b:]

    @staticmethod
    def backward(ctx, g1, g2):
        if (g1.is_contiguous() and g2.is_contiguous() and g1.dtype == g2.dtype and g1.shape == g2.shape
                and g1.untyped_storage().data_ptr() == g2.untyped_storage().data_ptr()
                and g2.data_ptr() == g1.data_ptr() + g1.numel() * g1.element_size()):
            return g1.as_strided((2 * g1.shape[0],) + tuple(g1.shape[1:]), g1.stride())
        return torch.cat([g1, g2], dim=0)


def split_features(mod_features):
    """First half of the feature = shared space, second half = private space (reference :37-59)."""
    out = {}
    for mod, f in mod_features.items():
        half = f.shape[-1] // 2
        out[mod] = {"shared": f[..., :half], "private": f[..., half:2 * half]}
    return out
