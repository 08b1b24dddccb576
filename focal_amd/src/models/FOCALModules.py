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
            both = {loc: {mod: torch.cat([x, aug_freq_input2[loc][mod]], dim=0) for mod, x in mods.items()}
                    for loc, mods in aug_freq_input1.items()}
            feats = self.backbone(both, class_head=False, proj_head=proj_head)
            first = next(iter(next(iter(aug_freq_input1.values())).values()))
            b = first.shape[0]
            return {m: f[:b] for m, f in feats.items()}, {m: f[b:] for m, f in feats.items()}
        kw = {}
        if "defer_join" in inspect.signature(self.backbone.forward).parameters:
            kw["defer_join"] = True
        mod_features1 = self.backbone(aug_freq_input1, class_head=False, proj_head=proj_head, **kw)
        mod_features2 = self.backbone(aug_freq_input2, class_head=False, proj_head=proj_head, **kw)
        if kw:
            from focal_amd import runtime
            runtime.join_all(next(self.backbone.parameters()).device)
        return mod_features1, mod_features2


def split_features(mod_features):
    """First half of the feature = shared space, second half = private space (reference :37-59)."""
    out = {}
    for mod, f in mod_features.items():
        half = f.shape[-1] // 2
        out[mod] = {"shared": f[..., :half], "private": f[..., half:2 * half]}
    return out
