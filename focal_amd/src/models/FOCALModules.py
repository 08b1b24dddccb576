"""FOCAL framework wrapper (reference: models/FOCALModules.py): the backbone is run on both augmented views."""
import torch.nn as nn


class FOCAL(nn.Module):
    def __init__(self, args, backbone):
        super().__init__()
        self.args = args
        self.config = args.dataset_config["FOCAL"]
        self.backbone_config = args.dataset_config[args.model]
        self.modalities = args.dataset_config["modality_names"]
        self.backbone = backbone

    def forward(self, aug_freq_input1, aug_freq_input2, proj_head=False):
        # two sequential calls, as in the reference (:21-34): BatchNorm statistics are per view
        mod_features1 = self.backbone(aug_freq_input1, class_head=False, proj_head=proj_head)
        mod_features2 = self.backbone(aug_freq_input2, class_head=False, proj_head=proj_head)
        return mod_features1, mod_features2


def split_features(mod_features):
    """First half of the feature = shared space, second half = private space (reference :37-59)."""
    out = {}
    for mod, f in mod_features.items():
        half = f.shape[-1] // 2
        out[mod] = {"shared": f[..., :half], "private": f[..., half:2 * half]}
    return out
