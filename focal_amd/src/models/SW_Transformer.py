"""SW_Transformer backbone -- same constructor contract, forward signature, module tree and state_dict as the
reference (models/SW_Transformer.py), executed on the MI355X HIP kernels.

`forward(freq_x, class_head=False, proj_head=...)` is the FOCAL pretraining path (reference :210-268, :294-304).
The classifier path (`class_head=True`: TransformerFusionBlock over the modality tokens + class layer, reference :244-276) runs for
the finetune stage (frozen encoders, head trained) and for supervised training from scratch (gradient flows on into the encoders and
the patch embedding); multi-location fusion raises.
"""
import os
import sys

import torch
import torch.nn as nn

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from focal_amd import runtime  # noqa: E402
from focal_amd.backbone import HipBackbone, run_stage  # noqa: E402
from focal_amd.head_engine import ClassifierHead  # noqa: E402
from focal_amd.swin_engine import ProjectorHead, SwinModEncoder  # noqa: E402
from input_utils.padding_utils import get_padded_size  # noqa: E402
from models.FusionModules import TransformerFusionBlock  # noqa: E402
from models.SwinModules import BasicLayer, PatchEmbed, PatchMerging  # noqa: E402


class SW_Transformer(HipBackbone):
    def __init__(self, args) -> None:
        super().__init__()
        self.args = args
        self.config = args.dataset_config["SW_Transformer"]
        self.modalities = args.dataset_config["modality_names"]
        self.locations = args.dataset_config["location_names"]
        self.num_segments = args.dataset_config["num_segments"]
        self.drop_rate = self.config["dropout_ratio"]
        self.attn_drop_rate = self.config["attn_drop_rate"]
        self.norm_layer = nn.LayerNorm
        # No batch statistics anywhere in this backbone: FOCAL may run both views as one batch of 2B
        # (FOCALModules.FOCAL.forward).  Measured r1: 9.48 ms (two passes on 4 streams) -> 9.12 ms (one pass, 2 streams);
        # cutting the heavier modality back into batch chunks to balance the streams was slower again (9.34 ms).
        # (one pass per view on four streams -- halved kernels, doubled launches -- cost 14 %: DESIGN 4; the switch was removed in round 4)
        self.views_share_pass = True
        self._init_hip(args)
        self.init_encoder()

    def init_encoder(self) -> None:
        cfg, dcfg = self.config, self.args.dataset_config
        if len(self.locations) != 1:
            raise NotImplementedError("the MI355X hot path covers single-location datasets (MOD); see DESIGN.md")
        self.freq_interval_layers = nn.ModuleDict()
        self.patch_embed = nn.ModuleDict()
        self.absolute_pos_embed = nn.ModuleDict()
        self.mod_patch_embed = nn.ModuleDict()
        self.mod_in_layers = nn.ModuleDict()
        self.geometry, self.block_windows, self.drop_path_rates = {}, {}, {}
        c0 = cfg["time_freq_out_channels"]
        for loc in self.locations:
            self.freq_interval_layers[loc] = nn.ModuleDict()
            self.patch_embed[loc] = nn.ModuleDict()
            self.absolute_pos_embed[loc] = nn.ParameterDict()
            self.mod_in_layers[loc] = nn.ModuleDict()
            self.geometry[loc] = {}
            for mod in self.modalities:
                stride = cfg["in_stride"][mod]
                if stride != 1:
                    raise NotImplementedError("in_stride != 1 is not used by any shipped config")
                spectrum = dcfg["loc_mod_spectrum_len"][loc][mod]
                window, patch = list(cfg["window_size"][mod]), list(cfg["patch_size"]["freq"][mod])
                depths = list(cfg["time_freq_block_num"][mod])
                padded = get_padded_size((self.num_segments, spectrum // stride), window, patch, len(depths))
                pe = PatchEmbed(img_size=padded, patch_size=patch, in_chans=dcfg["loc_mod_in_freq_channels"][loc][mod] * stride,
                                embed_dim=c0, norm_layer=self.norm_layer)
                self.patch_embed[loc][mod] = pe
                grid = pe.patches_resolution
                self.absolute_pos_embed[loc][mod] = nn.Parameter(torch.zeros(1, pe.num_patches, c0))
                nn.init.trunc_normal_(self.absolute_pos_embed[loc][mod], std=0.02)
                dpr = [x.item() for x in torch.linspace(0, cfg["drop_path_rate"], sum(depths))]
                self.drop_path_rates[mod] = dpr
                layers, stages = nn.ModuleList(), []
                for i, depth in enumerate(depths):
                    res, dim = (grid[0] // 2 ** i, grid[1] // 2 ** i), c0 * 2 ** i
                    layer = BasicLayer(dim=dim, input_resolution=res, num_heads=cfg["time_freq_head_num"],
                                       window_size=list(window), depth=depth, drop=self.drop_rate,
                                       attn_drop=self.attn_drop_rate, drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])],
                                       norm_layer=self.norm_layer, downsample=PatchMerging if i < len(depths) - 1 else None)
                    layers.append(layer)
                    stages.append(dict(H=res[0], W=res[1], C=dim, depth=depth, downsample=i < len(depths) - 1))
                    for bi, blk in enumerate(layer.blocks):
                        self.block_windows[(loc, mod, i, bi)] = (*blk.window_size, *blk.shift_size)
                self.freq_interval_layers[loc][mod] = layers
                last = stages[-1]
                self.mod_in_layers[loc][mod] = nn.Linear(last["H"] * last["W"] * last["C"], cfg["loc_out_channels"])
                self.geometry[loc][mod] = dict(grid=grid, patch=patch, window=window, stages=stages,
                                               heads=cfg["time_freq_head_num"], pad_img=padded)
                if spectrum // patch[1] > grid[1] or self.num_segments > grid[0]:
                    raise ValueError("padded patch grid smaller than the input")
        out_dim = dcfg["FOCAL"]["emb_dim"]
        self.mod_projectors = nn.ModuleDict()
        for mod in self.modalities:
            self.mod_projectors[mod] = nn.Sequential(nn.Linear(cfg["loc_out_channels"], out_dim), nn.ReLU(),
                                                     nn.Linear(out_dim, out_dim))
        self.mod_fusion_layers = TransformerFusionBlock(cfg["loc_out_channels"], cfg["loc_head_num"],
                                                        cfg["dropout_ratio"], cfg["dropout_ratio"])
        self.sample_dim = cfg["loc_out_channels"]
        n_cls = dcfg[self.args.task]["num_classes"]
        if self.args.train_mode == "supervised" or cfg["pretrained_head"] == "linear":
            self.class_layer = nn.Sequential(nn.Linear(self.sample_dim, n_cls))
        else:
            self.class_layer = nn.Sequential(nn.Linear(self.sample_dim, cfg["fc_dim"]), nn.GELU(), nn.Linear(cfg["fc_dim"], n_cls))
        self._encoders = {(loc, mod): SwinModEncoder(self, loc, mod, mi)
                          for loc in self.locations for mi, mod in enumerate(self.modalities)}
        self._heads = {mod: ProjectorHead(self, mod) for mod in self.modalities}
        self._class_head = ClassifierHead(self, "mod_fusion_layers", cfg["loc_head_num"], cfg["dropout_ratio"])

    def forward_encoder(self, freq_x, class_head=True, proj_head=False, defer_join=False):
        if class_head:
            return self.forward_classifier(freq_x)
        loc = self.locations[0]
        view = self._fwd_calls
        self._fwd_calls = (self._fwd_calls + 1) & 0xFFFF
        # one HIP stream per modality encoder (see focal_amd/runtime.py: side streams); joined before returning
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            from focal_amd._lib import FocalHipError
            raise FocalHipError("the FOCAL HIP path needs the model on a ROCm device (no CPU fallback)")
        cur = torch.cuda.current_stream(dev)
        out = {}
        # The parameter arena (and its bf16 shadow) must exist BEFORE the fork point: built lazily inside the first encoder's pass it was
        # filled on that encoder's stream AFTER the point the other encoders start from, and on a fresh model they could read a zero arena
        # (round 5: test_train_step_loss_and_gradients[bf16] saw an all-zero seismic embedding in 2 of 16 fresh processes).
        self.arena()
        point = runtime.fork_point(dev)  # every encoder starts from here: none waits for the one launched before it
        # The heaviest modality is enqueued first: the order of enqueueing is the order of the nodes in the captured step, and what the
        # runtime dispatches first gets a head start on the stream that ends the step (audio has 2/3 of the MOD step's work).  Autograd
        # then reaches the heaviest encoder's backward pass last; that order was measured not to matter.  +1.4 % on the step
        # (profiles/r3_heavy_first_ab.txt).
        order = list(range(len(self.modalities)))
        tokens = [self.geometry[loc][m]["stages"][0]["H"] * self.geometry[loc][m]["stages"][0]["W"] for m in self.modalities]
        order.sort(key=lambda i: -tokens[i])  # (stable: equal modalities keep the configuration order)
        for mi in order:
            mod = self.modalities[mi]
            # stream per modality; with one backbone pass per view the two passes of a step alternate between two sets of streams so
            # that they overlap.  One pass per step (views_share_pass) always uses the same set -- the first modality stays on the
            # caller's stream: alternating there only doubled the streams every join has to wait for (-1.6 % on the step,
            # profiles/r3_encoder_streams_ab.txt)
            slot = 0 if self.views_share_pass else view % 2
            st = runtime.fork_from(dev, slot * len(self.modalities) + mi, point)
            with torch.cuda.stream(st):
                f = run_stage(self, self._encoders[(loc, mod)], freq_x[loc][mod], view, self.training)
                out[mod] = run_stage(self, self._heads[mod], f) if proj_head else f
                out[mod].record_stream(cur)
        if not defer_join:  # FOCAL.forward joins once after both views so that their encoders overlap
            runtime.join_all(dev)
        return {mod: out[mod] for mod in self.modalities}

    def forward_classifier(self, freq_x):
        """`backbone(freq_x, class_head=True)` -> logits (reference: models/SW_Transformer.py:269-276).  This is the finetuning path: the encoders in front run
        forward-only (finetuning freezes them, general_utils/weight_utils.py:61-80), the head -- modality fusion + class layer -- is one
        differentiable node (focal_amd/head_engine.py)."""
        if self._hot.__name__ == "is_hot":
            raise NotImplementedError("class_head=True needs the classifier head in the parameter arena: build the model with "
                                      "args.stage = 'finetune' (or supervised train_mode)")
        if self.supervised:  # supervised training from scratch (train_utils/supervised_train.py): the gradient flows on into the encoders
            feats = self.forward_encoder(freq_x, class_head=False, proj_head=False)
        else:
            with torch.no_grad():
                feats = self.forward_encoder(freq_x, class_head=False, proj_head=False)
        x = torch.stack([feats[m] for m in self.modalities], dim=1)  # [b, M, c] (the reference's [b, 1, M, c] with i = 1)
        return run_stage(self, self._class_head, x, self.training)

    def forward(self, freq_x, class_head=True, proj_head=False, defer_join=False):
        return self.forward_encoder(freq_x, class_head, proj_head, defer_join)
