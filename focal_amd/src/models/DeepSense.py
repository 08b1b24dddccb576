"""DeepSense backbone -- same constructor contract, forward signature, module tree and state_dict as the reference
(models/DeepSense.py), executed on the MI355X HIP kernels: the FOCAL pretraining path (`class_head=False`) and the classifier path
(`class_head=True`: concatenated features -> class layer) of the finetune and supervised stages; multi-location fusion raises."""
import os
import sys

import torch
import torch.nn as nn

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from focal_amd import runtime  # noqa: E402
from focal_amd.backbone import HipBackbone, run_stage  # noqa: E402
from focal_amd.deepsense_engine import DeepSenseModEncoder  # noqa: E402
from focal_amd.head_engine import ClassifierHead  # noqa: E402
from focal_amd.swin_engine import ProjectorHead  # noqa: E402
from models.ConvModules import ConvBlock  # noqa: E402
from models.FusionModules import MeanFusionBlock  # noqa: E402
from models.RecurrentModule import RecurrentBlock  # noqa: E402


class DeepSense(HipBackbone):
    def __init__(self, args) -> None:
        super().__init__()
        self.args = args
        self.config = args.dataset_config["DeepSense"]
        self.device = args.device
        self.modalities = args.dataset_config["modality_names"]
        self.locations = args.dataset_config["location_names"]
        self.multi_location_flag = len(self.locations) > 1
        self.drop_rate = self.config["dropout_ratio"]
        # Data parallel: BatchNorm2d batch statistics are per call (ConvModules.py:86), so the single-device reference at
        # the global batch normalises over ALL windows.  sync_bn=True all-reduces the per-channel sums (2C floats, twice
        # per BN layer per step) and reproduces that exactly; False = per-rank statistics (throughput mode).
        self.sync_bn = bool(getattr(args, "sync_bn", False)) or os.environ.get("FOCAL_SYNC_BN", "0") == "1"
        self._init_hip(args)
        self.init_encoder(args)

    def init_encoder(self, args):
        cfg, dcfg = self.config, args.dataset_config
        if self.multi_location_flag:
            raise NotImplementedError("the MI355X hot path covers single-location datasets (MOD); see DESIGN.md")
        self.loc_mod_extractors = nn.ModuleDict()
        self.geometry = {}
        for loc in self.locations:
            self.loc_mod_extractors[loc] = nn.ModuleDict()
            self.geometry[loc] = {}
            for mod in self.modalities:
                if type(cfg["loc_mod_conv_lens"]) is dict:
                    conv_lens, in_stride = cfg["loc_mod_conv_lens"][mod], cfg["loc_mod_in_conv_stride"][mod]
                else:
                    conv_lens, in_stride = cfg["loc_mod_conv_lens"], 1
                blk = ConvBlock(in_channels=dcfg["loc_mod_in_freq_channels"][loc][mod], out_channels=cfg["loc_mod_out_channels"],
                                in_spectrum_len=dcfg["loc_mod_spectrum_len"][loc][mod], conv_lens=conv_lens,
                                dropout_ratio=cfg["dropout_ratio"], num_inter_layers=cfg["loc_mod_conv_inter_layers"], in_stride=in_stride)
                self.loc_mod_extractors[loc][mod] = blk
                self.geometry[loc][mod] = dict(blk.geometry, H=cfg["recurrent_dim"], n_rnn=cfg["recurrent_layers"])
        self.loc_fusion_layers = nn.ModuleDict()
        self.mod_extractors = nn.ModuleDict()
        for mod in self.modalities:
            self.loc_fusion_layers[mod] = MeanFusionBlock()
            self.mod_extractors[mod] = ConvBlock(in_channels=1, out_channels=cfg["loc_out_channels"], in_spectrum_len=cfg["loc_mod_out_channels"],
                                                 conv_lens=cfg["loc_conv_lens"], dropout_ratio=cfg["dropout_ratio"],
                                                 num_inter_layers=cfg["loc_conv_inter_layers"])
        self.recurrent_layers = nn.ModuleDict()
        for mod in self.modalities:
            self.recurrent_layers[mod] = RecurrentBlock(in_channel=cfg["loc_out_channels"], out_channel=cfg["recurrent_dim"],
                                                        num_layers=cfg["recurrent_layers"], dropout_ratio=cfg["dropout_ratio"])
        out_dim = dcfg["FOCAL"]["emb_dim"]
        self.mod_projectors = nn.ModuleDict()
        for mod in self.modalities:
            self.mod_projectors[mod] = nn.Sequential(nn.Linear(cfg["recurrent_dim"] * 2, out_dim), nn.ReLU(), nn.Linear(out_dim, out_dim))
        self.sample_dim = cfg["recurrent_dim"] * 2 * len(self.modalities)
        n_cls = dcfg[args.task]["num_classes"]
        if args.train_mode == "supervised" or cfg["pretrained_head"] == "linear":
            self.class_layer = nn.Sequential(nn.Linear(self.sample_dim, n_cls))
        else:
            self.class_layer = nn.Sequential(nn.Linear(self.sample_dim, cfg["fc_dim"]), nn.GELU(), nn.Linear(cfg["fc_dim"], n_cls))
        self._encoders = {(loc, mod): DeepSenseModEncoder(self, loc, mod, mi)
                          for loc in self.locations for mi, mod in enumerate(self.modalities)}
        self._heads = {mod: ProjectorHead(self, mod) for mod in self.modalities}
        self._class_head = ClassifierHead(self)
        self._buffers_by_name = None

    @property
    def views_share_pass(self):
        """FOCAL.forward hands the two views to ONE backbone call as a batch of 2B (models/FOCALModules.py) while training on per-rank
        statistics: every BatchNorm then keeps one set of batch statistics per view (focal_amd/deepsense_engine.py).  Evaluation (running
        statistics) and the global-batch -sync_bn mode keep the reference's two calls.  FOCAL_DEEPSENSE_TWO_PASSES=1: two calls always."""
        if os.environ.get("FOCAL_DEEPSENSE_TWO_PASSES", "0") == "1" or not self.training:
            return False
        from focal_amd import ops
        return not (self.sync_bn and ops._sync_world() > 1)

    def bump_bn_counters(self, prefix, n=1):
        """num_batches_tracked += n for every BatchNorm under `prefix` (torch does it per layer in forward, nn/modules/batchnorm.py).
        The counters of one encoder are 0-dim views of ONE int64 tensor, so this is a single launch instead of one per layer;
        state_dict() / load_state_dict() see ordinary per-layer buffers."""
        cache = self.__dict__.setdefault("_bn_counter_blocks", {})
        mods = [m for n, m in self.named_modules() if n.startswith(prefix + ".") and isinstance(m, nn.BatchNorm2d)]
        blk = cache.get(prefix)
        stale = blk is None or any(m.num_batches_tracked.untyped_storage().data_ptr() != blk.untyped_storage().data_ptr() for m in mods)
        if stale:  # first use, or .to(device) replaced the buffers
            blk = torch.stack([m.num_batches_tracked.detach().reshape(()) for m in mods]).contiguous()
            for i, m in enumerate(mods):
                m._buffers["num_batches_tracked"] = blk[i]
            cache[prefix] = blk
            self._buffers_by_name = None
        blk.add_(n)

    def buffer(self, name):
        if self._buffers_by_name is None or self._buffers_by_name.get("__dev") != next(self.parameters()).device:
            self._buffers_by_name = dict(self.named_buffers())
            self._buffers_by_name["__dev"] = next(self.parameters()).device
        return self._buffers_by_name[name]

    def forward_encoder(self, freq_x, class_head=True, proj_head=False, defer_join=False, view_index=None, views_in_batch=1):
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
        # The two views of a step also get their own streams: the GRU sequence
        # kernels are latency-bound launches of 32 workgroups, and view 2's convolution stack fills the chip under view 1's GRU.
        # BatchNorm's running buffers must see view 1 before view 2: an encoder's pass starts after the previous pass of the same
        # encoder has left its convolution stack (the last BatchNorm), see deepsense_engine.forward.
        # (view_index: FOCAL.forward numbers its two backbone calls 0 / 1; any other caller runs one stream per modality)
        view_streams = view_index is not None and self.training
        self.arena()  # built (and its bf16 shadow filled) on the caller's stream before any encoder stream forks from it
        if view_index in (None, 0):
            for enc in self._encoders.values():
                enc.prepare_packs()  # re-ordered weights for both views' passes: one launch per encoder, before the streams fork
        point = runtime.fork_point(dev)  # every encoder starts from here: none waits for the one launched before it
        order = list(range(len(self.modalities)))
        order.sort(key=lambda i: -freq_x[loc][self.modalities[i]].numel())  # the heaviest encoder is enqueued first (see SW_Transformer.forward_encoder)
        for mi in order:
            mod = self.modalities[mi]
            # with one stream per (view, modality) no encoder runs on the caller's stream (index 0): view 2's forks would otherwise
            # wait for the view-1 pass that was enqueued there
            st = runtime.fork_from(dev, (view_index * len(self.modalities) + 1 if view_streams else 0) + mi, point)
            with torch.cuda.stream(st):
                self._encoders[(loc, mod)].pass_order = view_index if view_streams else None
                self._encoders[(loc, mod)].views_in_batch = views_in_batch if self.training else 1
                f = run_stage(self, self._encoders[(loc, mod)], freq_x[loc][mod], view, self.training)
                out[mod] = run_stage(self, self._heads[mod], f) if proj_head else f
                out[mod].record_stream(cur)
        if not defer_join:  # FOCAL.forward joins once after both views so that their encoders overlap
            runtime.join_all(dev)
        return {mod: out[mod] for mod in self.modalities}

    def finish_views(self):
        """After both views' passes of a step (FOCAL.forward): the BatchNorm running-buffer updates the passes recorded (see
        focal_amd/deepsense_engine.py: the passes run side by side, the updates are applied here in the reference's order)."""
        dev = next(self.parameters()).device
        for enc in self._encoders.values():
            enc.finish_views(dev)

    def forward_classifier(self, freq_x):
        """`backbone(freq_x, class_head=True)` -> logits (reference: models/DeepSense.py:154-157).  This is the finetuning path: the encoders in front run
        forward-only (finetuning freezes them, general_utils/weight_utils.py:61-80), the head -- the class layer on the concatenated features -- is one
        differentiable node (focal_amd/head_engine.py)."""
        if self._hot.__name__ == "is_hot":
            raise NotImplementedError("class_head=True needs the classifier head in the parameter arena: build the model with "
                                      "args.stage = 'finetune' (or supervised train_mode)")
        if self.supervised:  # supervised training from scratch (train_utils/supervised_train.py): the gradient flows on into the encoders
            feats = self.forward_encoder(freq_x, class_head=False, proj_head=False)
        else:
            with torch.no_grad():
                feats = self.forward_encoder(freq_x, class_head=False, proj_head=False)
        x = torch.cat([feats[m] for m in self.modalities], dim=1)
        return run_stage(self, self._class_head, x, self.training)

    def forward(self, freq_x, class_head=True, proj_head=False, defer_join=False, view_index=None, views_in_batch=1):
        return self.forward_encoder(freq_x, class_head, proj_head, defer_join, view_index, views_in_batch)
