"""Batches of whole subsequences for FOCAL pretraining.

The reference loads one `.pt` dict per window through a torch DataLoader with a sequence-aware batch sampler
(input_utils/multi_modal_dataloader.py:12-78, multi_modal_dataset.py:58-130).  Feeding 1e4-1e5 windows/s from
per-sample files is a separate "next" component; this build ships the synthetic equivalent the
benchmark and tests use: seeded N(0,1) time-domain windows in the reference's collated layout
`({loc: {mod: [B, c, i, s]}}, labels)`, batch = `batch_size // seq_len` subsequences of `seq_len` windows.
"""
import torch


class SyntheticSequenceLoader:
    def __init__(self, args, batch_size, num_batches=8, seed=1234, device=None):
        cfg = args.dataset_config
        self.cfg, self.batch_size, self.num_batches, self.seed = cfg, batch_size, num_batches, seed
        self.device = device or args.device
        self.task = getattr(args, "task", None)
        seq = cfg["seq_len"]
        if batch_size % seq != 0:
            raise ValueError(f"batch size {batch_size} must hold whole subsequences of {seq} windows (models/loss.py:152-155)")

    def __len__(self):
        return self.num_batches

    def __iter__(self):
        cfg = self.cfg
        for k in range(self.num_batches):
            g = torch.Generator(device="cpu").manual_seed(self.seed + k)
            batch = {}
            for loc in cfg["location_names"]:
                batch[loc] = {}
                for mod in cfg["modality_names"]:
                    shape = (self.batch_size, cfg["loc_mod_in_time_channels"][loc][mod], cfg["num_segments"],
                             cfg["loc_mod_spectrum_len"][loc][mod])
                    batch[loc][mod] = torch.randn(shape, generator=g)
            n_cls = cfg.get(getattr(self, "task", None) or "vehicle_classification", {}).get("num_classes", 2)
            labels = torch.randint(0, n_cls, (self.batch_size,), generator=g)  # synthetic class labels (KNN validation needs > 1 class)
            yield batch, labels


def create_dataloader(option, args, batch_size=64, workers=5):
    index_file = args.dataset_config.get("pretrain_index_file", "synthetic")
    if index_file != "synthetic":
        raise NotImplementedError("per-sample .pt loading is not part of this round's hot path; set "
                                  "pretrain_index_file: \"synthetic\" (see DESIGN.md, 'what comes next')")
    n = getattr(args, "synthetic_batches", 8)
    return SyntheticSequenceLoader(args, batch_size, num_batches=n if option == "train" else 1, seed=1234 if option == "train" else 99)
