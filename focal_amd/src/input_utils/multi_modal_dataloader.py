"""Batches of whole subsequences for FOCAL pretraining.

The reference loads one `.pt` dict per window through a torch DataLoader with a sequence-aware batch sampler
(input_utils/multi_modal_dataloader.py:12-78, multi_modal_dataset.py:58-130); `create_dataloader` keeps that path, adds a
packed-shard loader for feeding 1e4-1e5 windows/s (input_utils/packed_shards.py) and the synthetic source
the benchmark and tests use: seeded N(0,1) time-domain windows in the reference's collated layout
`({loc: {mod: [B, c, i, s]}}, labels)`, batch = `batch_size // seq_len` subsequences of `seq_len` windows.
"""
import torch


class SyntheticSequenceLoader:
    def __init__(self, args, batch_size, num_batches=8, seed=1234, device=None):
        cfg = args.dataset_config
        self.cfg, self.batch_size, self.num_batches, self.seed = cfg, batch_size, num_batches, seed
        self.device = device or args.device
        # every epoch yields the same seeded batches: they are generated once and kept (on the device when it is a GPU -- the data is
        # synthetic, there is nothing to load -- so that an epoch measures the training step, not torch.randn on the host)
        self.resident = getattr(args, "synthetic_resident", True) and torch.device(self.device).type == "cuda"
        self._cache = None
        self.task = getattr(args, "task", None)
        seq = cfg["seq_len"]
        if batch_size % seq != 0:
            raise ValueError(f"batch size {batch_size} must hold whole subsequences of {seq} windows (models/loss.py:152-155)")

    def __len__(self):
        return self.num_batches

    def __iter__(self):
        if self._cache is None:
            self._cache = []
            for batch, labels in self._generate():
                if self.resident:
                    batch = {loc: {mod: t.to(self.device) for mod, t in mods.items()} for loc, mods in batch.items()}
                self._cache.append((batch, labels))
        return iter(self._cache)

    def _generate(self):
        cfg = self.cfg
        rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
        for k in range(self.num_batches):
            # data parallel: `batch_size` is the per-rank share and every rank draws its own windows (bench.py's convention)
            g = torch.Generator(device="cpu").manual_seed(self.seed + k + 100003 * rank)
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


def rank_share(chunk, world, rank):
    """This rank's part of one global batch of subsequence ids.  Every rank takes the SAME number of subsequences (the
    all-gather of the embeddings and the gradient all-reduce need equal shapes and an equal number of steps on all ranks):
    len(chunk) // world each, in rank order; a remainder (only the last batch of an epoch can have one) is dropped, and a batch
    that would leave the global batch with fewer than the 2 subsequences the loss needs is skipped by every rank."""
    if world == 1:
        return chunk
    per = len(chunk) // world
    if per * world < 2:
        return []
    return chunk[rank * per:(rank + 1) * per]


class BatchSeqSampler(torch.utils.data.Sampler):
    """Batches of whole subsequences in one shuffled order per epoch (reference :51-78).  Under torch.distributed, `batch_size`
    is the GLOBAL batch: all ranks shuffle with the same seeded generator and each yields its `rank_share` of every batch."""

    def __init__(self, args, batch_size, dataset, seed=0, shard=True):
        """shard=False: whole batches on whichever rank iterates (validation / test sets: rank 0 evaluates them alone)."""
        self.dataset = dataset
        self.subseq_batch_size = batch_size // args.dataset_config["seq_len"]
        self.subseq_count = len(dataset.subseqs)
        self.subseq_indices = list(range(self.subseq_count))
        dist = torch.distributed
        self.world = dist.get_world_size() if (shard and dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.seed, self.epoch = seed, 0

    def set_epoch(self, epoch):
        """The training epoch the next pass belongs to (the rank-consistent shuffle is seeded by it): passes that are not training
        epochs -- the KNN fit over the training set -- and a resumed run then do not shift the sequence of orders."""
        self.epoch, self._epoch_set = int(epoch), True

    def _batches(self):
        import random
        if self.world > 1:
            order = list(range(self.subseq_count))
            random.Random(self.seed + self.epoch).shuffle(order)
        else:
            random.shuffle(self.subseq_indices)
            order = self.subseq_indices
        for b in range(0, self.subseq_count, self.subseq_batch_size):
            mine = rank_share(order[b:b + self.subseq_batch_size], self.world, self.rank)
            if mine:
                yield mine

    def __iter__(self):
        for mine in self._batches():
            out = []
            for sid in mine:
                out.extend(self.dataset.subseq_to_sample_idx[self.dataset.subseqs[sid]])
            yield out
        if not getattr(self, "_epoch_set", False):
            self.epoch += 1

    def __len__(self):
        if self.world == 1:
            return -(-self.subseq_count // self.subseq_batch_size)
        full, rem = divmod(self.subseq_count, self.subseq_batch_size)
        return full + (1 if (rem // self.world) * self.world >= 2 else 0)


def _index_file(option, args):
    if option == "train":
        if args.train_mode not in {"supervised"} and args.stage == "pretrain":
            return args.dataset_config["pretrain_index_file"]
        return args.dataset_config[args.task]["train_index_file"]
    return args.dataset_config[args.task]["val_index_file" if option == "val" else "test_index_file"]


def create_dataloader(option, args, batch_size=64, workers=5):
    """reference :12-48.  Index file "synthetic" -> seeded synthetic windows; a directory holding a packed shard
    (input_utils/packed_shards.py: pack_index) -> the prefetching packed loader; otherwise the reference's per-sample `.pt`
    files through a torch DataLoader, with the sequence-aware batch sampler for contrastive pretraining."""
    index_file = _index_file(option, args)
    if index_file == "synthetic":
        n = getattr(args, "synthetic_batches", 8)
        world = torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1
        if option == "train" and world > 1:  # -batch_size is the global batch, as for the file-backed loaders
            if batch_size % (world * args.dataset_config["seq_len"]) != 0:
                raise ValueError(f"global batch {batch_size} does not split into whole subsequences over {world} ranks")
            batch_size //= world
        return SyntheticSequenceLoader(args, batch_size, num_batches=n if option == "train" else 1, seed=1234 if option == "train" else 99)
    import os
    if os.path.isdir(index_file):
        from input_utils.packed_shards import PackedSequenceLoader
        return PackedSequenceLoader(args, index_file, batch_size, shuffle=(option == "train"), shard=(option == "train"))
    from torch.utils.data import DataLoader
    from input_utils.multi_modal_dataset import MultiModalDataset, MultiModalSequenceDataset
    if args.sequence_sampler and args.train_mode == "contrastive" and args.stage == "pretrain":
        dataset = MultiModalSequenceDataset(args, index_file)
        batch_size = min(batch_size, len(dataset) * args.dataset_config["seq_len"])
        return DataLoader(dataset, batch_sampler=BatchSeqSampler(args, batch_size, dataset, shard=(option == "train")), num_workers=workers)
    dataset = MultiModalDataset(args, index_file, getattr(args, "label_ratio", 1) if option == "train" else 1)
    return DataLoader(dataset, batch_size=min(batch_size, len(dataset)), shuffle=(option == "train"), num_workers=workers)
