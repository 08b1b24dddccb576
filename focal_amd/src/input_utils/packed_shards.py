"""Packed sample shards + a prefetching loader (no counterpart in the reference).

The reference reads one `.pt` file per 130 KB window through DataLoader workers; at the 3e4 - 6e4 windows/s this build trains at
that is 4 - 8 GB/s of tiny-file I/O and unpickling.  `pack_index()` converts an index file ONCE into one flat float32 file
per (location, modality) plus labels and the sample names; `PackedSequenceLoader` memory-maps them, assembles every batch of
whole subsequences straight into pinned host buffers in a background thread (two batches ahead) and hands out device tensors
copied on a side stream.  Subsequence partition, shuffling (`random.shuffle`, one order per epoch) and batch composition are
the reference's (BatchSeqSampler, input_utils/multi_modal_dataloader.py:51-78); under torch.distributed every rank draws the
SAME order (shared seed) and takes its contiguous share of each global batch, so the global batch is the single-device one."""
import json
import os
import queue
import random
import threading

import numpy as np
import torch

from input_utils.multi_modal_dataset import partition_subsequences, read_index, sample_label

MANIFEST = "manifest.json"


def pack_index(args, index_file, out_dir):
    """`.pt` samples listed in `index_file` -> out_dir/{manifest.json, labels.npy, <loc>.<mod>.f32}."""
    files = read_index(index_file)
    os.makedirs(out_dir, exist_ok=True)
    first = torch.load(files[0])
    shapes = {loc: {mod: list(t.shape) for mod, t in mods.items()} for loc, mods in first["data"].items()}
    maps = {(loc, mod): np.lib.format.open_memmap(os.path.join(out_dir, f"{loc}.{mod}.npy"), mode="w+", dtype=np.float32,
                                                  shape=(len(files), *shp))
            for loc, mods in shapes.items() for mod, shp in mods.items()}
    labels = []
    for i, f in enumerate(files):
        s = first if i == 0 else torch.load(f)
        for (loc, mod), mm in maps.items():
            mm[i] = s["data"][loc][mod].float().numpy()
        lab = sample_label(args, s)
        labels.append(np.asarray(lab.numpy() if torch.is_tensor(lab) else lab))
    for mm in maps.values():
        mm.flush()
    np.save(os.path.join(out_dir, "labels.npy"), np.stack(labels))
    with open(os.path.join(out_dir, MANIFEST), "w") as fh:
        json.dump({"files": [os.path.basename(f) for f in files], "shapes": shapes}, fh)
    return out_dir


class PackedSequenceLoader:
    def __init__(self, args, pack_dir, batch_size, shuffle=True, device=None, prefetch=2, seed=0, resident=None, shard=True):
        """shard=False: the whole batch on every rank that iterates (validation / test sets, which rank 0 evaluates alone)."""
        with open(os.path.join(pack_dir, MANIFEST)) as fh:
            man = json.load(fh)
        self.args, self.device = args, device if device is not None else args.device
        seq_len = args.dataset_config["seq_len"]
        delimiter = "-" if args.dataset == "RealWorld_HAR" else "_"
        self.subseqs, self.subseq_to_sample_idx = partition_subsequences(man["files"], seq_len, delimiter)
        paths = {(loc, mod): os.path.join(pack_dir, f"{loc}.{mod}.npy") for loc, mods in man["shapes"].items() for mod in mods}
        if resident is None:  # shards that fit comfortably are read into RAM once (a cold memory map costs a page fault per 4 KB
            total = sum(os.path.getsize(p) for p in paths.values())       # on its first pass: ~1e3 windows/s instead of >1e4)
            try:
                phys = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
            except (ValueError, OSError):
                phys = 0
            resident = total < 0.25 * phys
        self.maps = {k: (np.load(p) if resident else np.load(p, mmap_mode="r")) for k, p in paths.items()}
        self.labels = np.load(os.path.join(pack_dir, "labels.npy"))
        batch_size = min(batch_size, len(self.subseqs) * seq_len)
        self.subseq_batch = batch_size // seq_len
        self.shuffle, self.prefetch, self.seed, self.epoch = shuffle, prefetch, seed, 0
        import torch.distributed as dist
        self.world = dist.get_world_size() if (shard and dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        if self.subseq_batch % self.world != 0:
            raise ValueError(f"{self.subseq_batch} subsequences per global batch do not split over {self.world} ranks")

    def __len__(self):
        if self.world == 1:
            return int(np.ceil(len(self.subseqs) / self.subseq_batch))
        full, rem = divmod(len(self.subseqs), self.subseq_batch)
        return full + (1 if (rem // self.world) * self.world >= 2 else 0)

    def set_epoch(self, epoch):
        """The training epoch the next pass belongs to (rank-consistent shuffling under torch.distributed is seeded by it): passes that
        are not training epochs -- the KNN fit over the training set -- and a resumed run then do not shift the sequence of orders."""
        self.epoch, self._epoch_set = int(epoch), True

    def batches(self):
        """Sample-index lists of this rank's share of every global batch, in the reference's order."""
        order = list(range(len(self.subseqs)))
        if self.shuffle:
            random.Random(self.seed + self.epoch).shuffle(order) if self.world > 1 else random.shuffle(order)
        from input_utils.multi_modal_dataloader import rank_share
        for lo in range(0, len(order), self.subseq_batch):
            # equal shares on every rank, also for the short last batch of an epoch (its remainder is dropped): the ranks must run
            # the same number of steps with the same local batch, or the all-gather / all-reduce of that step cannot match
            mine = rank_share(order[lo:lo + self.subseq_batch], self.world, self.rank)
            idx = []
            for s in mine:
                idx.extend(self.subseq_to_sample_idx[self.subseqs[s]])
            if idx:
                yield idx

    def _ring(self, rows):
        """Fixed host staging buffers (pinned when a GPU is present), allocated once: a fresh 16 MB allocation per batch costs
        thousands of page faults (or a hipHostMalloc) every time."""
        if getattr(self, "_slots", None) is None or self._slot_rows < rows:
            n = self.prefetch + 2
            pin = torch.cuda.is_available()
            self._slots = []
            for _ in range(n):
                bufs = {k: torch.empty((rows,) + tuple(mm.shape[1:]), dtype=torch.float32, pin_memory=pin) for k, mm in self.maps.items()}
                self._slots.append({"bufs": bufs, "event": None})
            self._slot_rows = rows
        return self._slots

    def _assemble(self, idx, slot):
        ii = np.asarray(idx)
        if slot["event"] is not None:
            slot["event"].synchronize()  # the previous H2D copy out of this slot has finished
        host = {}
        for (loc, mod), mm in self.maps.items():
            dst = slot["bufs"][(loc, mod)][:len(ii)]
            np.take(mm, ii, axis=0, out=dst.numpy())
            host.setdefault(loc, {})[mod] = dst
        return host, torch.from_numpy(self.labels[ii])

    def __iter__(self):
        q = queue.Queue(maxsize=self.prefetch)
        batches = list(self.batches())
        slots = self._ring(max((len(b) for b in batches), default=1))

        def producer():
            for k, idx in enumerate(batches):
                slot = slots[k % len(slots)]
                q.put((self._assemble(idx, slot), slot))
            q.put(None)
        threading.Thread(target=producer, daemon=True).start()
        if not getattr(self, "_epoch_set", False):
            self.epoch += 1
        use_gpu = torch.cuda.is_available() and torch.device(self.device).type == "cuda"
        copy_stream = torch.cuda.Stream(device=self.device) if use_gpu else None
        while True:
            item = q.get()
            if item is None:
                return
            (host, lab), slot = item
            if not use_gpu:
                yield {loc: {mod: t.clone() for mod, t in mods.items()} for loc, mods in host.items()}, lab
                continue
            with torch.cuda.stream(copy_stream):
                dev = {loc: {mod: t.to(self.device, non_blocking=True) for mod, t in mods.items()} for loc, mods in host.items()}
                slot["event"] = torch.cuda.Event()
                slot["event"].record(copy_stream)
            torch.cuda.current_stream(self.device).wait_stream(copy_stream)
            for mods in dev.values():
                for t in mods.values():
                    t.record_stream(torch.cuda.current_stream(self.device))
            yield dev, lab
