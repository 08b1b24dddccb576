#!/usr/bin/env python3
"""windows/s of the two real-data paths on MOD-shaped samples (130 KB / window): reference-style per-sample `.pt` files through
a torch DataLoader vs the packed-shard prefetching loader.  Host-side only (runs anywhere): python tools/bench_loader.py [N]"""
import os
import random
import sys
import tempfile
import time
import types

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "focal_amd", "src"))
sys.path.insert(0, ROOT)
from torch.utils.data import DataLoader  # noqa: E402

from input_utils.multi_modal_dataloader import BatchSeqSampler  # noqa: E402
from input_utils.multi_modal_dataset import MultiModalSequenceDataset  # noqa: E402
from input_utils.packed_shards import PackedSequenceLoader, pack_index  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cfg = {"seq_len": 4}
args = types.SimpleNamespace(dataset="MOD", dataset_config=cfg, task="vehicle_classification", device=torch.device("cpu"))
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
    files = []
    for s in range(N // 64):
        for k in range(64):
            f = os.path.join(d, f"seq{s}_shake_{k}.pt")
            torch.save({"label": torch.tensor(k % 7), "flag": {"shake": {"audio": True, "seismic": True}},
                        "data": {"shake": {"audio": torch.randn(1, 10, 1600), "seismic": torch.randn(1, 10, 20)}}}, f)
            files.append(f)
    idx = os.path.join(d, "index.txt")
    open(idx, "w").write("\n".join(files) + "\n")
    ds = MultiModalSequenceDataset(args, idx)
    for workers in (0, 6):
        dl = DataLoader(ds, batch_sampler=BatchSeqSampler(args, 256, ds), num_workers=workers)
        t0 = time.time()
        n = sum(lab.shape[0] for _, lab in dl)
        print(f"per-sample .pt files, DataLoader workers={workers}: {n / (time.time() - t0):9.0f} windows/s")
    t0 = time.time()
    pack = pack_index(args, idx, os.path.join(d, "pack"))
    print(f"pack_index: {N / (time.time() - t0):9.0f} windows/s (one-off)")
    loader = PackedSequenceLoader(args, pack, 256, device=torch.device("cpu"))
    for _ in range(2):
        t0 = time.time()
        n = sum(lab.shape[0] for _, lab in loader)
        print(f"packed shards, prefetch thread:          {n / (time.time() - t0):9.0f} windows/s")
