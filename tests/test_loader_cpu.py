"""Host logic of the data path: sample-file grouping and the sequence-aware batch sampler against the
reference's own classes (tests/golden/loader_partition.json, tests/golden/gen_golden_loader.py), and the packed-shard loader
against the per-sample `.pt` path on a small synthetic index."""
import json
import os
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "focal_amd", "src")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _args(seq_len=4, dataset="MOD"):
    cfg = {"seq_len": seq_len, "location_names": ["shake"], "modality_names": ["audio", "seismic"]}
    return types.SimpleNamespace(dataset=dataset, dataset_config=cfg, task="vehicle_classification", device=torch.device("cpu"),
                                 train_mode="contrastive", stage="pretrain", sequence_sampler=True)


def test_partition_and_sampler_match_reference_fixture(tmp_path):
    from input_utils.multi_modal_dataloader import BatchSeqSampler
    from input_utils.multi_modal_dataset import MultiModalSequenceDataset
    fx = json.load(open(os.path.join(GOLD, "loader_partition.json")))
    idx = tmp_path / "index.txt"
    idx.write_text("\n".join(fx["files"]) + "\n")
    ds = MultiModalSequenceDataset(_args(fx["seq_len"]), str(idx))
    assert ds.subseqs == fx["subseqs"]
    assert ds.subseq_to_sample_idx == fx["subseq_to_sample_idx"]
    sampler = BatchSeqSampler(_args(fx["seq_len"]), fx["batch_size"], ds)
    assert len(sampler) == fx["len"]
    random.seed(fx["seed"])
    assert [list(b) for b in sampler] == fx["batches"]


def _write_samples(root, seqs):
    files = []
    g = torch.Generator().manual_seed(1)
    for seq, n in seqs:
        for k in range(n):
            f = os.path.join(root, f"{seq}_{k}.pt")
            torch.save({"label": torch.tensor(len(files) % 3), "flag": {"shake": {"audio": True, "seismic": True}},
                        "data": {"shake": {"audio": torch.randn(1, 10, 16, generator=g), "seismic": torch.randn(1, 10, 4, generator=g)}}}, f)
            files.append(f)
    return files


def test_packed_loader_equals_per_sample_files(tmp_path):
    """pack_index + PackedSequenceLoader yields the same tensors, labels and batch composition as the reference-style DataLoader
    over the `.pt` files (same `random` seed -> same subsequence order)."""
    from torch.utils.data import DataLoader
    from input_utils.multi_modal_dataloader import BatchSeqSampler
    from input_utils.multi_modal_dataset import MultiModalSequenceDataset
    from input_utils.packed_shards import PackedSequenceLoader, pack_index
    files = _write_samples(str(tmp_path), [("runA_shake", 9), ("runB_shake", 5), ("runC_shake", 4)])
    random.Random(3).shuffle(files)
    idx = tmp_path / "index.txt"
    idx.write_text("\n".join(files) + "\n")
    args = _args()
    pack = pack_index(args, str(idx), str(tmp_path / "pack"))
    ds = MultiModalSequenceDataset(args, str(idx))
    ref_loader = DataLoader(ds, batch_sampler=BatchSeqSampler(args, 8, ds), num_workers=0)
    random.seed(11)
    ref = [(d, l) for d, l in ref_loader]
    fast = PackedSequenceLoader(args, pack, 8, shuffle=True, device=torch.device("cpu"))
    random.seed(11)
    got = list(fast)
    assert len(got) == len(ref) == len(fast)
    for (dr, lr_), (dg, lg) in zip(ref, got):
        assert torch.equal(lr_, lg)
        for m in ("audio", "seismic"):
            assert torch.equal(dr["shake"][m], dg["shake"][m])
    # batches hold whole subsequences: 4 consecutive windows of one sequence, the last one padded by repetition
    names = [os.path.basename(f) for f in files]
    for idxs in fast.batches():
        for j in range(0, len(idxs), 4):
            seqs = {names[i].rsplit("_", 1)[0] for i in idxs[j:j + 4]}
            assert len(seqs) == 1


def test_device_list_spawns_ranks_only_for_focal_pretraining():
    """`-gpu=0,1` starts a data-parallel job only where the loop is data-parallel (FOCAL pretraining); the supervised and finetune
    stages take the first device, as the reference does with a device list (params/params_util.py:43) -- ranks of such a job would
    each train their own model and write the same checkpoints."""
    import argparse
    import importlib.util
    spec = importlib.util.spec_from_file_location("focal_train_entry", os.path.join(ROOT, "focal_amd", "src", "train.py"))
    src = open(spec.origin).read()
    ns = {}
    start = src.index("def wants_data_parallel")
    exec(src[start:src.index("def main_train")], ns)
    w = ns["wants_data_parallel"]
    assert w(argparse.Namespace(learn_framework="FOCAL", stage="pretrain"))
    assert not w(argparse.Namespace(learn_framework="FOCAL", stage="finetune"))
    assert not w(argparse.Namespace(learn_framework="no", stage="pretrain"))
