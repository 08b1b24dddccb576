"""Sample files and their grouping into fixed-length subsequences (reference: input_utils/multi_modal_dataset.py:9-130).

A sample is a `.pt` dict `{"label": tensor | {task: tensor}, "flag": {loc: {mod: bool}}, "data": {loc: {mod: float [c, i, s]}}}`
(data_preprocess/MOD/extract_samples.py:191-226); an index file lists one path per line.  File names are `{sequence}_{k}.pt`
(`-` instead of `_` for RealWorld_HAR): consecutive k of one sequence are consecutive windows in time, and FOCAL's temporal
terms need batches made of whole subsequences of `seq_len` windows (the last one padded by repeating its final sample)."""
import os
from random import shuffle

import numpy as np
import torch
from torch.utils.data import Dataset


def read_index(index_file):
    return [str(s) for s in np.atleast_1d(np.loadtxt(index_file, dtype=str))]


def sample_label(args, sample):
    """reference :43-55 / :116-128: task-keyed label dicts (ACIDS / Parkland) or a plain tensor."""
    lab = sample["label"]
    if isinstance(lab, dict):
        key = {"vehicle_classification": "vehicle_type", "distance_classification": "distance", "speed_classification": "speed"}.get(args.task)
        if key is None:
            raise ValueError(f"Unknown task: {args.task}")
        return lab[key]
    return lab


def partition_subsequences(sample_files, seq_len, delimiter="_"):
    """(subseqs, subseq_to_sample_idx) exactly as MultiModalSequenceDataset.partition_subsequences builds them (reference :68-108):
    sequences in first-appearance order, samples sorted by their integer suffix, chunks of seq_len, last chunk padded."""
    seq_to_samples = {}
    for idx, f in enumerate(sample_files):
        seq = os.path.basename(f).rsplit(delimiter, 1)[0]
        seq_to_samples.setdefault(seq, []).append((idx, f))
    subseqs, subseq_to_sample_idx = [], {}
    for seq, items in seq_to_samples.items():
        items.sort(key=lambda x: int(os.path.basename(x[1]).rsplit(delimiter, 1)[1].split(".")[0]))
        ids = [e[0] for e in items]
        for i in range(0, len(ids), seq_len):
            name = f"{seq}_{i}"
            chunk = ids[i:i + seq_len]
            while len(chunk) < seq_len:
                chunk.append(chunk[-1])
            subseqs.append(name)
            subseq_to_sample_idx[name] = chunk
    return subseqs, subseq_to_sample_idx


class MultiModalDataset(Dataset):
    """Flat dataset (validation / test / supervised splits), reference :9-57."""

    def __init__(self, args, index_file, label_ratio=1):
        self.args = args
        self.sample_files = read_index(index_file)
        if label_ratio < 1:
            shuffle(self.sample_files)
            self.sample_files = self.sample_files[: round(len(self.sample_files) * label_ratio)]

    def __len__(self):
        return len(self.sample_files)

    def __getitem__(self, idx):
        sample = torch.load(self.sample_files[idx])
        return sample["data"], sample_label(self.args, sample)


class MultiModalSequenceDataset(Dataset):
    """Dataset addressed by SAMPLE index whose length is the number of subsequences (reference :58-130; BatchSeqSampler hands it
    the sample indices of whole subsequences)."""

    def __init__(self, args, index_file):
        self.args = args
        self.sample_files = read_index(index_file)
        delimiter = "-" if args.dataset == "RealWorld_HAR" else "_"
        self.subseqs, self.subseq_to_sample_idx = partition_subsequences(self.sample_files, args.dataset_config["seq_len"], delimiter)

    def __len__(self):
        return len(self.subseqs)

    def __getitem__(self, sample_idx):
        sample = torch.load(self.sample_files[sample_idx])
        return sample["data"], sample_label(self.args, sample)
