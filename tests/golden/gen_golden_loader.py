#!/usr/bin/env python3
"""Fixture for the sample-file grouping and the sequence-aware batch sampler, produced by the REFERENCE's
MultiModalSequenceDataset / BatchSeqSampler in the build container (needs /root/reference):

    python tests/golden/gen_golden_loader.py

Only file NAMES matter here (no sample is loaded): an index of irregular, unsorted `{sequence}_{k}.pt` names, the subsequence
partition the reference derives from it, and the batches its sampler yields after `random.seed(5)`."""
import json
import os
import random
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import OUT, REF, install_reference, ref_args  # noqa: E402


def main():
    install_reference()
    import yaml
    from input_utils.multi_modal_dataloader import BatchSeqSampler
    from input_utils.multi_modal_dataset import MultiModalSequenceDataset
    cfg = yaml.safe_load(open(os.path.join(REF, "data", "MOD.yaml")))
    args = ref_args("SW_Transformer", cfg)
    rng = random.Random(123)
    names = []
    for seq, n in (("run3_veh_a_shake", 11), ("run1_veh_b_shake", 4), ("run7_x_shake", 1), ("run2_veh_a_shake", 9), ("r_10_z", 6)):
        names += [f"/data/samples/{seq}_{k}.pt" for k in range(n)]
    rng.shuffle(names)
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as fh:
        fh.write("\n".join(names) + "\n")
    ds = MultiModalSequenceDataset(args, fh.name)
    batch = 16
    sampler = BatchSeqSampler(args, batch, ds)
    random.seed(5)
    batches = [list(map(int, b)) for b in sampler]
    fix = {"seq_len": cfg["seq_len"], "files": names, "subseqs": ds.subseqs,
           "subseq_to_sample_idx": {k: list(map(int, v)) for k, v in ds.subseq_to_sample_idx.items()},
           "seed": 5, "batch_size": batch, "batches": batches, "len": int(len(sampler))}
    with open(os.path.join(OUT, "loader_partition.json"), "w") as f:
        json.dump(fix, f, indent=0)
    print("wrote loader_partition.json:", len(ds.subseqs), "subsequences,", len(batches), "batches")


if __name__ == "__main__":
    main()
