#!/usr/bin/env python3
"""Convert an index file of per-sample `.pt` windows into a packed shard directory (input_utils/packed_shards.py).

    python tools/pack_shards.py -dataset=MOD -task=vehicle_classification <index.txt> <out_dir>

Point `pretrain_index_file` (or a task's train/val/test index) in the dataset YAML at <out_dir> to train from it."""
import argparse
import os
import sys
import types

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "focal_amd", "src"))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("-dataset", default="MOD")
    p.add_argument("-task", default="vehicle_classification")
    p.add_argument("index_file")
    p.add_argument("out_dir")
    a = p.parse_args()
    from input_utils.packed_shards import pack_index
    from input_utils.yaml_utils import load_yaml
    cfg = load_yaml(os.path.join(ROOT, "focal_amd", "src", "data", f"{a.dataset}.yaml"))
    args = types.SimpleNamespace(dataset=a.dataset, task=a.task, dataset_config=cfg)
    print("packed into", pack_index(args, a.index_file, a.out_dir))


if __name__ == "__main__":
    main()
