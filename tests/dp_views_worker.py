"""Worker of tests/test_dp_parity_gpu.py::test_two_ranks_draw_the_same_views: one rank of a world_size-2 job (both on cuda:0, gloo).
Each rank draws the random views of three consecutive steps the way train.py's captured step does (Augmenter.forward_random_pair) on ITS
half of a global batch, and one dropout mask from the per-rank seed word.  Rank 0 checks: the view plans are identical on both ranks at
every step and change from step to step; the dropout masks differ."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "focal_amd", "src"), os.path.dirname(__file__)):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from conftest import make_args
    from data_augmenter.Augmenter import Augmenter
    from focal_amd import ops, runtime
    from input_utils.yaml_utils import load_yaml
    cfg = load_yaml(os.path.join(ROOT, "focal_amd", "src", "data", "MOD.yaml"))
    aug = Augmenter(make_args(cfg, "SW_Transformer", dev, "bf16"))
    assert aug.device_draws_supported()
    g = torch.Generator().manual_seed(5 + rank)
    x = {"shake": {"audio": torch.randn(4, 1, 10, 1600, generator=g).to(dev), "seismic": torch.randn(4, 1, 10, 20, generator=g).to(dev)}}
    plans = []
    for _ in range(3):
        aug.forward_random_pair(x)
        st = next(iter(aug._dev_states.values()))
        plans.append(st["plans"].cpu().clone())
    mask = ops.dropout(torch.ones(4096, device=dev), runtime.rng_state(dev), 77, 0.5).cpu()
    mine = torch.cat([p.flatten() for p in plans])
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    masks = [torch.empty_like(mask) for _ in range(world)]
    dist.all_gather(masks, mask)
    ok = True
    if rank == 0:
        same = all(torch.equal(both[0], b) for b in both[1:])
        moves = not (torch.equal(plans[0], plans[1]) and torch.equal(plans[1], plans[2]))
        differ = not torch.equal(masks[0], masks[1])
        print(f"plans identical on all ranks: {same}; plans change from step to step: {moves}; dropout masks differ between ranks: {differ}")
        ok = same and moves and differ
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, 0)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) else 1)


if __name__ == "__main__":
    main()
