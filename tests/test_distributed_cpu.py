"""Data-parallel exchange on 2 CPU ranks over gloo: all-gathered embeddings give every rank the global
batch, each rank back-propagates only its slice, and the SUM of the ranks' parameter gradients equals the
single-process gradient of the global-batch loss.  (The HIP loss head needs a GPU; the oracle loss stands in for it
here -- what is under test is the exchange, not the loss.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    sys.path.insert(0, root)
    from focal_amd.distributed import all_reduce_gradients, gather_features
    from oracle.config import load_config
    from oracle.loss import focal_loss_terms
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    cfg = load_config()
    mods, Bl, D = cfg["modality_names"], 8, 256
    W = {m: torch.randn(16, D) * 0.3 for m in mods}  # a shared "encoder": z = x @ W
    xs = {v: {m: torch.randn(world * Bl, 16, generator=torch.Generator().manual_seed(10 * v + i)) for i, m in enumerate(mods)} for v in (1, 2)}

    def loss_of(Wp, rows):
        f = [{m: xs[v][m][rows] @ Wp[m] for m in mods} for v in (1, 2)]
        return f

    # single-process reference on the global batch
    Wr = {m: W[m].clone().requires_grad_(True) for m in mods}
    f1, f2 = loss_of(Wr, slice(None))
    focal_loss_terms(f1, f2, cfg, "SW_Transformer")["total"].backward()
    # data parallel: local shard -> gather -> same loss on every rank -> local backward -> all-reduce(SUM)
    Wd = {m: W[m].clone().requires_grad_(True) for m in mods}
    l1, l2 = loss_of(Wd, slice(rank * Bl, (rank + 1) * Bl))
    g1, g2 = gather_features([l1, l2])
    assert g1[mods[0]].shape == (world * Bl, D)
    total = focal_loss_terms(g1, g2, cfg, "SW_Transformer")["total"]
    total.backward()

    class Arena:
        pass
    ar = Arena()
    ar.grad = torch.cat([Wd[m].grad.reshape(-1) for m in mods])
    all_reduce_gradients(ar, bucket_bytes=4096)
    ref = torch.cat([Wr[m].grad.reshape(-1) for m in mods])
    err = ((ar.grad - ref).norm() / ref.norm()).item()
    out[rank] = err
    dist.destroy_process_group()


def test_two_rank_gather_and_gradient_sum():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == world
    for r in range(world):
        assert out[r] < 1e-5, dict(out)


def test_gather_is_identity_without_process_group():
    from focal_amd.distributed import gather_features, is_dist
    assert not is_dist()
    d = [{"a": torch.ones(2, 3)}, {"a": torch.zeros(2, 3)}]
    assert gather_features(d) is d
