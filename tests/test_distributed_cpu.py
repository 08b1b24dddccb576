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
    # the one collective inside the row-sharded loss head: every rank's chunk, in rank order, on persistent buffers
    from focal_amd import distributed as fd

    class Head:
        send = torch.full((40,), float(rank + 1))
        chunks = torch.zeros(world, 40)
    fd.exchange_loss_chunks(Head)
    assert all(torch.all(Head.chunks[r] == r + 1) for r in range(world))
    # sharded by default from 6 ranks up; the switch forces either form; never inside local_only()
    assert fd.shard_loss_head() is False
    os.environ["FOCAL_LOSS_SHARD"] = "1"
    assert fd.shard_loss_head() is True
    with fd.local_only():
        assert fd.shard_loss_head() is False
    os.environ["FOCAL_LOSS_SHARD"] = "0"
    assert fd.shard_loss_head() is False
    del os.environ["FOCAL_LOSS_SHARD"]
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


def _share_worker(rank, world, port, out, tmp):
    """Both loaders of the data path under 2 gloo ranks on a subsequence count that does NOT divide the global batch: every
    rank must see the same number of batches with the same local size, the shares must be disjoint and, together, be the
    leading part of the single-process batch."""
    import sys
    import types
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    for p in (root, os.path.join(root, "focal_amd", "src")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from input_utils.multi_modal_dataloader import BatchSeqSampler
    from input_utils.multi_modal_dataset import MultiModalSequenceDataset
    from input_utils.packed_shards import PackedSequenceLoader
    cfg = {"seq_len": 4, "location_names": ["shake"], "modality_names": ["audio", "seismic"]}
    args = types.SimpleNamespace(dataset="MOD", dataset_config=cfg, task="vehicle_classification", device=torch.device("cpu"),
                                 train_mode="contrastive", stage="pretrain", sequence_sampler=True)
    ds = MultiModalSequenceDataset(args, os.path.join(tmp, "index.txt"))
    sampler = BatchSeqSampler(args, 16, ds, seed=5)           # global batch: 4 subsequences; 11 subsequences in the index
    a = [list(b) for b in sampler]
    assert len(a) == len(sampler)
    packed = PackedSequenceLoader(args, os.path.join(tmp, "pack"), 16, shuffle=True, device=torch.device("cpu"), seed=5)
    b = list(packed.batches())
    assert len(b) == len(packed)
    whole = PackedSequenceLoader(args, os.path.join(tmp, "pack"), 16, shuffle=False, device=torch.device("cpu"), shard=False)
    out[rank] = {"sampler": a, "packed": b, "n_sub": len(ds.subseqs), "whole": [len(x) for x in whole.batches()]}
    dist.destroy_process_group()


def test_two_rank_loaders_take_equal_shares_of_a_ragged_epoch(tmp_path):
    import sys
    import types
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    for p in (root, os.path.join(root, "focal_amd", "src")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from input_utils.packed_shards import pack_index
    files = []
    g = torch.Generator().manual_seed(3)
    for seq, n in (("runA_shake", 17), ("runB_shake", 14), ("runC_shake", 9)):   # 5 + 4 + 3 = 12 ... minus: 17->5, 14->4, 9->3
        for k in range(n):
            f = os.path.join(str(tmp_path), f"{seq}_{k}.pt")
            torch.save({"label": torch.tensor(len(files) % 3), "flag": {"shake": {"audio": True, "seismic": True}},
                        "data": {"shake": {"audio": torch.randn(1, 10, 16, generator=g), "seismic": torch.randn(1, 10, 4, generator=g)}}}, f)
            files.append(f)
    files = files[:-5]  # 17 + 14 + 4 windows -> 5 + 4 + 1 = 10 subsequences... keep it ragged against a global batch of 4
    (tmp_path / "index.txt").write_text("\n".join(files) + "\n")
    cfg = {"seq_len": 4, "location_names": ["shake"], "modality_names": ["audio", "seismic"]}
    args = types.SimpleNamespace(dataset="MOD", dataset_config=cfg, task="vehicle_classification", device=torch.device("cpu"),
                                 train_mode="contrastive", stage="pretrain", sequence_sampler=True)
    pack_index(args, str(tmp_path / "index.txt"), str(tmp_path / "pack"))
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_share_worker, args=(world, port, out, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    n_sub = r0["n_sub"]
    assert n_sub % 4 != 0, "the fixture must leave a short last batch"
    for kind in ("sampler", "packed"):
        a, b = r0[kind], r1[kind]
        assert len(a) == len(b) and len(a) >= 2, kind                               # same number of steps on both ranks
        assert [len(x) for x in a] == [len(x) for x in b], kind                     # same local batch at every step
        assert all(len(x) % 4 == 0 and len(x) >= 4 for x in a), kind                # whole subsequences
        full = n_sub // 4
        assert all(len(x) == 8 for x in a[:full]), kind                             # 2 of the 4 subsequences each
        rem = n_sub % 4
        assert len(a) == full + (1 if (rem // 2) * 2 >= 2 else 0), kind
        if len(a) > full:
            assert len(a[-1]) == (rem // 2) * 4, kind                               # the remainder's odd subsequence is dropped
    # un-sharded loaders (validation / test) hand the whole batch to whichever rank iterates them
    assert r0["whole"] == r1["whole"] and sum(r0["whole"]) == n_sub * 4


def _capture_agreement_worker(rank, world, port, out):
    """A rank whose capture fails OUTSIDE a segment capture (after its peers have finished theirs) must take every rank back to the
    eager step: the final agreement of CapturedTrainStep.__call__.  The capture itself is stubbed -- it needs a GPU."""
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from focal_amd import graph_step
    torch.cuda.synchronize = lambda *a, **k: None

    class Seg:
        device, loss = torch.device("cpu"), torch.zeros(())

        def run(self):
            pass

    res = {}
    for failing in (1, None):
        step = graph_step.CapturedTrainStep(None, None, None, warm_steps=1)
        step._segments = lambda v1, v2: Seg()

        def capture(make_segments, step=step, failing=failing):
            if rank == failing:
                raise RuntimeError("out of memory between two segments")
            step.replay, step.loss = (lambda: None), Seg.loss
        step._capture = capture
        v = {"shake": {"audio": torch.zeros(2)}}
        step(v, v)
        res[failing] = (step.replay is None, step.enabled)
    out[rank] = res
    dist.destroy_process_group()


def test_capture_failure_on_one_rank_keeps_every_rank_eager():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_capture_agreement_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        assert out[r][1] == (True, False), dict(out)     # rank 1 failed: nobody replays
        assert out[r][None] == (False, True), dict(out)  # nobody failed: everybody replays


def _between_segments_worker(rank, world, port, out):
    """StepSegments.capture with the graph machinery stubbed (no GPU here): rank 1 fails BETWEEN segment A and segment B (its eager
    exchange raises).  Every rank must leave capture() with CaptureAborted and the NEXT collective -- the caller's final agreement --
    must pair up on both ranks (the failing rank used to skip segment B's agreement slot)."""
    import contextlib
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FOCAL_LOSS_SHARD="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from focal_amd import graph_step
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.graph = lambda *a, **k: contextlib.nullcontext()

    class G:
        def pool(self):
            return None

        def replay(self):
            pass
    torch.cuda.CUDAGraph = G

    class Opt:
        def sync_lr(self):
            pass

        def reduce_buckets(self, model):
            return None

        def wait_reductions(self):
            pass

        def reduce_gradients(self):
            pass

    class Seg(graph_step.StepSegments):
        def seg_a(self):
            pass

        def seg_b(self):
            pass

        def seg_c(self):
            pass

        def exchange(self):
            if rank == 1:
                raise RuntimeError("out of memory between two segments")

    seg = Seg(None, None, Opt(), lambda: None, torch.device("cpu"))
    try:
        seg.capture(None)
        got = "captured"
    except graph_step.CaptureAborted as e:
        got = "aborted: " + str(e)
    final = graph_step.agree(got == "captured", torch.device("cpu"))   # what CapturedTrainStep._step does next
    out[rank] = (got, final)
    dist.destroy_process_group()


def test_failure_between_segment_captures_keeps_the_agreements_paired():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_between_segments_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0][0].startswith("aborted") and "between segment captures" in out[1][0], dict(out)
    assert out[0][1] is False and out[1][1] is False


def _bf16_reduce_worker(rank, world, port, out):
    """The bf16-wire gradient reduction (FOCAL_GRAD_REDUCE=bf16) against the fp32 all-reduce on the same gradients: the summed gradient
    within 1e-2 of its norm (observed ~2e-3: one bf16 rounding per contribution and one of the sum), a rank's own shard better than the
    others' (its contribution is never rounded), sizes that do not divide by the world size, repeated calls on the persistent buffers."""
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from focal_amd import distributed as fd

    class Arena:
        pass
    res = {}
    for n in (1001, 4096):
        for rep in range(2):
            g = torch.Generator().manual_seed(100 * n + 10 * rep + rank)
            mine = torch.randn(n, generator=g) * torch.logspace(-3, 1, n)   # gradients over four decades
            ref = mine.clone()
            dist.all_reduce(ref)
            ar = Arena()
            ar.grad = mine.clone()
            os.environ["FOCAL_GRAD_REDUCE"] = "bf16"
            fd.all_reduce_gradients(ar)
            del os.environ["FOCAL_GRAD_REDUCE"]
            mag = mine.abs()
            dist.all_reduce(mag)   # sum over ranks of |contribution|: what an element's rounding errors scale with (the sum itself may cancel)
            res[(n, rep)] = ((ar.grad - ref).norm() / ref.norm()).item(), ((ar.grad - ref).abs() / mag.clamp_min(1e-30)).max().item()
            ar2 = Arena()
            ar2.grad = mine.clone()
            fd.all_reduce_gradients(ar2)                                     # the default stays the exact fp32 sum
            assert torch.equal(ar2.grad, ref)
    assert fd.grad_reduce_dtype() == "fp32"
    out[rank] = res
    dist.destroy_process_group()


def test_bf16_wire_gradient_reduction_matches_fp32_all_reduce():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bf16_reduce_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        for key, (rel_l2, rel_max) in out[r].items():
            assert rel_l2 < 1e-2, (r, key, rel_l2)        # of the update's norm
            assert rel_max < 1.2e-2, (r, key, rel_max)    # element-wise, of the summed magnitudes: bf16 roundings of 2^-9 each


def _view_seed_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from focal_amd import runtime
    cpu = torch.device("cpu")
    v = runtime.view_state(cpu)     # collective on first use: rank 0's word, broadcast
    d = runtime.rng_state(cpu)      # per rank, from os.urandom
    again = runtime.view_state(cpu)
    out[rank] = (v.tolist(), d.tolist(), again.data_ptr() == v.data_ptr())
    dist.destroy_process_group()


def test_two_ranks_share_the_view_seed_and_keep_their_own_dropout_seed():
    """the random-view draws of a data-parallel job are keyed by ONE word (rank 0's, broadcast) so that the global
    batch gets one augmenter / coin / permutation / scale / phase per view as the reference's batch does (Augmenter.py:76-113);
    the dropout streams keep a word per rank.  (The plans drawn from equal states are equal: tests/test_kernels_gpu.py::
    test_shared_view_draws_advance_their_own_state; the two-rank plans on a GPU: tests/test_dp_parity_gpu.py.)"""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_view_seed_worker, args=(world, port, out), nprocs=world, join=True)
    (v0, d0, same0), (v1, d1, same1) = out[0], out[1]
    assert v0 == v1 and len(v0) == 4 and v0[1:] == [0, 0, 0] and same0 and same1
    assert d0[0] != d1[0]   # (two os.urandom words: equal with probability 2^-31)
