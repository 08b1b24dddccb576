"""Data-parallel FOCAL pretraining: one process per GPU, RCCL over xGMI through torch.distributed.

The path shards by whole subsequences.  There is exactly one exchange step on the data path:
  * all-gather of the projected embeddings (2 views x M modalities, packed into ONE buffer = one RCCL call,
    ~1 MB per rank: latency-bound on xGMI, so never more than one call), after which every rank evaluates the loss
    head on the global batch (global negatives / global ranking) and back-propagates only its own slice;
  * gradients: because every rank differentiates the SAME global loss w.r.t. its own samples, the exact gradient of
    the global-batch loss is the SUM over ranks -> one all-reduce(SUM) over the flat gradient arena (no averaging).
"""
import os

import torch
import torch.distributed as dist


import contextlib

_LOCAL_ONLY = [0]


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and not _LOCAL_ONLY[0]


def world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


@contextlib.contextmanager
def local_only():
    """Inside: the data path issues no collective (rank 0's validation / checkpoint branch runs alone while the other
    ranks wait at the barrier that follows it)."""
    _LOCAL_ONLY[0] += 1
    try:
        yield
    finally:
        _LOCAL_ONLY[0] -= 1


@torch.no_grad()
def broadcast_module(module, src=0):
    """Identical initial weights / buffers on every rank (the reference seeds nothing: each process would draw its own)."""
    if not is_dist():
        return
    for t in list(module.state_dict().values()):
        if t.is_floating_point() or t.dtype in (torch.int64, torch.int32):
            dist.broadcast(t, src)


@torch.no_grad()
def all_gather_rows(t):
    """[n, ...] per rank (equal n) -> [W * n, ...], rank-major."""
    if not is_dist():
        return t
    t = t.contiguous()
    out = torch.empty((dist.get_world_size() * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out


_STATIC = {}


def _static(tag, shape, dtype, device):
    """Persistent exchange buffers: stable addresses, so that a captured hipGraph can fill / read them while the collective
    itself stays an eager RCCL call between graph segments (bench.py)."""
    key = (tag, tuple(shape), dtype, torch.device(device))
    if key not in _STATIC:
        _STATIC[key] = torch.empty(shape, dtype=dtype, device=device)
    return _STATIC[key]


class _PackFeatures(torch.autograd.Function):
    """[B_local, D] x n  ->  one [n, B_local, D] send buffer (ONE collective for all views and modalities)."""

    @staticmethod
    def forward(ctx, *feats):
        buf = _static("send", (len(feats),) + tuple(feats[0].shape), feats[0].dtype, feats[0].device)
        torch.stack([f.contiguous() for f in feats], 0, out=buf)
        return buf

    @staticmethod
    def backward(ctx, g):
        return tuple(g.unbind(0))


class _GatherPacked(torch.autograd.Function):
    @staticmethod
    def forward(ctx, packed):
        W = dist.get_world_size()
        flat = _static("recv", (W * packed.shape[0],) + tuple(packed.shape[1:]), packed.dtype, packed.device)
        dist.all_gather_into_tensor(flat, packed)  # concatenation along dim 0: [rank0 feats.., rank1 feats.., ...]
        ctx.rank = dist.get_rank()
        return flat.view((W,) + tuple(packed.shape))

    @staticmethod
    def backward(ctx, g):
        # every rank evaluates the same global loss: the gradient of ITS samples is its own slice (the sum over ranks is
        # taken by the gradient all-reduce)
        return g[ctx.rank]


def pack_features(feature_dicts):
    keys = [(i, m) for i, d in enumerate(feature_dicts) for m in d]
    return _PackFeatures.apply(*[feature_dicts[i][m] for i, m in keys]), keys


def exchange_packed(packed):
    """The one exchange step of the data path: [n, B_local, D] -> [W, n, B_local, D]."""
    return _GatherPacked.apply(packed)


@torch.no_grad()
def replay_exchange(packed):
    """Same collective on the same persistent buffers without autograd (between replays of captured graph segments)."""
    flat = _static("recv", (dist.get_world_size() * packed.shape[0],) + tuple(packed.shape[1:]), packed.dtype, packed.device)
    dist.all_gather_into_tensor(flat, packed)


def unpack_gathered(gathered, keys, n_dicts):
    # rank-major concat keeps subsequences contiguous (models/loss.py:152-155 reshapes [B] -> [b, seq])
    out = [dict() for _ in range(n_dicts)]
    for j, (i, m) in enumerate(keys):
        out[i][m] = gathered[:, j].reshape(-1, gathered.shape[-1])
    return out


def gather_features(feature_dicts):
    """[{mod: [B_local, D]}, ...] -> same structure with [B_global, D] tensors (identity when not distributed)."""
    if not is_dist():
        return feature_dicts
    packed, keys = pack_features(feature_dicts)
    return unpack_gathered(exchange_packed(packed), keys, len(feature_dicts))


@torch.no_grad()
def exchange_loss_chunks(head):
    """The one collective inside the row-sharded loss head (ops.ShardedLossHead): every rank's log-sum-exps / diagonal block means /
    partial loss terms, ~70 KB per rank at a global batch of 2048 -- latency-bound, one call."""
    dist.all_gather_into_tensor(head.chunks.view(-1), head.send)


def shard_loss_head():
    """Row-shard the loss head over the ranks?  It costs one more graph-segment boundary and a small all-gather per step (~0.1 ms)
    and saves (1 - 1 / world) of the head's matrix work: a loss at 2 ranks (192 vs 175 us for the head alone), about even at 4
    (252 vs 275 us), 0.28 ms ahead at 8 (366 vs 647 us; profiles/r2_loss_head_sharded.txt) -- so by default from 6 ranks up.
    FOCAL_LOSS_SHARD=1 / 0 forces it on (any world > 1) / off."""
    if not is_dist():
        return False
    force = os.environ.get("FOCAL_LOSS_SHARD")
    if force is not None:
        return force != "0"
    return world() >= 6


def all_reduce_spans_async(arena, spans):
    """SUM-reduce [lo, hi) ranges of the gradient arena without holding the caller's stream: the collective waits for what that
    stream has enqueued so far and runs on the backend's own stream beside what comes next.  Returns the work handles
    (wait_all() before anything reads the gradients)."""
    if not is_dist():
        return []
    return [dist.all_reduce(arena.grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True) for lo, hi in spans if hi > lo]


def wait_all(works):
    for w in works:
        w.wait()  # (the caller's stream waits; the host does not block under RCCL)


def grad_reduce_dtype():
    """'fp32' (default: one all-reduce(SUM) of the fp32 arena) or 'bf16' (FOCAL_GRAD_REDUCE=bf16): half the bytes on the wire, sums kept
    in fp32 -- all_reduce_gradients_bf16."""
    v = os.environ.get("FOCAL_GRAD_REDUCE", "fp32").lower()
    if v not in ("fp32", "bf16"):
        raise ValueError(f"FOCAL_GRAD_REDUCE must be fp32 or bf16 (got {v!r})")
    return v


@torch.no_grad()
def all_reduce_gradients_bf16(arena):
    """The gradient SUM with bf16 on the wire and fp32 accumulation on arrival (45.8 -> 22.9 MB per direction for
    SW_Transformer).  An all-reduce in bf16 would add the ranks' contributions in bf16, one rounding per hop; instead the reduction is
    spelled out as its two halves:
      1. all-to-all: rank r receives every rank's bf16 copy of SHARD r of the arena and adds them up in fp32 -- its own contribution
         straight from its fp32 gradient, never rounded;
      2. all-gather of the reduced shards, rounded to bf16 once.
    Every element is therefore (W - 1) values rounded to bf16 + one exact value, summed in fp32, rounded to bf16 once: a relative error
    of a few 1e-3 of the gradient whatever the world size (tests/test_distributed_cpu.py holds it to 1e-2 of the update).  Persistent
    staging buffers (stable addresses between replays of captured graph segments); blocking on the caller's stream."""
    if not is_dist():
        return
    W, r = dist.get_world_size(), dist.get_rank()
    g = arena.grad
    n = g.numel()
    per = (n + W - 1) // W
    dev = g.device
    send = _static("g16_send", (W, per), torch.bfloat16, dev)
    recv = _static("g16_recv", (W, per), torch.bfloat16, dev)
    red = _static("g16_red", (per,), torch.bfloat16, dev)
    out = _static("g16_out", (W, per), torch.bfloat16, dev)
    flat = send.view(-1)
    flat[:n].copy_(g)           # (fp32 -> bf16, round to nearest even)
    if W * per > n:
        flat[n:].zero_()
    dist.all_to_all_single(recv.view(-1), send.view(-1))
    acc = recv.float().sum(0)   # fp32 accumulate on arrival ...
    lo, hi = r * per, min(n, (r + 1) * per)
    if hi > lo:                 # ... with this rank's own shard exact instead of rounded
        acc[:hi - lo] += g[lo:hi] - recv[r, :hi - lo].float()
    red.copy_(acc)
    dist.all_gather_into_tensor(out.view(-1), red)
    g.copy_(out.view(-1)[:n])


def all_reduce_gradients(arena, bucket_bytes=64 << 20):
    """SUM-reduce the arena's gradient buffer in place.  xGMI is point-to-point (7 links/GPU): a few large buckets
    keep every link busy; 46 MB (SW_Transformer fp32) is a single bucket."""
    if not is_dist():
        return
    if grad_reduce_dtype() == "bf16":
        return all_reduce_gradients_bf16(arena)
    n = arena.grad.numel()
    step = max(1, bucket_bytes // 4)
    for lo in range(0, n, step):
        dist.all_reduce(arena.grad[lo:lo + step], op=dist.ReduceOp.SUM)
