"""Data-parallel FOCAL pretraining: one process per GPU, RCCL over xGMI through torch.distributed.

The path shards by whole subsequences.  There is exactly one exchange step on the data path:
  * all-gather of the projected embeddings (2 views x M modalities, packed into ONE buffer = one RCCL call,
    ~1 MB per rank: latency-bound on xGMI, so never more than one call), after which every rank evaluates the loss
    head on the global batch (global negatives / global ranking) and back-propagates only its own slice;
  * gradients: because every rank differentiates the SAME global loss w.r.t. its own samples, the exact gradient of
    the global-batch loss is the SUM over ranks -> one all-reduce(SUM) over the flat gradient arena (no averaging).
"""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class _AllGatherFeatures(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *feats):
        W, r = dist.get_world_size(), dist.get_rank()
        packed = torch.stack([f.contiguous() for f in feats], 0)  # [n, B_local, D]
        n = packed.shape[0]
        flat = torch.empty((W * n,) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
        dist.all_gather_into_tensor(flat, packed)  # concatenation along dim 0: [rank0 feats.., rank1 feats.., ...]
        out = flat.view((W, n) + tuple(packed.shape[1:]))
        ctx.rank, ctx.b = r, packed.shape[1]
        # rank-major concat keeps subsequences contiguous (models/loss.py:152-155 reshapes [B] -> [b, seq])
        return tuple(out[:, i].reshape(-1, packed.shape[2]) for i in range(len(feats)))

    @staticmethod
    def backward(ctx, *grads):
        lo = ctx.rank * ctx.b
        return tuple(g[lo:lo + ctx.b].contiguous() for g in grads)


def gather_features(feature_dicts):
    """[{mod: [B_local, D]}, ...] -> same structure with [B_global, D] tensors (identity when not distributed)."""
    if not is_dist():
        return feature_dicts
    keys = [(i, m) for i, d in enumerate(feature_dicts) for m in d]
    flat = _AllGatherFeatures.apply(*[feature_dicts[i][m] for i, m in keys])
    out = [dict() for _ in feature_dicts]
    for (i, m), t in zip(keys, flat):
        out[i][m] = t
    return out


def all_reduce_gradients(arena, bucket_bytes=64 << 20):
    """SUM-reduce the arena's gradient buffer in place.  xGMI is point-to-point (7 links/GPU): a few large buckets
    keep every link busy; 46 MB (SW_Transformer fp32) is a single bucket."""
    if not is_dist():
        return
    n = arena.grad.numel()
    step = max(1, bucket_bytes // 4)
    for lo in range(0, n, step):
        dist.all_reduce(arena.grad[lo:lo + step], op=dist.ReduceOp.SUM)
