"""Closed-form CPU restatement of FOCALLoss (test infrastructure; see oracle/__init__.py).

The reference builds the InfoNCE logits by materialising a [seq, 2b, 2b, d] broadcast and index-masking
(models/loss.py:48-87); algebraically that is `mean_i [logsumexp_{j != i} S[i, j] - S[i, pos(i)]]`, which is what
is written here (and what the HIP loss head computes).
"""
import torch


def _unit(x, eps):
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


def info_nce(e1, e2, temperature):
    """forward_contrastive_loss(finegrain=False), models/loss.py:48-87.  e1, e2: [b, seq, d]."""
    b = e1.shape[0]
    z = _unit(torch.cat([e1.transpose(0, 1), e2.transpose(0, 1)], 1), 1e-8)  # [seq, 2b, d]; CosineSimilarity eps
    S = z @ z.transpose(1, 2) / temperature  # [seq, 2b, 2b]
    n = 2 * b
    idx = torch.arange(n)
    pos = S[:, idx, (idx + b) % n]  # diagonals at +-b, loss.py:75-79
    S = S.masked_fill(torch.eye(n, dtype=torch.bool)[None], float("-inf"))  # self-similarity removed, loss.py:35-42
    return (torch.logsumexp(S, -1) - pos).mean()  # CrossEntropy(label 0) over [pos, negatives], loss.py:83-85


def orthogonality(e1, e2):
    """forward_orthogonality_loss, models/loss.py:89-106: CosineEmbeddingLoss(target=-1, margin=0, mean)."""
    x1, x2 = e1.reshape(-1, e1.shape[-1]), e2.reshape(-1, e2.shape[-1])
    eps = 1e-12  # ATen cosine_embedding_loss adds EPSILON to the squared norms
    cos = (x1 * x2).sum(-1) / torch.sqrt(((x1 * x1).sum(-1) + eps) * ((x2 * x2).sum(-1) + eps))
    return cos.clamp_min(0).mean()


def temporal_ranking(e, margin):
    """forward_temporal_inter_ranking_loss, models/loss.py:108-137.  e: [b, seq, d] (un-normalised)."""
    b, seq, d = e.shape
    x = e.reshape(b * seq, d)
    D = (x[:, None, :] - x[None, :, :]).norm(dim=-1)  # torch.cdist(p=2), loss.py:117
    D = D.view(b, seq, b, seq).permute(0, 2, 1, 3)  # [b, b, seq, seq]
    keep = 1.0 - torch.eye(b * seq, dtype=e.dtype).view(b, seq, b, seq).permute(0, 2, 1, 3)
    Dbar = (D * keep).sum((2, 3)) / keep.sum((2, 3))  # block means: /12 on the diagonal, /16 elsewhere
    intra = torch.diagonal(Dbar)[:, None]  # [b, 1]
    hinge = (intra - Dbar + margin).clamp_min(0)  # MarginRankingLoss(y=-1): max(0, x1 - x2 + margin)
    off = ~torch.eye(b, dtype=torch.bool)
    return hinge[off].mean()


def focal_loss_terms(feat1, feat2, cfg, model, tag=None):
    """FOCALLoss.forward, models/loss.py:139-218.  feat{1,2}: {mod: [B, emb]}.  Returns the four un-weighted
    terms and the weighted total."""
    fc = cfg["FOCAL"]
    T = fc["temperature"][model] if isinstance(fc["temperature"], dict) else fc["temperature"]
    seq = cfg["seq_len"]
    mods = cfg["modality_names"]
    views = []
    for feats in (feat1, feat2):
        views.append({m: feats[m].reshape(-1, seq, feats[m].shape[-1]) for m in mods})
    half = views[0][mods[0]].shape[-1] // 2
    sh = [{m: v[m][..., :half] for m in mods} for v in views]
    pr = [{m: v[m][..., half:2 * half] for m in mods} for v in views]

    shared = 0.0
    for vi in range(2):
        src = views[vi] if tag == "noPrivate" else sh[vi]
        for i, m1 in enumerate(mods):
            for m2 in mods[i + 1:]:
                shared = shared + info_nce(src[m1], src[m2], T)
    private = 0.0
    for m in mods:
        private = private + info_nce(pr[0][m], pr[1][m], T)
    rank = 0.0
    for vi in range(2):
        for m in mods:
            rank = rank + temporal_ranking(views[vi][m], fc["inter_rank_margin"])
    orth = 0.0
    for vi in range(2):
        for i, m in enumerate(mods):
            orth = orth + orthogonality(sh[vi][m], pr[vi][m])
            for m2 in mods[i + 1:]:
                orth = orth + orthogonality(pr[vi][m], pr[vi][m2])
    total = (shared * fc["shared_contrastive_loss_weight"] + private * fc["private_contrastive_loss_weight"]
             + orth * fc["orthogonal_loss_weight"] + rank * fc["rank_loss_weight"])
    return dict(shared=shared, private=private, orth=orth, rank=rank, total=total)
