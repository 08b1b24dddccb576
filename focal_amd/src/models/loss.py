"""FOCALLoss (reference: models/loss.py:8-218) on the fused HIP loss head.

Same constructor and call contract: `FOCALLoss(args)(mod_features1, mod_features2) -> 0-dim tensor`.  The four
un-weighted terms of the last call are kept in `last_terms` (device tensor [shared, private, orth, rank, total]);
the reference only returns their weighted sum.
"""
import os
import sys

import torch
import torch.nn as nn

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from focal_amd import distributed as fdist, ops  # noqa: E402


class _LossHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, n_mod, *feats):
        if owner._head is not None:  # row-sharded evaluation: phase A ran in begin(), the ranks' chunks have been exchanged
            head, owner._head = owner._head, None
            terms, g1, g2, flat = head.phase_b()
        else:
            feats = [f.contiguous().float() for f in feats]
            terms, g1, g2, flat = ops.loss_head(feats[:n_mod], feats[n_mod:], owner.temperature, owner.config["inter_rank_margin"],
                                                owner.weights, owner.seq_len, owner.args.tag == "noPrivate", return_flat=True)
        owner.last_terms = terms
        ctx.grads, ctx.flat = g1 + g2, flat  # the gradients are views of one flat buffer
        return terms[4]

    @staticmethod
    def backward(ctx, gout):
        from focal_amd import runtime
        if not runtime.is_unit_grad(gout):  # (the training step hands over runtime.unit_grad: exactly 1 -- no launch)
            ctx.flat.mul_(gout)  # one launch for all 2M gradients
        grads, ctx.grads, ctx.flat = ctx.grads, None, None
        return (None, None, *grads)


class FOCALLoss(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.config = args.dataset_config["FOCAL"]
        self.modalities = args.dataset_config["modality_names"]
        self.seq_len = args.dataset_config["seq_len"]
        t = self.config["temperature"]
        self.temperature = t[args.model] if isinstance(t, dict) else t
        self.weights = (self.config["shared_contrastive_loss_weight"], self.config["private_contrastive_loss_weight"],
                        self.config["orthogonal_loss_weight"], self.config["rank_loss_weight"])
        self.last_terms = None
        self._head, self._feats = None, None

    def forward(self, mod_features1, mod_features2, index=None):
        """Under data parallelism the features are the all-gathered global batch (focal_amd.distributed.gather_features) and the
        head is evaluated row-sharded: every rank its own samples' rows of every cross-sample matrix, one small all-gather in the
        middle (begin -> exchange -> finish; bench.py captures the two halves into separate graph segments)."""
        if fdist.shard_loss_head():
            self.begin(mod_features1, mod_features2)
            fdist.exchange_loss_chunks(self._head)
            return self.finish()
        feats = [mod_features1[m] for m in self.modalities] + [mod_features2[m] for m in self.modalities]
        return _LossHeadFn.apply(self, len(self.modalities), *feats)

    def begin(self, mod_features1, mod_features2):
        self._feats = [mod_features1[m] for m in self.modalities] + [mod_features2[m] for m in self.modalities]
        n_mod = len(self.modalities)
        with torch.no_grad():
            fl = [f.detach().contiguous().float() for f in self._feats]
            self._head = ops.ShardedLossHead(fdist.rank(), fdist.world())
            self._head.phase_a(fl[:n_mod], fl[n_mod:], self.temperature, self.config["inter_rank_margin"], self.weights, self.seq_len,
                               self.args.tag == "noPrivate")
        return self._head

    def finish(self):
        feats, self._feats = self._feats, None
        return _LossHeadFn.apply(self, len(self.modalities), *feats)


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        loss, dlogits = ops.cross_entropy(logits, labels)
        ctx.save_for_backward(dlogits)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss() of the finetune / supervised stages (reference: train_utils/model_selection.py:39-45), value and
    gradient from one HIP launch (focal_cross_entropy); one-hot / soft labels are reduced to class indices as the reference's
    evaluation does."""

    def forward(self, logits, labels):
        if labels.dim() > 1:
            labels = labels.argmax(dim=1)
        return _CrossEntropyFn.apply(logits, labels.to(logits.device))
