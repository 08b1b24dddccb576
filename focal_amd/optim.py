"""Fused AdamW over the parameter arena (reference: torch.optim.AdamW built in train_utils/optimizer.py:27-32).

Keeps the torch.optim.Optimizer surface the reference loop and timm-style schedulers use (`param_groups[i]["lr"]`,
`zero_grad()`, `step()`), but `step()` is one streaming HIP kernel over the arena's hot region; parameters whose
`.grad` is None (frozen patch embedding, unused heads) are skipped entirely, exactly as torch does.
"""
import torch

from . import distributed, ops, runtime


class FocalAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, l2_decay=False):
        """l2_decay=True gives torch.optim.Adam semantics (g += wd * p): the reference's finetune optimizer."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._l2 = bool(l2_decay)
        self._lr_dev = None
        self._lr_host = None
        self._step_state = None  # device {., step}: this optimizer's own step counter (AdamW bias correction)

    def _arenas(self):
        arenas, seen = [], set()
        for group in self.param_groups:
            for p in group["params"]:
                ar = getattr(p, "_focal_arena", None)
                if ar is None or not ar.owns(p):
                    if p.grad is not None:
                        raise ops._lib.FocalHipError("FocalAdamW: a parameter with a gradient lives outside the arena")
                    continue
                if id(ar) not in seen:
                    seen.add(id(ar))
                    arenas.append(ar)
        return arenas

    def _span(self, ar):
        """[lo, hi) of the arena this optimizer updates.  Pretraining: everything.  Finetuning hands the optimizer only the
        head's parameters (general_utils/weight_utils.py:61-80): they sit back to back at the arena's end, and the frozen
        encoder weights in front of them must not even see weight decay -- torch skips parameters it was not given."""
        mine = {id(p) for g in self.param_groups for p in g["params"]}
        owned = [(ar.index[n][0], ar.index[n][1], id(p) in mine) for n, p in ar.params.items()]
        if all(t for _, _, t in owned):
            return 0, ar.size
        sel = [(o, k) for o, k, t in owned if t]
        lo, hi = min(o for o, _ in sel), max(o + k for o, k in sel)
        hi = min(ar.size, (hi + 7) // 8 * 8)
        if any((lo <= o < hi) and not t for o, _, t in owned):
            raise ops._lib.FocalHipError("FocalAdamW: the parameters to update are not contiguous in the arena")
        return lo, hi

    def zero_grad(self, set_to_none=True):
        # gradients are views of the arena's grad buffer: zero the buffer, keep the views
        for ar in self._arenas():
            runtime.join_all(ar.device)  # encoder backward kernels run on side streams and write the arena directly
            ar.zero_grad()

    def sync_lr(self):
        """Push param_groups[0]['lr'] to the device scalar the kernel reads (call outside graph capture)."""
        lr = float(self.param_groups[0]["lr"])
        if self._lr_dev is None:
            dev = self._arenas()[0].device
            self._lr_dev = torch.empty(1, dtype=torch.float32, device=dev)
        if lr != self._lr_host:
            self._lr_dev.fill_(lr)
            self._lr_host = lr

    @torch.no_grad()
    def reduce_gradients(self):
        """Data parallel: exact global-batch gradient = SUM over ranks of the flat gradient arena (no-op on one rank)."""
        arenas = self._arenas()
        if arenas:
            runtime.join_all(arenas[0].device)
        for ar in arenas:
            distributed.all_reduce_gradients(ar)

    @torch.no_grad()
    def reduce_buckets(self, model):
        """Data parallel, split backward pass (HipBackbone.split_backward): ([spans final after the first phase], [the rest]) per
        arena, or None when the model's backward cannot be split (DeepSense; supervised stages)."""
        bb = getattr(model, "backbone", model)
        if distributed.grad_reduce_dtype() != "fp32":
            return None   # (the bf16-wire reduction is an all-to-all + all-gather over the whole arena: one blocking call behind backward)
        if not hasattr(bb, "final_after_first_phase") or not hasattr(bb, "_encoders"):
            return None
        if not all(hasattr(e, "backward_rest") for e in bb._encoders.values()):
            return None
        arenas = self._arenas()
        if len(arenas) != 1 or arenas[0] is not bb.arena():
            return None
        ar = arenas[0]
        first = set(bb.final_after_first_phase())
        rest = [n for n in ar.index if n not in first]
        if not rest:
            return None
        return ar, ar.spans(first), ar.spans(rest)

    @torch.no_grad()
    def reduce_async(self, ar, spans):
        runtime.join_all(ar.device)
        self._works = getattr(self, "_works", []) + distributed.all_reduce_spans_async(ar, spans)

    def wait_reductions(self):
        distributed.wait_all(getattr(self, "_works", []))
        self._works = []

    @torch.no_grad()
    def step(self, closure=None, reduce=True):
        """reduce=False: the caller already ran reduce_gradients() (bench.py does, eagerly, between captured graph segments)."""
        arenas = self._arenas()
        if not arenas:
            raise ops._lib.FocalHipError("FocalAdamW: no arena-backed parameters (run the backbone on the GPU first)")
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        runtime.join_all(arenas[0].device)
        if reduce:
            self.reduce_gradients()
        g0 = self.param_groups[0]
        dev = arenas[0].device
        if self._step_state is None:
            self._step_state = ops.new_step_state(dev)
        # this optimizer's step count (on the device: graph-replay safe) and the dropout seed of the next forward pass are advanced
        # by the AdamW kernel itself -- its last workgroup -- instead of two one-thread launches on the serial tail of the step
        segs = []
        for ar in arenas:
            m, v = ar.moments()
            lo, hi = self._span(ar)
            segs.append((ar.flat[lo:hi], ar.grad[lo:hi], m[lo:hi], v[lo:hi], ar.shadow[lo:hi] if ar.shadow is not None else None))
        ops.adamw_multi(segs, self._lr_dev, self._step_state, g0["betas"][0], g0["betas"][1], g0["eps"],
                        g0["weight_decay"], self._l2, advance=True, seed_state=runtime.rng_state(dev))
        for ar in arenas:
            ar.mark_shadow_fresh()

    # ---- resume support (the reference saves weights only; SURVEY 8f rank 3 asks for optimizer state as well)
    def train_state(self):
        """Everything `step()` depends on besides the weights: the step count and both moment buffers of every arena."""
        torch.cuda.synchronize()
        arenas = self._arenas()
        return {"step": int(self._step_state[1].item()) if self._step_state is not None else 0,
                "lr": float(self.param_groups[0]["lr"]),  # informational: on resume the replayed scheduler owns the LR
                "moments": [tuple(t.detach().cpu().clone() for t in ar.moments()) for ar in arenas]}

    def load_train_state(self, state):
        arenas = self._arenas()
        if len(arenas) != len(state["moments"]):
            raise ops._lib.FocalHipError("FocalAdamW.load_train_state: arena count differs (run one forward/backward first)")
        for ar, (m, v) in zip(arenas, state["moments"]):
            am, av = ar.moments()
            if am.numel() != m.numel():
                raise ops._lib.FocalHipError("FocalAdamW.load_train_state: moment size differs from the arena")
            am.copy_(m.to(am.device))
            av.copy_(v.to(av.device))
        dev = arenas[0].device
        if self._step_state is None:
            self._step_state = ops.new_step_state(dev)
        self._step_state[1] = int(state["step"])
