"""Flat parameter arena: the HBM layout of everything the optimizer and the gradient all-reduce touch.

All parameters that receive a gradient in FOCAL pretraining ("hot" parameters) live back to back in ONE fp32
buffer, with a same-shaped gradient buffer, optional AdamW moment buffers and (bf16 mode) a bf16 shadow used as the
matrix-core operand.  `nn.Parameter.data` / `.grad` of the reference-named modules are views into these buffers, so
`state_dict()`, `load_state_dict()` and `optimizer.zero_grad()` keep working, while
  * AdamW is one fused streaming kernel over the arena (28 B/param of HBM traffic, + 2 B for the shadow),
  * the data-parallel gradient reduction is one RCCL all-reduce over `grad` (bucketed views of it),
  * weight-gradient GEMMs accumulate straight into `grad` with fp32 atomics (both views of a step add up there).
Parameters outside the hot set (frozen patch embedding, classifier head, fusion block, ...) are left where they
are with `.grad = None`, which is exactly how `torch.optim.AdamW` skips them in the reference
(train_utils/optimizer.py:27-32).
"""
import torch

from . import ops

ALIGN = 8  # elements: keeps every segment 32-byte (fp32) / 16-byte (bf16) aligned for vector loads


class ParamArena:
    def __init__(self, module, is_hot, compute_dtype):
        named = [(n, p) for n, p in module.named_parameters()]
        hot = [(n, p) for n, p in named if is_hot(n)]
        if not hot:
            raise ValueError("ParamArena: no hot parameters")
        dev = hot[0][1].device
        if dev.type != "cuda":
            raise ops._lib.FocalHipError("the FOCAL HIP path needs the model on a ROCm device (no CPU fallback)")
        self.device = dev
        self.compute_dtype = compute_dtype
        self.index = {}
        off = 0
        for n, p in hot:
            self.index[n] = (off, p.numel(), tuple(p.shape))
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.size = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=dev) if compute_dtype == torch.bfloat16 else None
        self.exp_avg = None
        self.exp_avg_sq = None
        self.params = {}
        with torch.no_grad():
            for n, p in hot:
                o, k, shp = self.index[n]
                self.flat[o:o + k].copy_(p.data.reshape(-1))
                p.data = self.flat[o:o + k].view(shp)
                p.grad = self.grad[o:o + k].view(shp)
                p._focal_arena = self
                self.params[n] = p
        self._shadow_version = -1
        self.sync_shadow()

    # ---- views
    def master(self, name):
        o, k, shp = self.index[name]
        return self.flat[o:o + k].view(shp)

    def g(self, name):
        o, k, shp = self.index[name]
        return self.grad[o:o + k].view(shp)

    def operand(self, name):
        """The tensor handed to the matrix cores for weight `name`: the fp32 master (fp32 mode) or its bf16 shadow."""
        o, k, shp = self.index[name]
        src = self.flat if self.shadow is None else self.shadow
        return src[o:o + k].view(shp)

    def spans(self, names):
        """Maximal contiguous [lo, hi) element ranges of the arena covered by `names` (alignment gaps between neighbours included)."""
        sel = sorted((self.index[n][0], self.index[n][0] + (self.index[n][1] + ALIGN - 1) // ALIGN * ALIGN) for n in names)
        out = []
        for lo, hi in sel:
            if out and lo <= out[-1][1]:
                out[-1][1] = max(out[-1][1], hi)
            else:
                out.append([lo, hi])
        return [(lo, min(hi, self.size)) for lo, hi in out]

    def owns(self, p):
        return getattr(p, "_focal_arena", None) is self and p.data_ptr() >= self.flat.data_ptr() and \
            p.data_ptr() < self.flat.data_ptr() + self.flat.numel() * 4

    def intact(self):
        """False once something (e.g. `module.to()` onto another device) re-bound a parameter away from the arena."""
        return all(self.owns(p) for p in self.params.values())

    # ---- bf16 shadow
    def sync_shadow(self, force=False):
        if self.shadow is None:
            return
        v = self.flat._version
        if force or v != self._shadow_version:
            ops.cast_bf16(self.flat, self.shadow)
            self._shadow_version = v

    def mark_shadow_fresh(self):
        self._shadow_version = self.flat._version

    def zero_grad(self):
        self.grad.zero_()
        ops.zero_pool_reset(self.grad.device)  # the step's small accumulation buffers (ops.pool_zeros): one launch for all of them

    def moments(self):
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.flat)
            self.exp_avg_sq = torch.zeros_like(self.flat)
        return self.exp_avg, self.exp_avg_sq
