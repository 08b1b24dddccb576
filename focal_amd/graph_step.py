"""The pretraining step as a replayed hipGraph (what bench.py measures, for `train.py`).

`train_utils/pretrain.py` launches ~700 kernels per SW_Transformer step through ctypes; eagerly that is host-bound (~10 ms per step
against 6.5 ms of GPU work).  `CapturedTrainStep` captures [zero_grad -> FOCAL(view 1, view 2) -> loss head -> backward -> AdamW]
once the shapes have repeated, and replays it; what stays eager per step is the part whose kernel sequence depends on host draws --
the two augmented views (Augmenter.forward("random"): one DFT launch per modality and view, plus a warp pass when drawn), written
into address-stable two-view buffers (`Augmenter.static_views`) that the captured step reads.  Dropout seeds and the AdamW step
counter advance on the device, the learning rate is a device word: replays differ as steps should.
A batch of another shape (the last batch of an epoch) or a data-parallel job (collectives between the segments) runs eagerly."""
import torch

from . import distributed


class CapturedTrainStep:
    def __init__(self, model, loss_func, optimizer, warm_steps=2, enabled=True):
        self.model, self.loss_func, self.opt = model, loss_func, optimizer
        self.warm_steps, self.enabled = warm_steps, enabled and not distributed.is_dist()
        self.key, self.graph, self.seen = None, None, 0
        self.loss = None
        self.replays = self.eager_steps = 0

    def _body(self, v1, v2):
        self.opt.zero_grad()
        f1, f2 = self.model(v1, v2, proj_head=True)
        f1, f2 = distributed.gather_features([f1, f2])
        loss = self.loss_func(f1, f2)
        loss.backward()
        self.opt.step()
        return loss.detach()

    @staticmethod
    def _key(v1, v2):
        return tuple((t.data_ptr(), tuple(t.shape)) for v in (v1, v2) for mods in v.values() for t in mods.values())

    def __call__(self, v1, v2):
        """One optimiser step on the two views; returns the (device) loss of the step."""
        key = self._key(v1, v2)
        if self.graph is not None and key == self.key:
            self.opt.sync_lr()
            self.graph.replay()
            self.replays += 1
            return self.loss
        self.eager_steps += 1
        out = self._body(v1, v2)
        if not self.enabled or self.graph is not None:
            return out
        # capture once the same buffers have come round `warm_steps` times (arena, moments and workspaces exist by then)
        self.seen = self.seen + 1 if key == self.key else 1
        self.key = key
        if self.seen >= self.warm_steps:
            try:
                self._capture(v1, v2)
            except Exception as e:  # noqa: BLE001 -- capture is an optimisation: stay eager, say so once
                import logging
                logging.warning(f"hipGraph capture of the training step unavailable ({type(e).__name__}: {e}); running eagerly")
                self.enabled, self.graph = False, None
                torch.cuda.synchronize()
        return out

    def _capture(self, v1, v2):
        torch.cuda.synchronize()
        self.opt.sync_lr()
        self.loss = torch.zeros((), device=next(iter(next(iter(v1.values())).values())).device)
        side = torch.cuda.Stream()
        pool, keep = None, []
        with torch.cuda.stream(side):
            for _ in range(2):  # captured twice into one pool, the second set is replayed (bench.py: Step.capture)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, stream=side):
                    self.loss.copy_(self._body(v1, v2))
                pool = g.pool()
                keep.append(g)
        torch.cuda.synchronize()
        self._keep, self.graph = keep, keep[-1]
