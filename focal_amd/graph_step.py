"""The pretraining step as replayed hipGraphs -- one implementation for `bench.py` and for `train.py`.

`train_utils/pretrain.py` launches ~280 kernels per SW_Transformer step through ctypes; eagerly that is host-bound (~10 ms per step
against 6 ms of GPU work).  `StepSegments` is the step body [zero_grad -> views -> FOCAL(view 1, view 2) -> loss head -> backward ->
AdamW] cut where the data-parallel collectives sit:

    A : zero_grad, views, both backbone passes, pack the embeddings       -> [all-gather of the embeddings]
    B : loss head on the global batch, backward                            -> [all-reduce of the gradient arena]
    C : AdamW, loss value
    (SW_Transformer: the backward pass stops where 88 % of the gradient bytes are final -- the last stage, mod_in and the projectors --
     and the rest, R, runs beside the all-reduce of that first bucket:  B -> [async all-reduce, bucket 1] -> R -> [all-reduce, bucket 2])
    (with the loss head row-sharded over the ranks B splits once more at the head's one small collective:
     B1 similarity / distance rows, log-sum-exps, hinges  -> [all-gather of ~70 KB chunks] ->  B2 coefficients, dL/dz, backward)

On one rank the whole step is ONE hipGraph; on N ranks every segment is a hipGraph (one shared memory pool) and the collectives are
issued eagerly between their replays (RCCL calls stay outside capture), so 1 ... 8 ranks all replay graphs.  Dropout seeds and the
AdamW step counter advance on the device, the learning rate is a device word: replays differ as steps should.

`CapturedTrainStep` is the training loop's wrapper: what stays eager per step is the part whose kernel sequence depends on host draws
-- the two augmented views (Augmenter.forward("random"): one DFT launch per modality and view, plus a warp pass when drawn), written
into address-stable two-view buffers (`Augmenter.static_views`) that the captured step reads.  A batch of another shape (the last
batch of an epoch) runs eagerly."""
import logging
import os

import torch

from . import distributed, runtime


class CaptureAborted(RuntimeError):
    """A segment failed to capture on SOME rank; every rank raises this at the same point, before the next collective is issued."""


class SegmentTimer:
    """HIP events between the pieces of a replayed data-parallel step (graph segments and the eager collectives between them): what the
    first RCCL runs will be diagnosed with (bench.py `dp`).  mark(name) closes the piece `name` on the current stream."""

    def __init__(self):
        self.steps, self.cur = [], None

    def begin(self):
        self.cur = [("start", self._event())]

    @staticmethod
    def _event():
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        return e

    def mark(self, name):
        self.cur.append((name, self._event()))

    def end(self):
        self.steps.append(self.cur)
        self.cur = None

    def medians_us(self):
        """{piece: median microseconds over the timed steps}, in step order."""
        torch.cuda.synchronize()
        out = {}
        for st in self.steps:
            for (_, e0), (name, e1) in zip(st[:-1], st[1:]):
                out.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3)
        med = lambda v: sorted(v)[len(v) // 2]
        return {k: round(med(v), 1) for k, v in out.items()}


class StepSegments:
    def __init__(self, model, loss_fn, opt, views, device):
        """views(): -> (view1, view2) frequency-domain inputs; called inside segment A (bench.py: the DFT of the resident windows is
        part of the step; train.py: the augmenter has already written the static view buffers, views() just returns them)."""
        self.model, self.loss_fn, self.opt, self.views = model, loss_fn, opt, views
        self.dist = distributed
        self.loss = torch.zeros((), device=device)
        self.feats = None
        self.device = device
        self.timer = None      # a SegmentTimer while bench.py measures the pieces of the replayed step
        self._agreed = 0       # per-segment agreements issued by the capture in progress
        self._buckets = False  # decided at the first data-parallel step (the arena exists by then)

    def buckets(self):
        """(arena, [spans final after the first backward phase], [the other spans]) or None: one blocking all-reduce after backward."""
        if self._buckets is False:
            self._buckets = None
            if self.dist.is_dist() and os.environ.get("FOCAL_NO_SPLIT_BACKWARD") != "1":
                self._buckets = self.opt.reduce_buckets(self.model)
        return self._buckets

    def _backward(self, loss):
        bb = getattr(self.model, "backbone", None)
        split = self.buckets() is not None and bb is not None
        if split:
            bb.split_backward = True
        try:
            # the root gradient is a cached scalar 1 (autograd's ones_like(loss) is a fill launch per step; the loss head recognises this
            # tensor and skips its x1 multiply: runtime.unit_grad)
            unit = runtime.unit_grad(loss.device) if (loss.dim() == 0 and loss.dtype == torch.float32 and loss.is_cuda) else None
            if unit is not None:
                loss.backward(gradient=unit)
            else:
                loss.backward()
        finally:
            if split:
                bb.split_backward = False

    def _keep_loss(self, loss):
        """self.loss = the step's loss.  A reference, not a copy: under replay the loss head writes the same graph-pool address every step
        (holding the tensor keeps its block out of the pool's hands), and the copy was a one-workgroup launch on the serial stretch between
        the end of backward and AdamW."""
        t = loss.detach()
        self.loss = t if (t.dim() == 0 and t.dtype == torch.float32) else t.float().reshape(())

    def seg_rest(self):
        """The parked part of a split backward pass (the encoder stages in front of the last one)."""
        self.model.backbone.backward_continue()

    def reduce_first(self):
        ar, first, _ = self.buckets()
        self.opt.reduce_async(ar, first)

    def reduce_second(self):
        ar, _, rest = self.buckets()
        self.opt.reduce_async(ar, rest)
        self.opt.wait_reductions()

    def seg_a(self):
        self.opt.zero_grad()
        v1, v2 = self.views()
        self.feats = self.model(v1, v2, proj_head=True)
        if self.dist.is_dist():
            self.packed, self.keys = self.dist.pack_features(list(self.feats))

    def exchange(self):
        if self.dist.is_dist():
            self.gathered = self.dist.exchange_packed(self.packed)  # the collective only: nothing else runs between segments

    def seg_b(self):
        if self.dist.is_dist():
            self.feats = self.dist.unpack_gathered(self.gathered, self.keys, 2)
        loss = self.loss_fn(*self.feats)
        self._backward(loss)
        self._keep_loss(loss)
        self.feats = None

    def seg_b1(self):
        self.feats = self.dist.unpack_gathered(self.gathered, self.keys, 2)
        self.head = self.loss_fn.begin(*self.feats)

    def exchange_head(self):
        self.dist.exchange_loss_chunks(self.head)

    def seg_b2(self):
        loss = self.loss_fn.finish()
        self._backward(loss)
        self._keep_loss(loss)
        self.feats = None

    def reduce(self):
        """Everything between the backward segment and AdamW: one blocking all-reduce, or (split backward pass) the first bucket's
        all-reduce started, the rest of backward run beside it, then the second bucket."""
        if self.buckets() is None:
            self.opt.reduce_gradients()
            return
        self.reduce_first()
        self.seg_rest()
        self.reduce_second()

    def seg_c(self):
        self.opt.step(reduce=False)

    def run(self):
        """One eager step (also what the captures record)."""
        self.seg_a()
        self.exchange()
        if self.dist.shard_loss_head():
            self.seg_b1()
            self.exchange_head()
            self.seg_b2()
        else:
            self.seg_b()
        self.reduce()
        self.seg_c()

    def _capture_segment(self, graph, fn, pool, stream, mode):
        """One segment into one hipGraph.  Under data parallelism the ranks agree on the outcome BEFORE anybody issues the collective that
        follows the segment: a rank whose capture failed (out of memory, a capture error on one rank) used to run into
        agree()'s all-reduce while its peers sat in the all-gather / all-reduce between the segments -- mismatched collectives, a hang
        until the process-group timeout.  Now every rank leaves at the same point with CaptureAborted and the step state is cleaned up."""
        err = None
        try:
            with torch.cuda.graph(graph, pool=pool, stream=stream, **mode):
                fn()
        except Exception as e:  # noqa: BLE001
            err = e
            torch.cuda.synchronize()
        if err is not None and not self.dist.is_dist():
            self.abort_capture()
            raise err
        self._agreed += 1
        if not agree(err is None, self.device):
            self.abort_capture()
            raise CaptureAborted(f"{type(err).__name__}: {err}" if err is not None else "another rank failed to capture this segment")

    def abort_capture(self):
        """Leave nothing of a half-captured step behind: parked backward halves hold tensors of the aborted graph pool (an eager
        backward_continue() would run them into the arena gradients), split_backward must not stay set, outstanding all-reduce handles
        are waited for (every rank issued them: the ranks agreed on the segment before) and dropped."""
        bb = getattr(self.model, "backbone", None)
        if bb is not None:
            bb.pending_backward = []
            bb.split_backward = False
        self.opt.wait_reductions()
        self.feats = None

    def capture(self, stream):
        """Returns the replay callable: one hipGraph of the whole step on one rank; with N > 1 ranks, hipGraphs of the segments (one
        shared memory pool) with the collectives issued eagerly between their replays.

        Everything is captured TWICE into the same graph-private pool and the second set is the one replayed: the first capture
        grows the pool segment by segment, the second sub-allocates the same tensors from the segments that now exist, and
        that placement replays 2.5 % faster (8.29 -> 8.08 ms, reproducible; the first set is kept alive so its blocks stay put)."""
        self.opt.sync_lr()
        multi = self.dist.is_dist()
        if not multi:
            return self._capture_all(stream, multi)
        # N ranks: every rank must issue the SAME sequence of agreements.  A rank that fails BETWEEN two segment captures (an eager
        # allocation, the pool hand-over) would go straight to its caller's final agreement while its peers sit in the next segment's --
        # mispaired, and the peers' own final agreement would then wait for nobody.  Such a rank takes the next segment's slot
        # with a "no": its peers raise CaptureAborted there, and everybody meets again in the final agreement.  (After the last segment's
        # slot nothing is owed: the final agreement is next on every rank.)
        self._agreed = 0
        slots = 2 * (3 + (1 if self.dist.shard_loss_head() else 0) + (1 if self.buckets() is not None else 0))
        try:
            return self._capture_all(stream, multi)
        except CaptureAborted:
            raise
        except Exception as e:  # noqa: BLE001
            torch.cuda.synchronize()
            if self._agreed < slots:
                self._agreed += 1
                agree(False, self.device)
            self.abort_capture()
            raise CaptureAborted(f"{type(e).__name__}: {e} (between segment captures)") from e

    def _capture_all(self, stream, multi):
        # with a process group alive its watchdog thread polls events (cudaEventQuery) at any time: under the default "global"
        # capture mode that would invalidate a capture in progress, "thread_local" restricts the checks to the capturing thread
        mode = {"capture_error_mode": "thread_local"} if multi else {}
        pool, self._warm_graphs = None, []
        for attempt in range(2):
            if not multi:
                # one rank: no collectives to interleave -> one graph for the whole step (each extra graph launch costs
                # ~0.1 ms of idle GPU per step)
                whole = torch.cuda.CUDAGraph()
                self._capture_segment(whole, self.run, pool, stream, {})
                graphs = (whole,)
            else:
                shard = self.dist.shard_loss_head()
                ga, gb, gb2, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                # (every segment: capture, then the ranks agree, then -- and only then -- the collective that follows it)
                self._capture_segment(ga, self.seg_a, pool, stream, mode)
                self.exchange()  # eager, autograd-aware: links segment B's backward to segment A's forward
                if shard:
                    self._capture_segment(gb, self.seg_b1, ga.pool(), stream, mode)
                    head = self.head
                    self.exchange_head()
                    self._capture_segment(gb2, self.seg_b2, ga.pool(), stream, mode)
                else:
                    head, gb2 = None, None
                    self._capture_segment(gb, self.seg_b, ga.pool(), stream, mode)
                gr = None
                if self.buckets() is None:
                    self.reduce()
                else:
                    self.reduce_first()
                    gr = torch.cuda.CUDAGraph()
                    self._capture_segment(gr, self.seg_rest, ga.pool(), stream, mode)
                    self.reduce_second()
                self._capture_segment(gc, self.seg_c, ga.pool(), stream, mode)
                graphs = (ga, gb, gb2, gc, gr)
            pool = graphs[0].pool()
            if attempt == 0:
                self._warm_graphs = graphs
        self._graphs = graphs
        if not multi:
            return graphs[0].replay
        ga, gb, gb2, gc, gr = graphs
        packed = self.packed

        def replay():
            t = self.timer
            mark = t.mark if t is not None else (lambda name: None)
            if t is not None:
                t.begin()
            ga.replay()
            mark("A: zero_grad, views, backbone passes, pack")
            self.dist.replay_exchange(packed)
            mark("exchange: all-gather of the embeddings")
            gb.replay()
            mark("B1: head rows" if gb2 is not None else "B: loss head, backward")
            if gb2 is not None:
                self.dist.exchange_loss_chunks(head)  # the persistent send / chunks buffers of the sharded head
                mark("exchange: all-gather of the head's chunks")
                gb2.replay()
                mark("B2: head coefficients, backward")
            if gr is None:
                self.opt.reduce_gradients()
                mark("all-reduce of the gradient arena (exposed)")
            else:
                self.reduce_first()   # bucket 1 (last stage, mod_in, projectors) travels ...
                mark("issue all-reduce of bucket 1")
                gr.replay()           # ... while the earlier stages' backward runs
                mark("R: rest of backward, beside bucket 1")
                self.reduce_second()
                mark("all-reduce: wait for bucket 1 + bucket 2 (exposed)")
            gc.replay()
            mark("C: AdamW")
            if t is not None:
                t.end()
        return replay

    def measure_pieces(self, run, steps=5):
        """`steps` more replays with HIP events between the pieces -> {piece: median us}.  Collective: every rank calls it."""
        self.timer = SegmentTimer()
        try:
            for _ in range(steps):
                run()
            return self.timer.medians_us()
        finally:
            self.timer = None


def agree(ok, device):
    """Capture failures must not be rank-dependent (a rank falling back to eager while another replays would desynchronise the
    collectives): the ranks agree on the outcome before using the graphs."""
    if not distributed.is_dist():
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
    return bool(int(flag.item()))


class CapturedTrainStep:
    def __init__(self, model, loss_func, optimizer, warm_steps=2, enabled=True):
        self.model, self.loss_func, self.opt = model, loss_func, optimizer
        # cross-rank BatchNorm statistics (DeepSense, -sync_bn) put collectives INSIDE the backbone passes: nothing there to cut a
        # segment at, so such a job stays eager
        inner_collectives = distributed.is_dist() and bool(getattr(getattr(model, "backbone", None), "sync_bn", False))
        self.warm_steps, self.enabled = warm_steps, enabled and not inner_collectives
        self.key, self.replay, self.seen = None, None, 0
        self.segments = None
        self.loss = None
        self.replays = self.eager_steps = 0

    @staticmethod
    def _key(v1, v2):
        return tuple((t.data_ptr(), tuple(t.shape)) for v in (v1, v2) for mods in v.values() for t in mods.values())

    def _segments(self, v1, v2):
        dev = next(iter(next(iter(v1.values())).values())).device
        return StepSegments(self.model, self.loss_func, self.opt, lambda: (v1, v2), dev)

    def __call__(self, v1, v2):
        """One optimiser step on the two views; returns the (device) loss of the step."""
        return self._step(self._key(v1, v2), lambda: self._segments(v1, v2), lambda: None)

    def step_from_inputs(self, augmenter, time_loc_inputs):
        """One optimiser step on a batch of TIME-domain windows (device tensors): the two random views are drawn on the device inside
        the step (Augmenter.forward_random_pair), so the replayed graph holds the whole reference loop body -- augmenter x 2, backbone,
        loss, backward, optimizer.  The batch is copied into address-stable input buffers in front of the replay (one device-to-device
        copy per modality; a loader that already fills fixed device buffers could hand those over instead)."""
        flat = [(loc, mod) for loc in time_loc_inputs for mod in time_loc_inputs[loc]]
        key = ("time",) + tuple((loc, mod, tuple(time_loc_inputs[loc][mod].shape)) for loc, mod in flat)
        st = self.__dict__.setdefault("_static_inputs", {})
        if key not in st:
            st[key] = {loc: {mod: torch.empty_like(t, dtype=torch.float32).contiguous() for mod, t in mods.items()}
                       for loc, mods in time_loc_inputs.items()}
        static = st[key]

        def load():
            for loc, mod in flat:
                static[loc][mod].copy_(time_loc_inputs[loc][mod], non_blocking=True)
        dev = static[flat[0][0]][flat[0][1]].device
        return self._step(key, lambda: StepSegments(self.model, self.loss_func, self.opt, lambda: augmenter.forward_random_pair(static), dev), load)

    def _step(self, key, make_segments, load_inputs):
        load_inputs()
        if self.replay is not None and key == self.key:
            self.opt.sync_lr()
            self.replay()
            self.replays += 1
            return self.loss
        self.eager_steps += 1
        seg = make_segments()
        seg.run()
        out = seg.loss
        if not self.enabled or self.replay is not None:
            return out
        # capture once the same buffers have come round `warm_steps` times (arena, moments and workspaces exist by then)
        self.seen = self.seen + 1 if key == self.key else 1
        self.key = key
        if self.seen >= self.warm_steps:
            # (the ranks agree segment by segment inside capture(): a segment that fails on any rank raises here on every rank, at the
            # same point.  A rank can still fail OUTSIDE a segment capture -- an eager allocation between segments, the pool hand-over --
            # after its peers have finished: the final agreement below makes every rank either replay or stay eager.)
            ok = True
            try:
                self._capture(make_segments)
            except Exception as e:  # noqa: BLE001 -- capture is an optimisation: stay eager, say so once
                ok = False
                logging.warning(f"hipGraph capture of the training step unavailable ({type(e).__name__}: {e}); running eagerly")
                torch.cuda.synchronize()
            if not agree(ok, seg.device):
                if ok:
                    logging.warning("hipGraph capture of the training step failed on another rank; running eagerly on all ranks")
                self.enabled, self.replay, self.segments = False, None, None
        return out

    def _capture(self, make_segments):
        torch.cuda.synchronize()
        seg = make_segments()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            replay = seg.capture(side)
        torch.cuda.synchronize()
        self.segments, self.loss, self.replay = seg, seg.loss, replay
