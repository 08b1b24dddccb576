#!/usr/bin/env python3
"""FOCAL pretraining throughput on MI355X: windows/sec of the full step (DFT of both views -> backbone x2 ->
FOCAL loss head -> backward -> fused AdamW), BASELINE.json's metric.

  python bench.py [--gpus N --steps K --warmup W] [--model SW_Transformer|DeepSense] [--batch 256] [--dtype bf16|fp32]
  N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)

Workload (config.workload): BASELINE.json configs[2] at N=1 -- SW_Transformer + FOCAL, bf16 operands, 256 synthetic
MOD-shaped 2-modality windows per GPU (audio [1,10,1600] @ 8 kHz, seismic [1,10,20] @ 100 Hz, N(0,1), 64
subsequences of 4), train mode (dropout 0.2 / attention dropout 0.2 / drop-path 0.1 active), random-init weights.
Views: view 1 = the window, view 2 = the window negated and scaled by 1.1 (two of the reference's FOCAL augmenters
with fixed draws), each followed by the time->frequency DFT, which is inside the timed step.  N > 1 = configs[3]:
embeddings all-gathered over RCCL so the loss sees the global batch, gradients all-reduced; weak scaling.
Prints ONE JSON line on rank 0.
"""
import argparse
import copy
import datetime
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "focal_amd", "src")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FLOP_PER_WINDOW = {"SW_Transformer/MOD": 4.376e9, "DeepSense/MOD": 0.809e9}  # SURVEY 8d (fwd + bwd, both views); other
# datasets: tests/golden/flops.json (torch.utils.flop_counter on the reference models, tests/golden/gen_flops.py)
HBM_PEAK_GBS = 8000.0
MFMA_BF16_PEAK_TF = 2500.0
MFMA_F32_PEAK_TF = 157.3


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--model", default="SW_Transformer")
    p.add_argument("--dataset", default="MOD", help="MOD (2 modalities, BASELINE configs 2-4) or HAR4 (4-modality synthetic IMU, config 5)")
    p.add_argument("--batch", type=int, default=256, help="windows per GPU")
    p.add_argument("--dtype", default="bf16")
    p.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-secondary", action="store_true", help="skip the DeepSense / HAR4 lines appended to the default single-GPU run")
    p.add_argument("--sync-bn", action="store_true", help="DeepSense under DP: cross-rank BatchNorm statistics (exact global-batch parity)")
    p.add_argument("--no-dropout", action="store_true", help="diagnostic: all dropout rates 0 (flagged in the JSON line)")
    p.add_argument("--cpu-steps", type=int, default=4)
    p.add_argument("--roofline-only", action="store_true", help="only run the dominant-kernel loop (for rocprofv3 --pmc passes)")
    p.add_argument("--roofline-iters", type=int, default=20)
    p.add_argument("--from-host", action="store_true", help="diagnostic (never the reported `value`): every step's windows start in pinned "
                   "host memory, as the reference's DataLoader hands them over; H2D on a copy stream one step ahead")
    p.add_argument("--views", default="fixed", choices=["fixed", "random", "random-host"],
                   help="fixed: identity / x * -1.1 folded into the DFT inside the captured step (the default, what `value` is quoted on); random: the "
                        "product Augmenter's views as train.py makes them since round 5 -- every draw on the device, inside the captured step "
                        "(Augmenter.forward_random_pair: focal_view_draw, the warp tables and the transforms read the drawn records); random-host: "
                        "the round-4 form, draws and warp tables on the host, eagerly before every replay, inside the timed region")
    p.add_argument("--trace-dump", default=None, help="write the library's launch trace of 5 eager steps (per-instance medians, both trace modes) to this JSON file and exit")
    p.add_argument("--no-roofline", action="store_true", help="profiling runs: skip the dominant-kernel timing loop (the JSON line then has roofline null)")
    return p.parse_args()


def make_args(cfg, model, device, dtype, dataset="MOD"):
    return argparse.Namespace(model=model, dataset=dataset, device=device, train_mode="contrastive", learn_framework="FOCAL",
                              stage="pretrain", task="vehicle_classification" if dataset == "MOD" else "activity_classification",
                              tag=None, dataset_config=cfg, compute_dtype=dtype)


def flops_per_window(model, dataset):
    key = f"{model}/{dataset}"
    f = os.path.join(ROOT, "tests", "golden", "flops.json")
    if os.path.exists(f):
        tab = json.load(open(f))
        if key in tab:
            return float(tab[key]["step_per_window"])
    return FLOP_PER_WINDOW.get(key)


class Step:
    """One pretraining step on resident time-domain windows (the reference loop body, train_utils/pretrain.py:62-74)."""

    def __init__(self, a, device):
        from focal_amd import ops
        from focal_amd import distributed
        from general_utils.weight_utils import freeze_patch_embedding
        from input_utils.yaml_utils import load_yaml
        from train_utils.model_selection import init_backbone_model, init_loss_func, init_pretrain_framework
        from train_utils.optimizer import define_optimizer
        self.ops, self.dist = ops, distributed
        cfg = load_yaml(os.path.join(ROOT, "focal_amd", "src", "data", f"{a.dataset}.yaml"))
        if a.no_dropout:  # diagnostic only (measures what the mask generation costs); never the reported configuration
            for k in ("dropout_ratio", "drop_path_rate", "attn_drop_rate"):
                cfg["SW_Transformer"][k] = 0.0
            cfg["DeepSense"]["dropout_ratio"] = 0.0
        self.cfg = cfg
        args = make_args(cfg, a.model, device, a.dtype, a.dataset)
        args.sync_bn = a.sync_bn
        torch.manual_seed(1234)
        self.backbone = init_backbone_model(args)
        self.model = init_pretrain_framework(args, self.backbone)
        self.loss_fn = init_loss_func(args)
        self.opt = define_optimizer(args, self.model.parameters())
        freeze_patch_embedding(args, self.model)
        self.model.train()
        rank = dist.get_rank() if dist.is_initialized() else 0
        g = torch.Generator().manual_seed(1234 + rank)
        self.x = {}
        for loc in cfg["location_names"]:
            self.x[loc] = {}
            for mod in cfg["modality_names"]:
                shape = (a.batch, cfg["loc_mod_in_time_channels"][loc][mod], cfg["num_segments"], cfg["loc_mod_spectrum_len"][loc][mod])
                self.x[loc][mod] = torch.randn(shape, generator=g).to(device)
        from focal_amd.graph_step import StepSegments
        self.aug = None
        self.host_draws = False
        if a.views in ("random", "random-host"):
            from data_augmenter.Augmenter import Augmenter
            self.aug = Augmenter(args)
            self.host_draws = a.views == "random-host" or not self.aug.device_draws_supported()
            if self.host_draws:
                self.aug.static_views = True  # address-stable two-view buffers: the captured step reads them
                self.draw_views()
        self.seg = StepSegments(self.model, self.loss_fn, self.opt, self.views, device)
        self.feed = None

    def enable_host_feed(self):
        """PCIe-inclusive mode: a pinned host copy of the batch (stands for the loader's pinned ring), two device staging sets and
        a copy stream; step k+1's windows cross PCIe while step k computes, the step begins with a D2D copy into its inputs."""
        flat = [(loc, mod) for loc in self.x for mod in self.x[loc]]
        self.feed = {"host": {k: self.x[k[0]][k[1]].cpu().pin_memory() for k in flat},
                     "stage": [{k: torch.empty_like(self.x[k[0]][k[1]]) for k in flat} for _ in range(2)],
                     "stream": torch.cuda.Stream(), "ready": [torch.cuda.Event(), torch.cuda.Event()],
                     "free": [torch.cuda.Event(), torch.cuda.Event()], "k": 0, "flat": flat}
        self.feed_h2d(0)

    def feed_h2d(self, slot):
        f = self.feed
        with torch.cuda.stream(f["stream"]):
            f["stream"].wait_event(f["free"][slot])
            for k in f["flat"]:
                f["stage"][slot][k].copy_(f["host"][k], non_blocking=True)
            f["ready"][slot].record(f["stream"])

    def feed_step_inputs(self):
        """Called on the compute stream before each step: consume the staged windows, start the next H2D."""
        f = self.feed
        slot = f["k"] & 1
        st = torch.cuda.current_stream()
        st.wait_event(f["ready"][slot])
        for k in f["flat"]:
            self.x[k[0]][k[1]].copy_(f["stage"][slot][k], non_blocking=True)
        f["free"][slot].record(st)
        f["k"] += 1
        self.feed_h2d(f["k"] & 1)

    def draw_views(self):
        """--views random: what train.py does before every replay (train_utils/pretrain.py): two draws of the product augmenter."""
        self.aug.begin_step()
        self.v1 = self.aug.forward("random", self.x)
        self.v2 = self.aug.forward("random", self.x)

    def views(self):
        if self.aug is not None:
            if not self.host_draws:   # the product's form since round 5: both views drawn on the device, inside the captured step
                return self.aug.forward_random_pair(self.x)
            return self.v1, self.v2
        # both views of a modality are written into the halves of one [2B, ...] tensor (what Augmenter.forward_random does for
        # back-to-back draws): SW_Transformer runs them as one batch without a concatenation
        both = {l: {m: torch.empty(2 * x.shape[0], 2 * x.shape[1], x.shape[2], x.shape[3], device=x.device) for m, x in mm.items()}
                for l, mm in self.x.items()}
        # (negation + scaling of view 2 folded into the DFT; all (view, modality) transforms in one call: the short-row modalities share a launch)
        flat = [(l, m) for l, mm in self.x.items() for m in mm]
        outs = self.ops.fft_realpack_multi([dict(x=self.x[l][m], out=both[l][m][:self.x[l][m].shape[0]]) for l, m in flat] +
                                           [dict(x=self.x[l][m], scale=-1.1, out=both[l][m][self.x[l][m].shape[0]:]) for l, m in flat])
        v1, v2 = {l: {} for l in self.x}, {l: {} for l in self.x}
        for i, (l, m) in enumerate(flat):
            v1[l][m], v2[l][m] = outs[i], outs[len(flat) + i]
        return v1, v2

    # The step itself -- capturable segments with the data-parallel collectives between them -- is the product's
    # (focal_amd/graph_step.py: StepSegments, also what train.py replays); here `views` is the DFT of the resident windows.
    def run(self):
        self.seg.run()

    def capture(self, stream):
        return self.seg.capture(stream)

    @property
    def loss(self):
        return self.seg.loss


def time_kernel(fn, iters=20):
    """Average duration (ms) of one launch of `fn` on the current stream, HIP events on that stream."""
    st = torch.cuda.current_stream()
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


# ---------------------------------------------------------------------------------------------------------------- roofline
# The dominant kernel is chosen and timed INSIDE the step (a warm back-to-back replay of one launch sits in the 256 MiB
# Infinity Cache and flattered the number by 2.6x), and timed by the library itself (event pairs recorded from Python
# bracketed allocator misses on a fresh box): while a launch trace is open (include/focal_hip.h: focal_trace_*), every kernel launch
# of libfocal_hip carries a start / stop event pair on the dispatch, so a record is the kernel's own begin -> end time -- what
# `rocprofv3 --kernel-trace` reports -- labelled with the launched kernel's symbol.  Groups are therefore rocprofv3's rows (one per
# kernel instantiation; an op that launches two kernels is two groups) and are checked against the committed trace of the same
# workload (profiles/r6_reference_<model>_<dataset>.json, made by tools/profile_round.sh from `rocprofv3 --kernel-trace --stats -M`).
# The Python side only attributes ALGORITHMIC bytes / flops: every public op of focal_amd.ops is wrapped, notes which trace records its
# call produced, and describes what the call has to move.
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_names import short_kernel_name  # noqa: E402

ROUND = "r6"


def _dw_bytes_flops(d):
    es = 2 if d.dtype == 1 else 4
    ey = 4 if d.y_dtype == 0 else es
    ex = 4 if d.x_dtype == 0 else es
    return d.M * d.N * ey + d.M * d.K * ex + d.N * d.K * 4, 2.0 * d.M * d.N * d.K


# ops of focal_amd.ops that launch nothing (descriptors, queries, allocation helpers): never traced
_NOT_LAUNCHES = {"code", "torch_dtype", "zero_pool_reset", "pool_zeros", "zeros", "drop_desc", "new_rng_state", "new_step_state", "linear_desc", "ln_desc",
                 "mlp_desc", "attn_desc", "conv_desc", "conv_in_desc", "bn_desc", "mlp_supported", "dw_group_supported", "dw_group_kind", "resid_ln_supported",
                 "bwd_data_ln_supported", "check", "linear_bwd_weight_group_workgroups", "linear", "gru_desc"}


def _tensor_bytes(args, kw):
    n = 0
    for t in list(args) + list(kw.values()):
        if isinstance(t, torch.Tensor):
            n += t.numel() * t.element_size()
        elif isinstance(t, (list, tuple)):
            n += sum(u.numel() * u.element_size() for u in t if isinstance(u, torch.Tensor))
    return n


class StepTracer:
    """Attributes algorithmic bytes / flops to the records of the library's launch trace.  Every public op of focal_amd.ops is wrapped; a
    call notes the range of trace records it produced and a description (bytes, flops, bound, instance label).  Algorithmic bytes of a call:
    the op's own formula where one is given below (weight gradients, LayerNorm backward, fused MLP, BatchNorm backward), otherwise every
    tensor argument and result once (inputs read once, outputs written once -- what a streaming or GEMM kernel has to move).  A call that
    launches several kernels without a formula of its own splits its bytes over them by launch size (threads)."""

    def __init__(self, ops):
        self.ops, self.calls, self.saved = ops, [], {}
        self.lib = __import__("focal_amd._lib", fromlist=["load"]).load()

    def _wrap(self, name, describe):
        orig = getattr(self.ops, name)
        self.saved[name] = orig
        lib = self.lib

        def traced(*args, **kw):
            n0 = lib.focal_trace_count()
            out = orig(*args, **kw)
            n1 = lib.focal_trace_count()
            if n1 > n0:
                self.calls.append((describe(out, *args, **kw), n0, n1))
            return out
        setattr(self.ops, name, traced)

    def install(self):
        ops = self.ops

        def out_bytes(out):
            return _tensor_bytes([out] if isinstance(out, torch.Tensor) else (out or ()), {})

        def dw(out, d, dy, x, dw_, db):
            b, f = _dw_bytes_flops(d)
            return dict(bytes=b, flops=f, bound="hbm", inst=f"dW[{d.N},{d.K}] over {d.M} rows", rows=d.M)

        def dwg(out, dtype_code, items, exclusive=True):
            b = sum(dy.numel() * dy.element_size() + x.numel() * x.element_size() + w.numel() * 4 for dy, x, w, _ in items)
            f = sum(2.0 * dy.shape[0] * dy.shape[1] * x.shape[1] for dy, x, _, _ in items)
            rows, C = items[0][0].shape[0], min(min(dy.shape[1], x.shape[1]) for dy, x, _, _ in items)
            return dict(bytes=b, flops=f, bound="hbm", inst=f"{len(items)} dW of a C={C} block over {rows} rows", rows=rows)

        def dwg32(out, compute_code, items, workgroups=0):
            b = sum(dy.numel() * dy.element_size() + x.numel() * x.element_size() + w.numel() * 4 for dy, x, w, _ in items)
            f = sum(2.0 * dy.shape[0] * dy.shape[1] * x.shape[1] for dy, x, _, _ in items)
            return dict(bytes=b, flops=f, bound="hbm", inst=f"{len(items)} dW with fp32 operands over {items[0][0].shape[0]} rows")

        def lnb(out, dy, x, stats, gamma, dx, accumulate, dgamma, dbeta, gather=None, desc=None, dx_masked=None, mask=None):
            rows, C = dy.shape
            es = dy.element_size()
            b = rows * C * (es + 4 + (8 if accumulate else 4) + (es if dx_masked is not None else 0)) + rows * 8
            return dict(bytes=b, flops=8.0 * rows * C, bound="hbm", inst=f"rows {rows} x C {C}", rows=rows * (4 if gather is not None else 1))

        def mlpb(out, d, gm, a, *rest, **kw):
            # recompute fc1 + dH + dX + dW1 + dW2 = 5 products of 2 M C H flops; bytes: gm, a in, da out
            return dict(bytes=d.M * d.C * 6, flops=10.0 * d.M * d.C * d.hidden, bound="mfma", inst=f"M {d.M}", rows=d.M)

        def mlpf(out, d, a, resid, *rest, **kw):
            return dict(bytes=_tensor_bytes((a, resid) + rest, kw) + out_bytes(out), flops=4.0 * d.M * d.C * d.hidden, bound="hbm", inst=f"M {d.M}", rows=d.M)

        def bnb(out, d, z, g, *rest, **kw):
            es_o = out.element_size() if isinstance(out, torch.Tensor) else 4
            n = d.rows * d.C
            return dict(bytes=n * (8 + 8 + es_o), flops=0.0, bound="hbm", inst=f"rows {d.rows} x C {d.C}",
                        parts={"bn_bwd_reduce": n * 8, "bn_bwd_apply": n * (8 + es_o)})

        def generic(name):
            def describe(out, *args, **kw):
                d = args[0] if args and hasattr(args[0], "_fields_") else None
                flops = 0.0
                if d is not None and all(hasattr(d, k) for k in ("M", "N", "K")):
                    flops = 2.0 * d.M * d.N * d.K
                elif d is not None and all(hasattr(d, k) for k in ("rows", "C_in", "C_out", "k")):
                    flops = 2.0 * d.rows * d.C_in * d.C_out * d.k
                elif d is not None and all(hasattr(d, k) for k in ("B", "T", "H")):
                    flops = 2.0 * d.B * d.T * 2 * 3 * d.H * d.H * (2 if name.endswith("bwd") else 1)
                shp = " x ".join(str(tuple(t.shape)) for t in args if isinstance(t, torch.Tensor))[:70]
                rows = max([t.shape[0] for t in list(args) + list(kw.values()) if isinstance(t, torch.Tensor) and t.dim() == 2] or [0])  # token rows the call works on
                return dict(bytes=_tensor_bytes(args, kw) + out_bytes(out), flops=flops, bound="hbm", inst=f"{name} {shp}", rows=rows)
            return describe

        def fftm(out, items):
            return dict(bytes=sum(3 * it["x"].numel() * 4 for it in items), flops=0.0, bound="hbm", inst=f"{len(items)} transforms")
        special = {"linear_bwd_weight": dw, "linear_bwd_weight_group": dwg, "linear_bwd_weight_group_f32": dwg32, "layernorm_bwd": lnb, "mlp_bwd": mlpb, "mlp_fwd": mlpf,
                   "fft_realpack_multi": fftm, "bn_act_bwd": bnb}
        for name in dir(ops):
            fn = getattr(ops, name)
            if (callable(fn) and not name.startswith("_") and name not in _NOT_LAUNCHES and not isinstance(fn, type)
                    and getattr(fn, "__module__", "") == ops.__name__):
                self._wrap(name, special.get(name) or generic(name))

    def remove(self):
        for k, v in self.saved.items():
            setattr(self.ops, k, v)

    def launches(self):
        """One dict per recorded launch, in launch order: kernel (short name), us, grid, bytes, flops, bound, inst, op."""
        from focal_amd._lib import TraceRecord
        n = self.lib.focal_trace_count()
        recs = (TraceRecord * max(n, 1))()
        if n:
            from focal_amd._lib import check
            check(self.lib.focal_trace_read(0, n, recs))
        out = [dict(kernel=short_kernel_name(recs[i].kernel.decode()), us=float(recs[i].us), wgs=recs[i].grid[0] * recs[i].grid[1] * recs[i].grid[2],
                    threads=recs[i].block[0] * recs[i].block[1] * recs[i].block[2], bytes=0.0, flops=0.0, bound="hbm", inst="(no op description)", rows=0)
               for i in range(n)]
        for desc, n0, n1 in self.calls:
            mine = out[n0:n1]
            parts = desc.get("parts")
            size = [r["wgs"] * r["threads"] for r in mine]
            for r, sz in zip(mine, size):
                share = 1.0 if len(mine) == 1 else sz / max(sum(size), 1)
                if parts:
                    hit = [v for k, v in parts.items() if k in r["kernel"]]
                    r["bytes"] = float(hit[0]) if hit else 0.0
                    r["flops"] = 0.0
                else:
                    r["bytes"], r["flops"] = desc["bytes"] * share, desc["flops"] * share
                r["bound"], r["inst"], r["rows"] = desc["bound"], desc["inst"], desc.get("rows", 0)
        return out


def _median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def trace_groups(per_step):
    """per_step: the launch lists of n traced steps (StepTracer.launches() cut per step).  Every launch position of the step is one
    instance; its duration is the MEDIAN over the steps (the steps launch the same sequence; when they do not -- never observed --
    the mean over all launches is used).  Returns the groups, one per kernel instantiation."""
    n = len(per_step)
    same = all(len(s) == len(per_step[0]) and all(a["kernel"] == b["kernel"] for a, b in zip(s, per_step[0])) for s in per_step)
    if same:
        inst = [dict(per_step[0][j], us=_median([s[j]["us"] for s in per_step])) for j in range(len(per_step[0]))]
        scale = 1.0
    else:
        inst = [r for s in per_step for r in s]
        scale = 1.0 / n
    groups = {}
    for r in inst:
        g = groups.setdefault(r["kernel"], dict(kernel=r["kernel"], bound=r["bound"], calls=0.0, us=0.0, bytes=0.0, flops=0.0, inst={}))
        g["calls"] += scale
        g["us"] += r["us"] * scale
        g["bytes"] += r["bytes"] * scale
        g["flops"] += r["flops"] * scale
        if r["bound"] == "mfma":
            g["bound"] = "mfma"
        i = g["inst"].setdefault((r["inst"], r["wgs"], r["threads"]), [0.0, 0.0, r["bytes"]])
        i[0] += scale
        i[1] += r["us"] * scale
    out = []
    for g in groups.values():
        inst_rows = sorted(({"instance": f"{k[0]} [{k[1]} workgroups x {k[2]}]", "calls_per_step": round(c, 2), "avg_us": round(t / c, 2),
                             "GBps": round(b / (t / c * 1e-6) / 1e9, 1)} for k, (c, t, b) in g["inst"].items()), key=lambda r: -r["calls_per_step"] * r["avg_us"])
        out.append(dict(kernel=g["kernel"], bound=g["bound"], calls_per_step=g["calls"], us_per_step=g["us"], avg_us=g["us"] / g["calls"],
                        bytes_per_launch=g["bytes"] / g["calls"], flops_per_launch=g["flops"] / g["calls"], instances=inst_rows[:12]))
    out.sort(key=lambda g: -g["us_per_step"])
    return out, same


def _reference(a):
    """The committed rocprofv3 figures of this workload (tools/profile_round.sh): per kernel instantiation calls per step, average
    duration and PMC fabric bytes per launch; the serialized kernel time and the PMC bytes of a whole step."""
    tf = os.path.join(ROOT, "profiles", f"{ROUND}_reference_{a.model}_{a.dataset}.json")
    if not os.path.exists(tf):
        return None, tf
    return json.load(open(tf)), tf


def run_traced_steps(step, n_steps, mode):
    """n_steps eager steps under an open launch trace -> (groups, launches per step, same_sequence).  One stream for the traced steps: the
    modality encoders normally run on their own streams, where a kernel's duration also holds the slow-down from the other streams'
    kernels (observed +20-45 %); rocprofv3 serialises the streams, and the committed trace this is checked against was taken that way.
    The step's `value` is measured with all streams, of course."""
    ops = step.ops
    lib = __import__("focal_amd._lib", fromlist=["load"]).load()
    from focal_amd._lib import check
    os.environ["FOCAL_NO_STREAMS"] = "1"
    tr = StepTracer(ops)
    try:
        # (only rank 0 gets here: its traced steps are local -- no collective that the other ranks are not in)
        with step.dist.local_only():
            for _ in range(2):  # untraced: every allocator pool of the one-stream eager form is warm before a record is taken
                step.run()
            torch.cuda.synchronize()
            tr.install()
            per_step = []
            for _ in range(n_steps):
                check(lib.focal_trace_begin(4096, mode))
                tr.calls = []
                step.run()
                torch.cuda.synchronize()
                lib.focal_trace_end()
                per_step.append(tr.launches())
    finally:
        lib.focal_trace_end()
        tr.remove()
        os.environ.pop("FOCAL_NO_STREAMS", None)
    groups, same = trace_groups(per_step)
    return groups, per_step, same


def roofline(a, step, device, ms_per_step=None):
    """`roofline` of the JSON line: the kernel (instantiation = row of `rocprofv3 --stats`) with the largest time per step, measured in the
    step.  HBM-bound kernels: achieved = algorithmic bytes per launch / average launch duration against the 8 TB/s peak; the fused MLP
    backward is compute-bound (its bytes are 6 B per token-channel): TFLOP/s against the dense bf16 MFMA peak.  `isolated`
    (SW_Transformer on MOD) repeats the round-1 measurement (one instance, back-to-back launches) warm and cold for comparison."""
    from focal_amd._lib import TRACE_DISPATCH, TRACE_EVENTS
    mode = TRACE_EVENTS if os.environ.get("FOCAL_BENCH_TRACE_MODE") == "events" else TRACE_DISPATCH
    n_steps = 5
    groups, per_step, same = run_traced_steps(step, n_steps, mode)
    g0 = groups[0]
    kern = g0["kernel"]
    bound = g0["bound"]
    if bound == "hbm":
        ach = g0["bytes_per_launch"] / (g0["avg_us"] * 1e-6) / 1e9
        peak, unit = HBM_PEAK_GBS, "GB/s"
    else:
        ach = g0["flops_per_launch"] / (g0["avg_us"] * 1e-6) / 1e12
        peak, unit = (MFMA_BF16_PEAK_TF if a.dtype == "bf16" else MFMA_F32_PEAK_TF), "TFLOP/s"
    kernel_ms = sum(g["us_per_step"] for g in groups) / 1e3
    ref, ref_file = _reference(a)
    ref_name = "profiles/" + os.path.basename(ref_file)
    traffic, traffic_src, check_, suspect, why = None, f"{ref_name} missing: no PMC pass committed for this workload", None, False, []
    if ref is not None:
        row = ref["kernels"].get(kern)
        prov = f"{ref_name} (rocprofv3 of the same workload at library build {ref.get('lib_sha16')}; this run's library {_lib_sha16()})"
        if row and row.get("hbm_bytes_per_launch") is not None:
            traffic, traffic_src = round(row["hbm_bytes_per_launch"]), prov + ": 2 x FETCH_SIZE + WRITE_SIZE per launch, launch-weighted over the kernel's launch shapes"
        else:
            traffic_src = f"{ref_name} has no PMC row for {kern}"
        if row:
            check_ = {"rocprof_avg_us": round(row["avg_us"], 2), "rocprof_calls_per_step": round(row["calls_per_step"], 2),
                      "avg_us_over_rocprof": round(g0["avg_us"] / row["avg_us"], 3), "rocprof_kernel_ms_per_step": round(ref["serialized_ms_per_step"], 3),
                      "kernel_ms_over_rocprof": round(kernel_ms / ref["serialized_ms_per_step"], 3), "source": prov}
            if abs(g0["avg_us"] / row["avg_us"] - 1.0) > 0.15:
                suspect = True
                why.append(f"avg_us {g0['avg_us']:.1f} vs rocprofv3 {row['avg_us']:.1f} for the same kernel: off by more than 15 %")
        else:
            suspect = True
            why.append(f"{ref_name} has no row for {kern}")
        if kernel_ms > 1.25 * ref["serialized_ms_per_step"]:
            suspect = True
            why.append(f"kernel_ms_per_step {kernel_ms:.2f} > 1.25 x the committed rocprofv3 serialized total {ref['serialized_ms_per_step']:.2f}")
    if not same:
        why.append("the traced steps did not launch identical sequences: means over all launches instead of per-instance medians")

    def grp(g):
        return {"kernel": g["kernel"], "calls_per_step": round(g["calls_per_step"], 2), "avg_us": round(g["avg_us"], 2),
                "ms_per_step": round(g["us_per_step"] / 1e3, 4), "GBps": round(g["bytes_per_launch"] / (g["avg_us"] * 1e-6) / 1e9, 1),
                "TFLOPps": round(g["flops_per_launch"] / (g["avg_us"] * 1e-6) / 1e12, 1),
                **({"rocprof_avg_us": round(ref["kernels"][g["kernel"]]["avg_us"], 2)} if ref and g["kernel"] in ref["kernels"] else {})}
    step_bytes = ref.get("hbm_bytes_per_step") if ref else None
    compulsory = _compulsory_bytes(a, step)
    out = {"bound": bound, "kernel": kern, "achieved": round(ach, 1), "peak": peak, "unit": unit,
           "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
           "measured": ("in the step: the library's launch trace (start / stop events on each dispatch of %d eager one-stream steps after 2 untraced ones; "
                        "per-launch-position median over the steps); algorithmic bytes (flops) of all the kernel's launches / their total time" % n_steps)
                       if mode == TRACE_DISPATCH else "in the step: hipEventRecord pairs around each launch inside the library (cross-check mode)",
           "suspect": suspect, "suspect_reason": "; ".join(why) or None, "check_against_profile": check_,
           "calls_per_step": round(g0["calls_per_step"], 2), "avg_us": round(g0["avg_us"], 2), "ms_per_step": round(g0["us_per_step"] / 1e3, 4),
           "launches_per_step": round(sum(g["calls_per_step"] for g in groups), 1),
           "kernel_ms_per_step": round(kernel_ms, 3),
           "algorithmic_bytes_per_launch": round(g0["bytes_per_launch"]), "flops_per_launch": round(g0["flops_per_launch"]),
           "instances": g0["instances"],
           "other_groups": [grp(g) for g in groups[1:13]],
           "families": _families(groups),
           "stages": _stage_table(a, step, per_step),
           "step": {"hbm_bytes_per_step_pmc": step_bytes,
                    "hbm_frac_step": (round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (step_bytes and ms_per_step) else None),
                    "compulsory_bytes_per_step": compulsory,
                    "traffic_over_compulsory": (round(step_bytes / compulsory, 1) if (step_bytes and compulsory) else None),
                    "note": "PMC bytes of one eager step at the L2 <-> fabric boundary (Infinity-Cache hits included) / the graph-replayed step time / 8 TB/s; "
                            "compulsory = the windows read once + 40 B per trained parameter (SURVEY 8d)"},
           "isolated": roofline_isolated(a, step, device) if (a.model == "SW_Transformer" and a.dataset == "MOD") else None}
    return out


def _stage_table(a, step, per_step):
    """SW_Transformer: the traced launches folded by Swin stage: a launch belongs to the stage whose token-row count
    (views x B x H x W of a modality's stage) its operands have; PatchMerging's reduction runs on the next stage's rows and counts there.
    launches / serialized ms / algorithmic MB per step, medians over the traced steps as everywhere in `roofline`."""
    geo = getattr(step.backbone, "geometry", None)
    if a.model != "SW_Transformer" or geo is None or not per_step:
        return None
    views = 2 if getattr(step.backbone, "views_share_pass", False) else 1
    owner = {}
    for loc, mods in geo.items():
        for mod, g in mods.items():
            for si, st in enumerate(g["stages"]):
                owner.setdefault(views * a.batch * st["H"] * st["W"], f"stage {si}")
    n = len(per_step)
    same = all(len(sx) == len(per_step[0]) for sx in per_step)
    rows = {}
    for j, r in enumerate(per_step[0]):
        us = _median([sx[j]["us"] for sx in per_step]) if same else r["us"]
        key = owner.get(r.get("rows", 0), "outside the encoder stages (DFT, embedding, mod_in, projector, loss head, AdamW)")
        e = rows.setdefault(key, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += us
        e[2] += r["bytes"]
    return [{"stage": k, "launches": v[0], "ms": round(v[1] / 1e3, 4), "algorithmic_MB": round(v[2] / 1e6, 1),
             "GBps": round(v[2] / (v[1] * 1e-6) / 1e9, 1) if v[1] > 0 else None} for k, v in sorted(rows.items())]


def _lib_sha16():
    import hashlib
    from focal_amd import _lib
    return hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()[:16]


def _compulsory_bytes(a, step):
    """every window read once per view (time domain, fp32) + 40 B per trained parameter (weights read three times,
    gradient, AdamW state)."""
    x = sum(t.numel() * 4 for mm in step.x.values() for t in mm.values())
    n_par = sum(p.numel() for p in step.model.parameters() if p.requires_grad)
    return int(2 * x + 40 * n_par)


def _families(groups):
    """The launch groups folded by what they compute (a family may span kernels and launch shapes: the weight gradients run on
    focal_dw_group_kernel, focal_dw_ring_kernel, focal_gemm_kernel<dW ..> at several grids and inside mlp_bwd_kernel)."""
    fam = {}
    for g in groups:
        k = g["kernel"]
        name = ("weight gradients (all dW launches)" if (k.startswith(("focal_dw_ring", "focal_dw_group", "focal_dw_tail_group")) or (k.startswith("focal_gemm_kernel<") and ", true, true," in k))
                else k.split("<")[0].split(" ")[0])
        f = fam.setdefault(name, [0.0, 0.0, 0.0, 0.0])
        f[0] += g["calls_per_step"]
        f[1] += g["us_per_step"]
        f[2] += g["bytes_per_launch"] * g["calls_per_step"]
        f[3] += g["flops_per_launch"] * g["calls_per_step"]
    out = [{"family": n, "calls_per_step": round(f[0], 2), "ms_per_step": round(f[1] / 1e3, 4), "GBps": round(f[2] / (f[1] * 1e-6) / 1e9, 1),
            "frac_of_hbm_peak": round(f[2] / (f[1] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "TFLOPps": round(f[3] / (f[1] * 1e-6) / 1e12, 1)}
           for n, f in fam.items()]
    out.sort(key=lambda e: -e["ms_per_step"])
    return out[:14]


def roofline_isolated(a, step, device):
    """Round 1's measurement, kept as a cross-check: the largest weight-gradient instance (stage-0 audio qkv: dW[192,64] +=
    dqkv[M,192]^T a1[M,64], M = 2 views x B x 576 tokens) launched back to back on one operand set (warm: it sits in the Infinity
    Cache) and rotating through > 512 MB of operand sets (cold: every launch reads HBM, as in the step)."""
    ops = step.ops
    ct = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    es = 2 if ct == torch.bfloat16 else 4
    geo = step.backbone.geometry[step.cfg["location_names"][0]][step.cfg["modality_names"][-1]]["stages"][0]
    views = 2 if getattr(step.backbone, "views_share_pass", False) else 1
    M, C = views * a.batch * geo["H"] * geo["W"], geo["C"]
    N, K = 3 * C, C
    d = ops.linear_desc(ops.code(ct), M, N, K, ops.code(ct), ops.code(ct))
    bytes_alg = M * N * es + M * K * es + N * K * 4
    nsets = max(2, (640 << 20) // (M * (N + K) * es) + 1)
    sets = [(torch.randn(M, N, device=device).to(ct), torch.randn(M, K, device=device).to(ct)) for _ in range(nsets)]
    dw, db = torch.zeros(N, K, device=device), torch.zeros(N, device=device)
    k = [0]

    def warm():
        ops.linear_bwd_weight(d, sets[0][0], sets[0][1], dw, db)

    def cold():
        g, x = sets[k[0] % nsets]
        k[0] += 1
        ops.linear_bwd_weight(d, g, x, dw, db)
    ms_w, ms_c = time_kernel(warm, a.roofline_iters), time_kernel(cold, max(a.roofline_iters, 2 * nsets))
    return {"kernel": "dW[%d,%d] += dy[%d,%d]^T x[%d,%d] (%s)" % (N, K, M, N, M, K, a.dtype), "algorithmic_bytes": bytes_alg,
            "warm_us": round(ms_w * 1e3, 2), "warm_frac": round(bytes_alg / (ms_w * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "cold_us": round(ms_c * 1e3, 2), "cold_frac": round(bytes_alg / (ms_c * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


def dump_trace(a, step):
    """tools/profile_round.sh: the launch trace in both modes next to the rocprofv3 pass of the same box (profiles/r6_z_*_trace_vs_rocprof.txt)."""
    from focal_amd._lib import TRACE_DISPATCH, TRACE_EVENTS
    for _ in range(2):
        step.run()
    torch.cuda.synchronize()
    out = {}
    for tag, mode in (("dispatch", TRACE_DISPATCH), ("events", TRACE_EVENTS)):
        groups, per_step, same = run_traced_steps(step, 5, mode)
        out[tag] = {"same_sequence": same, "kernel_ms_per_step": sum(g["us_per_step"] for g in groups) / 1e3,
                    "kernels": {g["kernel"]: {"calls_per_step": g["calls_per_step"], "avg_us": g["avg_us"], "bytes_per_launch": g["bytes_per_launch"],
                                              "flops_per_launch": g["flops_per_launch"], "bound": g["bound"]} for g in groups}}
        if tag == "dispatch" and same:  # every launch position of the step, in launch order (median over the steps): tools/step_instances.py reads this
            out["launches"] = [dict(kernel=r["kernel"], inst=r["inst"], wgs=r["wgs"], threads=r["threads"], bytes=r["bytes"], flops=r["flops"],
                                    us=_median([s[j]["us"] for s in per_step])) for j, r in enumerate(per_step[0])]
    json.dump(out, open(a.trace_dump, "w"), indent=1)


def cpu_baseline(a, cfg):
    """The oracle (CPU restatement of the reference step: "port") timed on this box's host cores, bounded sample, under the GPU
    leg's conditions: two DISTINCT views (x and -1.1 x), FFT + fwd x2 + loss + bwd + AdamW.  The reference loop itself cannot travel
    to the GPU box; profiles/cpu_equivalence.json (tools/probe_reference_cpu.py, build container) records how much slower it is
    than this port on the same cores -> `reference_ratio_probed`, `reference_equivalent_value`."""
    from oracle.step import OracleTrainer, fft_realpack
    from oracle.weights import seeded_values, swt_state_spec, deepsense_state_spec, synthetic_time_input
    task = "vehicle_classification" if a.dataset == "MOD" else "activity_classification"
    spec = swt_state_spec(cfg, task) if a.model == "SW_Transformer" else deepsense_state_spec(cfg, task)
    state = {}
    for k, shp in spec.items():
        if k.endswith(("relative_position_index", "num_batches_tracked")):
            state[k] = torch.zeros(shp, dtype=torch.long)
        elif k.endswith("attn_mask"):
            state[k] = torch.zeros(shp)
        else:
            state[k] = seeded_values(k, shp)
    # torch's CPU kernels stop scaling (and collapse under oversubscription) long before a 256-thread host is full:
    # use at most 32 threads and say so
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    B = 32
    tr = OracleTrainer(a.model, cfg, state)
    x = synthetic_time_input(cfg, B, 5)
    x2 = {l: {m: -1.1 * v for m, v in mm.items()} for l, mm in x.items()}
    one = lambda: tr.step(freq_pair=(fft_realpack(x), fft_realpack(x2)))
    t0 = time.time()
    one()  # warm-up
    warm = time.time() - t0
    t0 = time.time()
    n = 0
    while n < a.cpu_steps and (time.time() - t0) + warm < 25:
        one()
        n += 1
    dt = time.time() - t0
    if n == 0:
        n, dt = 1, warm
    out = {"value": round(B * n / dt, 2), "unit": "windows/s", "cores": cores, "kind": "port",
           "sample": f"{n} steps of batch {B} (fp32, two distinct views, FFT + fwd x2 + loss + bwd + AdamW, dropout off) after 1 warm-up"}
    eq = os.path.join(ROOT, "profiles", "cpu_equivalence.json")
    if os.path.exists(eq):
        e = json.load(open(eq))
        r = e["models"].get(a.model, {}).get("reference_over_oracle")
        if r:
            out["reference_ratio_probed"] = r
            out["reference_equivalent_value"] = round(out["value"] * r, 2)
            out["reference_ratio_note"] = (f"reference train loop / this port, both timed in the build container on {e['threads']} threads "
                                           "(profiles/cpu_equivalence.json); the reference's Python cannot travel to the GPU box")
    return out


_CLOCK_FILES = {}


def _clock_files(device):
    """sysfs nodes of THIS rank's GPU, found by its PCI address (a box may expose many cards; the visible device is not card0)."""
    import glob
    key = device.index or 0
    if key not in _CLOCK_FILES:
        pr = torch.cuda.get_device_properties(device)
        d = "/sys/bus/pci/devices/%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        first = lambda pat: (sorted(glob.glob(d + pat)) or [None])[0]
        _CLOCK_FILES[key] = {"sclk": first("/hwmon/hwmon*/freq1_input"), "power": first("/hwmon/hwmon*/power1_input") or first("/hwmon/hwmon*/power1_average"),
                             "cap": first("/hwmon/hwmon*/power1_cap"), "mclk": d + "/pp_dpm_mclk" if os.path.exists(d + "/pp_dpm_mclk") else None}
    return _CLOCK_FILES[key]


def gpu_clocks(device):
    """sclk (MHz), mclk level, socket power (W) and power cap of this rank's GPU from sysfs (a few tens of microseconds: the paths are
    resolved once).  Read right before the timed region and right after it (FOCAL_BENCH_MID_CLOCKS=1: also once in its middle, after the
    middle step's loss.item() -- the only sample that sees the GPU under load): tells a power- or clock-limited box from a fast one in the
    driver's record.  None where sysfs is not readable (the failed lookup is remembered: it is not repeated)."""
    key = device.index or 0
    if _CLOCK_FILES.get(key) is False:
        return None
    try:
        f = _clock_files(device)
        rd = lambda path, div: (int(open(path).read().strip()) // div) if path else None
        mclk = [ln.split(":")[1].strip().rstrip("*").strip() for ln in open(f["mclk"]) if "*" in ln] if f["mclk"] else []
        return {"sclk_mhz": rd(f["sclk"], 1000000), "mclk": mclk[0] if mclk else None, "power_w": rd(f["power"], 1000000), "power_cap_w": rd(f["cap"], 1000000)}
    except Exception:  # noqa: BLE001  (a diagnostic must not cost the line)
        _CLOCK_FILES[key] = False
        return None


def timed_steps(a, step, world, rank, device, steps, warmup):
    """Capture (unless --no-graph), `warmup` untimed steps, then exactly `steps` steps between barrier + synchronize on both sides.
    Returns (seconds of this rank, whether graphs were replayed, the stream everything ran on)."""
    graphed = False
    run = step.run
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(2):
            step.run()  # builds the arena / moments / workspaces outside capture
        torch.cuda.synchronize()
        if not a.no_graph:
            # capture failures are not rank-dependent: StepSegments.capture makes the ranks agree after every segment, before the
            # collective that follows it, so either every rank gets its graphs or every rank raises here
            try:
                run, graphed = step.capture(side), True
            except Exception as e:  # noqa: BLE001
                print(f"[bench] rank {rank}: hipGraph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                torch.cuda.synchronize()
        if step.host_draws:
            drawn = run

            def run():
                step.draw_views()
                drawn()
        if a.from_host:
            step.enable_host_feed()
            inner = run

            def run():
                step.feed_step_inputs()
                inner()
        # Every step ends with loss.item(), so host stamps after it are the steps' durations: the JSON line carries the whole series
        # (`step_series`), warm-up included.  After the `warmup` requested steps the UNTIMED phase goes on until three consecutive steps
        # agree to 1 % AND the untimed phase as a whole has lasted 0.2 s (cap: 0.5 s of settling): the first replays of a cold process run
        # 3-10 % slow while clocks / power ramp (profiles/r6_warmup_transient.txt), the next ~30 another ~1 % (same box, --steps 20: warm-up 5
        # -> 51 137 / 51 340, warm-up 30 -> 51 668 / 51 820, warm-up 60 -> 51 691; step-to-step noise is +-1 %, so three agreeing steps alone do
        # not see that tail), and a 0.1 s timed region must not start inside either.  The timed region below is untouched: exactly `steps` full
        # steps.  `warmup_effective` = the untimed steps actually run.  FOCAL_BENCH_MIN_UNTIMED_S overrides the 0.2 s.
        series = step.series = {"warmup_ms": [], "settle_ms": [], "timed_ms": []}
        min_untimed = float(os.environ.get("FOCAL_BENCH_MIN_UNTIMED_S", "0.2"))
        tp = t_untimed = time.perf_counter()
        for _ in range(warmup):
            run()
            step.loss.item()
            tn = time.perf_counter()
            series["warmup_ms"].append(round((tn - tp) * 1e3, 3))
            tp = tn
        t_settle = tp
        recent = series["warmup_ms"][-3:]
        while os.environ.get("FOCAL_BENCH_NO_SETTLE") != "1" and warmup > 0:
            agree = len(recent) == 3 and max(recent) <= 1.01 * min(recent)
            more = (not agree or (tp - t_untimed) < min_untimed) and (tp - t_settle) < 0.5
            if world > 1:  # the ranks must leave this loop together (the step holds collectives)
                f = torch.tensor([1 if more else 0], device=device, dtype=torch.int32)
                dist.all_reduce(f, op=dist.ReduceOp.MAX)
                more = bool(f.item())
                tp = time.perf_counter()
            if not more:
                break
            run()
            step.loss.item()
            tn = time.perf_counter()
            series["settle_ms"].append(round((tn - tp) * 1e3, 3))
            recent = (recent + [series["settle_ms"][-1]])[-3:]
            tp = tn
        series["warmup_effective"] = warmup + len(series["settle_ms"])
        series["clocks_before_timed"] = gpu_clocks(device)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        gpu_clocks(device)  # (resolves the sysfs paths outside the timed region)
        t0 = time.perf_counter()
        lag = os.environ.get("FOCAL_BENCH_LAGGED_LOSS") == "1"  # diagnostic: read step k-1's loss after launching step k
        prev = None
        stamps = [t0]
        mid_clocks = None
        for k_step in range(steps):
            run()
            if lag:
                if prev is not None:
                    prev.item()
                prev = step.loss.clone()
            else:
                step.loss.item()  # the reference syncs on loss.item() every step (pretrain.py:74)
            stamps.append(time.perf_counter())
            if k_step == steps // 2 and os.environ.get("FOCAL_BENCH_MID_CLOCKS") == "1":
                # opt-in: one sysfs sample under load.  It was the default until the round-6 evidence run showed what it costs where sysfs does
                # NOT answer: the failed lookup (device properties + glob, retried) stretched that one step by 0.5-0.6 ms -- 0.6 % of a 20-step
                # timed region (step_series.timed_ms of profiles/r6_z_bench_driver_line.json: 5.53 ms at index 11, 4.9-5.1 elsewhere)
                mid_clocks = gpu_clocks(device)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        series["clocks_mid_timed"] = mid_clocks
        series["clocks_after_timed"] = gpu_clocks(device)
        series["timed_ms"] = [round((b - a_) * 1e3, 3) for a_, b in zip(stamps[:-1], stamps[1:])]
        step.last_run = run
    return dt, graphed, side


def dp_diagnostics(a, step, world, device, graphed):
    """N > 1: what the first RCCL runs are read with (every rank calls this; rank 0 reports): the backend and its version, the number of
    ranks that really take part (an all-reduce of ones), and the medians of HIP events between the pieces of 5 more replayed steps --
    the graph segments, the embedding all-gather, the gradient all-reduce and how much of it stays exposed behind the split backward."""
    ones = torch.ones(1, device=device)
    dist.all_reduce(ones)
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None
    except Exception as e:  # noqa: BLE001  (a version query must not cost the line)
        ver = f"unavailable ({type(e).__name__})"
    out = {"backend": dist.get_backend(), "world": world, "ranks_seen": int(ones.item()), "rccl_version": ver,
           "loss_head_sharded": bool(step.dist.shard_loss_head()), "split_backward": step.seg.buckets() is not None}
    if graphed:
        # (collective: every rank replays these 5 steps; a failure here is a failure of the step itself and is not swallowed)
        pieces = step.seg.measure_pieces(step.last_run, 5)
        out["us_segments"] = pieces
        out["us_exchange"] = round(sum(v for k, v in pieces.items() if k.startswith("exchange")), 1)
        out["us_allreduce_exposed"] = round(sum(v for k, v in pieces.items() if "exposed" in k), 1)
        out["us_step_from_events"] = round(sum(pieces.values()), 1)
    return out


def secondary_workloads(a, device):
    """BASELINE configs[1] (DeepSense, MOD) and the per-GPU workload of configs[4] (SW_Transformer, 4-modality HAR4) timed by the same
    process right after the headline, the same way (captured step, 10 warm-up + 20 timed steps, loss read every step): so that the
    driver's own run carries them.  The headline keys of the JSON line are untouched."""
    out = {}
    for model, dataset in (("DeepSense", "MOD"), ("SW_Transformer", "HAR4")):
        if (model, dataset) == (a.model, a.dataset):
            continue
        b = copy.copy(a)
        b.model, b.dataset, b.views, b.from_host = model, dataset, "fixed", False
        try:
            st = Step(b, device)
            dt, graphed, _ = timed_steps(b, st, 1, 0, device, 20, 10)
            out[f"{model}/{dataset}"] = {"value": round(b.batch * 20 / dt, 1), "unit": "windows/s", "ms_per_step": round(dt / 20 * 1e3, 3), "steps": 20, "warmup": 10,
                                         "hip_graph": graphed, "dtype": b.dtype, "batch": b.batch, "last_loss": round(st.loss.item(), 4)}
            if model == "DeepSense" and not a.no_cpu_baseline:
                # BASELINE configs[0] is DeepSense at batch 32 on the CPU: the oracle step on this box's host cores, next to the GPU line
                out[f"{model}/{dataset}"]["cpu_baseline"] = cpu_baseline(b, st.cfg)
            del st
        except Exception as e:  # noqa: BLE001  (a secondary workload must never take the headline line down with it)
            out[f"{model}/{dataset}"] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return out


def main():
    a = parse()
    test_backend = os.environ.get("FOCAL_BENCH_TEST_BACKEND")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # `python bench.py --gpus N` without a launcher: N fresh child ranks (focal_amd/launch.py, what `train.py -gpu=0,..,N-1` uses),
        # spawned BEFORE any HIP call in this process (device_count() does not initialise the GPU); never a silent 1-GPU run
        from focal_amd.launch import spawn_ranks
        have = torch.cuda.device_count()
        if have < a.gpus and not test_backend:
            sys.exit(f"bench.py --gpus {a.gpus}: {have} GPU(s) visible.  Launch line on an {a.gpus}-GPU node: python -m torch.distributed.run --nnodes=1 "
                     f"--nproc-per-node {a.gpus} --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus {a.gpus} --steps {a.steps} --warmup {a.warmup}")
        sys.exit(spawn_ranks(__file__, sys.argv[1:], list(range(a.gpus)), narrow_visible=not test_backend))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit(f"bench.py --gpus {a.gpus} launched with WORLD_SIZE={world}: the two must agree (the JSON line's n_gpus is the number of ranks that ran)")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (tests/ and single-GPU boxes only): FOCAL_BENCH_TEST_BACKEND=gloo runs every rank on cuda:0 over gloo so
    # that the N>1 control flow can be exercised on a 1-GPU box; the driver's multi-GPU runs use RCCL, one GPU per rank.
    if test_backend:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        if test_backend:
            dist.init_process_group(test_backend)
        else:
            dist.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(minutes=30))
    rank = dist.get_rank() if world > 1 else 0
    if os.environ.get("FOCAL_ABLATE"):  # timing diagnostic (tools/ablate_shim.py): the JSON line is marked invalid below
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import ablate_shim
        ablate_shim.install()
    step = Step(a, device)
    if a.roofline_only:
        print(json.dumps(roofline(a, step, device)))
        return
    if a.trace_dump:
        dump_trace(a, step)
        return

    dt, graphed, side = timed_steps(a, step, world, rank, device, a.steps, a.warmup)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    last_loss = step.loss.item()
    with torch.cuda.stream(side):
        dp = dp_diagnostics(a, step, world, device, graphed) if world > 1 else None
    with torch.cuda.stream(side):  # the stream the warm-up and the timed steps ran on: its allocator pools are the warm ones
        rl = roofline(a, step, device, dt / a.steps * 1e3) if (rank == 0 and not a.no_roofline) else None
    # (the plain single-GPU run only: profiling / diagnostic invocations -- --no-graph, --no-roofline, --no-cpu-baseline -- measure one workload)
    plain = not (a.no_secondary or a.no_graph or a.no_roofline or a.no_cpu_baseline or a.no_dropout or a.from_host or a.views != "fixed" or a.batch != 256)
    sec = secondary_workloads(a, device) if (rank == 0 and world == 1 and plain and a.model == "SW_Transformer" and a.dataset == "MOD") else None
    cb = cpu_baseline(a, step.cfg) if (rank == 0 and world == 1 and not a.no_cpu_baseline) else None
    if world > 1:
        dist.barrier()  # the other ranks wait here while rank 0 measures its roofline: all leave the process group together
    if rank == 0:
        wps = a.batch * world * a.steps / dt
        out = {"metric": "pretrain windows/sec (whole node), FOCAL " + a.model, "value": round(wps, 1), "unit": "windows/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "warmup_effective": step.series.get("warmup_effective"), "ms_per_step": round(dt / a.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": f"{a.model} + FOCAL pretrain step, {a.batch} {a.dataset}-shaped {len(step.cfg['modality_names'])}-modality windows/GPU "
                                      f"(global batch {a.batch * world}), train mode, DFT + fwd x2 + loss + bwd + AdamW",
                          "global_batch": a.batch * world, "parallelism": f"dp{world}", "hip_graph": graphed, **({"DIAGNOSTIC_no_dropout": True} if a.no_dropout else {}),
                          **({"DIAGNOSTIC_from_host": True} if a.from_host else {}),
**({"DIAGNOSTIC_ABLATED_INVALID": os.environ["FOCAL_ABLATE"]} if os.environ.get("FOCAL_ABLATE") else {}),
                          "views": ("identity / negation+scaling (x * -1.1) folded into the DFT" if a.views == "fixed" else
                                    "random: the product Augmenter's views, every draw made on the device inside the captured step (train.py's form; not the quoted configuration)"
                                    if not step.host_draws else
                                    "random-host: the product Augmenter's draws on the host, made eagerly before every replay inside the timed region (DIAGNOSTIC: not the quoted configuration)"),
                          "last_loss": round(last_loss, 4),
                          "parity": "this configuration (train mode, dropout / DropPath on) is covered by statistical and finite-difference tests; "
                                    "the same kernels with dropout off are pinned bit-for-tolerance against reference fixtures (tests/golden, DESIGN 1): "
                                    "bf16 embeddings <= 1e-2 of scale (observed <= 0.87e-2, tests/golden/OBSERVED_r6.json) and loss terms within 1e-2 max(1, |term|) "
                                    "(the B = 8 ranking term: 0.97e-2).  NOT part of the bf16 claim: DeepSense in eval mode on the seeded (deliberately mismatched) "
                                    "running statistics of DeepSense_b8.npz (2.5e-2 on the audio embedding; that fixture pins the fp32 eval path)"},
               "model_flops_frac_of_bf16_mfma_peak": (round(wps / world * flops_per_window(a.model, a.dataset) / (MFMA_BF16_PEAK_TF * 1e12), 5)
                                                       if flops_per_window(a.model, a.dataset) else None),
               "flops_per_window": flops_per_window(a.model, a.dataset),
               "step_series": step.series, "roofline": rl, "cpu_baseline": cb, "secondary": sec, "dp": dp}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
