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

FLOP_PER_WINDOW = {"SW_Transformer": 4.376e9, "DeepSense": 0.809e9}  # SURVEY 8d (fwd+bwd, both views)
HBM_PEAK_GBS = 8000.0
MFMA_BF16_PEAK_TF = 2500.0
MFMA_F32_PEAK_TF = 157.3


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=30)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--model", default="SW_Transformer")
    p.add_argument("--batch", type=int, default=256, help="windows per GPU")
    p.add_argument("--dtype", default="bf16")
    p.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--sync-bn", action="store_true", help="DeepSense under DP: cross-rank BatchNorm statistics (exact global-batch parity)")
    p.add_argument("--no-dropout", action="store_true", help="diagnostic: all dropout rates 0 (flagged in the JSON line)")
    p.add_argument("--cpu-steps", type=int, default=4)
    p.add_argument("--roofline-only", action="store_true", help="only run the dominant-kernel loop (for rocprofv3 --pmc passes)")
    p.add_argument("--roofline-iters", type=int, default=20)
    p.add_argument("--from-host", action="store_true", help="diagnostic (never the reported `value`): every step's windows start in pinned "
                   "host memory, as the reference's DataLoader hands them over; H2D on a copy stream one step ahead")
    p.add_argument("--no-roofline", action="store_true", help="profiling runs: skip the dominant-kernel timing loop (the JSON line then has roofline null)")
    return p.parse_args()


def make_args(cfg, model, device, dtype):
    return argparse.Namespace(model=model, dataset="MOD", device=device, train_mode="contrastive", learn_framework="FOCAL",
                              stage="pretrain", task="vehicle_classification", tag=None, dataset_config=cfg,
                              compute_dtype=dtype)


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
        cfg = load_yaml(os.path.join(ROOT, "focal_amd", "src", "data", "MOD.yaml"))
        if a.no_dropout:  # diagnostic only (measures what the mask generation costs); never the reported configuration
            for k in ("dropout_ratio", "drop_path_rate", "attn_drop_rate"):
                cfg["SW_Transformer"][k] = 0.0
            cfg["DeepSense"]["dropout_ratio"] = 0.0
        self.cfg = cfg
        args = make_args(cfg, a.model, device, a.dtype)
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
        self.loss = torch.zeros((), device=device)
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

    def views(self):
        # both views of a modality are written into the halves of one [2B, ...] tensor (what Augmenter.forward_random does for
        # back-to-back draws): SW_Transformer runs them as one batch without a concatenation
        both = {l: {m: torch.empty(2 * x.shape[0], 2 * x.shape[1], x.shape[2], x.shape[3], device=x.device) for m, x in mm.items()}
                for l, mm in self.x.items()}
        v1 = {l: {m: self.ops.fft_realpack(x, out=both[l][m][:x.shape[0]]) for m, x in mm.items()} for l, mm in self.x.items()}
        v2 = {l: {m: self.ops.fft_realpack(x, scale=-1.1, out=both[l][m][x.shape[0]:]) for m, x in mm.items()}
              for l, mm in self.x.items()}  # negation + scaling, folded into the DFT
        return v1, v2

    # The step in three capturable segments with the two data-parallel collectives between them:
    #   A: zero_grad, views (DFT), both backbone passes, pack the embeddings     -> [all-gather embeddings]
    #   B: loss head on the global batch, backward                                -> [all-reduce gradient arena]
    #   C: AdamW, loss value
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
        loss.backward()
        self.loss.copy_(loss.detach())
        self.feats = None

    def reduce(self):
        self.opt.reduce_gradients()

    def seg_c(self):
        self.opt.step(reduce=False)

    def run(self):
        self.seg_a()
        self.exchange()
        self.seg_b()
        self.reduce()
        self.seg_c()

    def capture(self, stream):
        """Returns the replay callable: one hipGraph of the whole step on one rank; with N > 1 ranks, hipGraphs of the three
        segments (one shared memory pool) with the two collectives issued eagerly between their replays.

        Everything is captured TWICE into the same graph-private pool and the second set is the one replayed: the first capture
        grows the pool segment by segment, the second sub-allocates the same tensors from the segments that now exist, and
        that placement replays 2.5 % faster (8.29 -> 8.08 ms, reproducible; the first set is kept alive so its blocks stay put)."""
        self.opt.sync_lr()
        multi = self.dist.is_dist()
        # with a process group alive its watchdog thread polls events (cudaEventQuery) at any time: under the default "global"
        # capture mode that would invalidate a capture in progress, "thread_local" restricts the checks to the capturing thread
        mode = {"capture_error_mode": "thread_local"} if multi else {}
        pool, self._warm_graphs = None, []
        for attempt in range(1 if os.environ.get("FOCAL_BENCH_SINGLE_CAPTURE") == "1" else 2):
            if not multi:
                # one rank: no collectives to interleave -> one graph for the whole step (each extra graph launch costs
                # ~0.1 ms of idle GPU per step)
                whole = torch.cuda.CUDAGraph()
                with torch.cuda.graph(whole, pool=pool, stream=stream):
                    self.run()
                graphs = (whole,)
            else:
                ga, gb, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga, pool=pool, stream=stream, **mode):
                    self.seg_a()
                self.exchange()  # eager, autograd-aware: links segment B's backward to segment A's forward
                with torch.cuda.graph(gb, pool=ga.pool(), stream=stream, **mode):
                    self.seg_b()
                self.reduce()
                with torch.cuda.graph(gc, pool=ga.pool(), stream=stream, **mode):
                    self.seg_c()
                graphs = (ga, gb, gc)
            pool = graphs[0].pool()
            if attempt == 0:
                self._warm_graphs = graphs
        if not multi:
            return graphs[0].replay
        ga, gb, gc = graphs
        packed = self.packed

        def replay():
            ga.replay()
            self.dist.replay_exchange(packed)
            gb.replay()
            self.opt.reduce_gradients()
            gc.replay()
        return replay


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


def roofline(a, step, device):
    """Live measurement of the dominant kernel (largest total time in the committed rocprofv3 summary, profiles/): the
    weight-gradient GEMM template focal_gemm_kernel<bf16: A = dy (transposed), B = activation (transposed), fp32 atomic
    out, 64x64 tiles, split over tokens>, timed on its largest instance exactly as the step launches it: dW of the stage-0
    audio MLP down-projection, dW[64,256] += gm[M,64]^T h[M,256], M = 2 views x B x 576 tokens (gm = the residual-stream
    gradient, already multiplied by the branch's dropout mask by the LayerNorm backward that produced it).
    HBM-bound (AI = 2*64*256 / ((64 + 256)*2) = 51 flop/B): algorithmic bytes = read gm + read h + write dW once."""
    ops = step.ops
    if a.model != "SW_Transformer":
        return roofline_deepsense(a, step, device)
    ct = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    es = 2 if ct == torch.bfloat16 else 4
    geo = step.backbone.geometry["shake"]["audio"]["stages"][0]
    views = 2 if getattr(step.backbone, "views_share_pass", False) else 1
    M, C = views * a.batch * geo["H"] * geo["W"], geo["C"]
    N, K = C, 4 * C  # the forward linear is [M, K=4C] -> [M, N=C]
    from focal_amd._lib import ACT_GELU
    g = torch.randn(M, N, device=device).to(ct)
    h = torch.randn(M, K, device=device).to(ct)
    dw = torch.zeros(N, K, device=device)
    db = torch.zeros(N, device=device)
    d = ops.linear_desc(ops.code(ct), M, N, K, ops.code(ct), ops.code(ct), ACT_GELU)
    ms = time_kernel(lambda: ops.linear_bwd_weight(d, g, h, dw, db), iters=a.roofline_iters)
    bytes_alg = M * N * es + M * K * es + N * K * 4
    flops = 2.0 * M * N * K
    gbs = bytes_alg / (ms * 1e-3) / 1e9
    traffic = None
    tf = os.path.join(ROOT, "profiles", "r1_u_pmc_roofline_kernel.json")
    if os.path.exists(tf):  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of `bench.py --roofline-only` (see the file's note)
        traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
    return {"bound": "hbm", "kernel": "focal_gemm_kernel<dW: gm[%d,%d]%s^T x h[%d,%d]%s -> fp32 atomics, 64x64 tiles>" % (M, N, a.dtype, M, K, a.dtype),
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": traffic, "algorithmic_bytes": bytes_alg, "ms_per_launch": round(ms, 5),
            "tflops": round(flops / (ms * 1e-3) / 1e12, 1)}


def roofline_deepsense(a, step, device):
    """DeepSense's dominant kernel family: the [1,5] inter-conv as a sliding-window MFMA GEMM
    (M = B*10*20 tokens, K = 5*64, N = 64): reads the bf16 activation once, writes the fp32 pre-BN output."""
    ops = step.ops
    ct = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    es = 2 if ct == torch.bfloat16 else 4
    C, S, k = 64, 20, 5
    rows = a.batch * 10 * S
    x = torch.randn(rows, C, device=device).to(ct)
    w = (torch.randn(C, k * C, device=device) * (k * C) ** -0.5).to(ct)
    b = torch.zeros(C, device=device)
    d = ops.conv_desc(ops.code(ct), rows, S, C, C, k)
    ms = time_kernel(lambda: ops.conv_fwd(d, x, w, b))
    bytes_alg = rows * C * es + rows * C * 4 + C * k * C * es
    gbs = bytes_alg / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "focal_gemm_kernel<conv window> rows=%d K=%d N=%d" % (rows, k * C, C), "achieved": round(gbs, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
            "ms_per_launch": round(ms, 5), "tflops": round(2.0 * rows * C * k * C / (ms * 1e-3) / 1e12, 1)}


def cpu_baseline(a, cfg):
    """The oracle (CPU restatement of the reference step: "port") timed on this box's host cores, bounded sample."""
    from oracle.step import OracleTrainer
    from oracle.weights import seeded_values, swt_state_spec, deepsense_state_spec, synthetic_time_input
    spec = swt_state_spec(cfg) if a.model == "SW_Transformer" else deepsense_state_spec(cfg)
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
    t0 = time.time()
    tr.step(time_x=x)  # warm-up
    warm = time.time() - t0
    t0 = time.time()
    n = 0
    while n < a.cpu_steps and (time.time() - t0) + warm < 25:
        tr.step(time_x=x)
        n += 1
    dt = time.time() - t0
    if n == 0:
        n, dt = 1, warm
    return {"value": round(B * n / dt, 2), "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of batch {B} (fp32, FFT+fwd x2+loss+bwd+AdamW, dropout off) after 1 warm-up"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (tests/ and single-GPU boxes only): FOCAL_BENCH_TEST_BACKEND=gloo runs every rank on cuda:0 over gloo so
    # that the N>1 control flow can be exercised on a 1-GPU box; the driver's multi-GPU runs use RCCL, one GPU per rank.
    test_backend = os.environ.get("FOCAL_BENCH_TEST_BACKEND")
    if test_backend:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        if test_backend:
            dist.init_process_group(test_backend)
        else:
            dist.init_process_group("nccl", device_id=device)
    rank = dist.get_rank() if world > 1 else 0
    step = Step(a, device)
    if a.roofline_only:
        print(json.dumps(roofline(a, step, device)))
        return

    graphed = False
    run = step.run
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(2):
            step.run()  # builds the arena / moments / workspaces outside capture
        torch.cuda.synchronize()
        if not a.no_graph:
            # capture failures must not be rank-dependent (a rank falling back to eager while another replays would
            # desynchronise the collectives): agree on the outcome before using the graphs
            ok = 1
            try:
                replay = step.capture(side)
            except Exception as e:  # noqa: BLE001
                ok = 0
                print(f"[bench] rank {rank}: hipGraph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                torch.cuda.synchronize()
            if world > 1:
                flag = torch.tensor([ok], device=device, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            if ok:
                run, graphed = replay, True
        if a.from_host:
            step.enable_host_feed()
            inner = run

            def run():
                step.feed_step_inputs()
                inner()
        for _ in range(a.warmup):
            run()
            step.loss.item()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lag = os.environ.get("FOCAL_BENCH_LAGGED_LOSS") == "1"  # diagnostic: read step k-1's loss after launching step k
        prev = None
        for _ in range(a.steps):
            run()
            if lag:
                if prev is not None:
                    prev.item()
                prev = step.loss.clone()
            else:
                step.loss.item()  # the reference syncs on loss.item() every step (pretrain.py:74)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    last_loss = step.loss.item()
    rl = roofline(a, step, device) if (rank == 0 and not a.no_roofline) else None
    cb = cpu_baseline(a, step.cfg) if (rank == 0 and world == 1 and not a.no_cpu_baseline) else None
    if rank == 0:
        wps = a.batch * world * a.steps / dt
        out = {"metric": "pretrain windows/sec (whole node), FOCAL " + a.model, "value": round(wps, 1), "unit": "windows/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": f"{a.model} + FOCAL pretrain step, {a.batch} MOD-shaped 2-modality windows/GPU "
                                      f"(global batch {a.batch * world}), train mode, DFT + fwd x2 + loss + bwd + AdamW",
                          "global_batch": a.batch * world, "parallelism": f"dp{world}", "hip_graph": graphed, **({"DIAGNOSTIC_no_dropout": True} if a.no_dropout else {}),
                          **({"DIAGNOSTIC_from_host": True} if a.from_host else {}),
**({"DIAGNOSTIC_ABLATED_INVALID": os.environ["FOCAL_ABLATE"]} if os.environ.get("FOCAL_ABLATE") else {}),
                          "views": "identity / negation+scaling (x * -1.1) folded into the DFT", "last_loss": round(last_loss, 4)},
               "model_flops_frac_of_bf16_mfma_peak": round(wps / world * FLOP_PER_WINDOW[a.model] / (MFMA_BF16_PEAK_TF * 1e12), 5),
               "roofline": rl, "cpu_baseline": cb}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
