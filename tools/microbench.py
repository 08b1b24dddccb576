#!/usr/bin/env python3
"""Per-kernel microbenchmark at the shapes of the SW_Transformer step (B=256 windows, two views in one pass) (HIP-event timed, GPU box only).
Prints one line per (op, shape): microseconds per launch, algorithmic GB/s and TFLOP/s."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from focal_amd import ops  # noqa: E402
from focal_amd._lib import ACT_GELU, ACT_NONE, EPI_GELU, EPI_NONE, EPI_RESIDUAL  # noqa: E402

DEV = "cuda"
CT = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
ES = 2 if CT == torch.bfloat16 else 4
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""


COLD = os.environ.get("FOCAL_MB_COLD") == "1"
_FLUSH = None


def timeit_cold(fn, iters=12):
    """Every timed launch starts with nothing cache-resident: a 768 MB fill (3 x the 256 MB infinity cache) runs before it, and
    only the launch itself sits between the two events.  This is how kernels meet their operands inside the training step
    (warm back-to-back replays flatter the numbers by 15-40 %)."""
    global _FLUSH
    if _FLUSH is None:
        _FLUSH = torch.empty(768 << 18, dtype=torch.float32, device=DEV)  # 768 MB
    for _ in range(2):
        fn()
    tot = 0.0
    for i in range(iters):
        _FLUSH.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e3  # us (includes ~2 us of eager launch + event overhead)


def timeit(fn, iters=20):
    """Time `fn` by replaying a captured hipGraph of `iters` launches (no host launch overhead in the number)."""
    if COLD:
        return timeit_cold(fn)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(3):
            g.replay()
        e1.record(st)
        e1.synchronize()
    return e0.elapsed_time(e1) / (3 * iters) * 1e3  # us


def report(name, us, nbytes, flops=0):
    print(f"{name:58s} {us:8.1f} us  {nbytes / us / 1e3:8.1f} GB/s  {flops / us / 1e6:8.1f} TF/s", flush=True)


def rnd(*s, dtype=torch.float32):
    return torch.randn(*s, device=DEV).to(dtype)


# (tag, tokens M, channels C) of the six (stage, modality) encoders; B = samples per backbone pass (both views = 512)
B = int(os.environ.get("FOCAL_MB_B", "512"))
STAGES = [("s0a", B * 576, 64), ("s0s", B * 288, 64), ("s1a", B * 144, 128), ("s1s", B * 72, 128), ("s2a", B * 36, 256), ("s2s", B * 18, 256)]
c, f32 = ops.code(CT), ops.code(torch.float32)

for tag, M, C in STAGES:
    if ONLY and ONLY not in ("gemm", tag):
        continue
    a = rnd(M, C, dtype=CT)
    x32 = rnd(M, C)
    g32 = rnd(M, C)
    for nm, N in (("qkv", 3 * C), ("fc1", 4 * C)):
        w, b = rnd(N, C, dtype=CT), rnd(N)
        y = torch.empty(M, N, dtype=CT, device=DEV)
        if nm == "qkv":
            d = ops.linear_desc(c, M, N, C, c, c)
            report(f"{tag} fwd {nm} [{M}x{C}]->{N}", timeit(lambda: ops.linear_fwd(d, a, w, b, None, y)), (M * C + M * N) * ES, 2.0 * M * N * C)
        else:
            y2 = torch.empty_like(y)
            d = ops.linear_desc(c, M, N, C, c, c, ACT_NONE, EPI_GELU)
            report(f"{tag} fwd {nm}+gelu [{M}x{C}]->{N} (2 outputs)", timeit(lambda: ops.linear_fwd(d, a, w, b, None, y, y2)), (M * C + 2 * M * N) * ES, 2.0 * M * N * C)
        dx = torch.empty(M, C, dtype=CT, device=DEV)
        dpl = ops.linear_desc(c, M, N, C, c, c)
        report(f"{tag} dX  {nm} dy[{M}x{N}]->{C}", timeit(lambda: ops.linear_bwd_data(dpl, y, w, None, dx)), (M * C + M * N) * ES, 2.0 * M * N * C)
        dw, db = torch.zeros(N, C, device=DEV), torch.zeros(N, device=DEV)
        report(f"{tag} dW  {nm} dy[{M}x{N}]^T x[{M}x{C}]", timeit(lambda: ops.linear_bwd_weight(dpl, y, a, dw, db)), (M * C + M * N) * ES, 2.0 * M * N * C)
    # proj / fc2 with residual epilogue
    for nm, K in (("proj", C), ("fc2", 4 * C)):
        xin = rnd(M, K, dtype=CT)
        w, b = rnd(C, K, dtype=CT), rnd(C)
        out = torch.empty(M, C, device=DEV)
        d = ops.linear_desc(c, M, C, K, c, f32, ACT_GELU if nm == "fc2" else ACT_NONE, EPI_RESIDUAL)
        report(f"{tag} fwd {nm}+res [{M}x{K}]->{C}", timeit(lambda: ops.linear_fwd(d, xin, w, b, x32, out)), M * K * ES + 2 * M * C * 4, 2.0 * M * C * K)
        dxin = torch.empty(M, K, dtype=CT, device=DEV)
        aux = rnd(M, K, dtype=CT) if nm == "fc2" else None
        report(f"{tag} dX  {nm} g[{M}x{C}]f32->{K}", timeit(lambda: ops.linear_bwd_data(d, g32, w, aux, dxin)), M * C * 4 + M * K * ES * (2 if aux is not None else 1), 2.0 * M * C * K)
        dw, db = torch.zeros(C, K, device=DEV), torch.zeros(C, device=DEV)
        report(f"{tag} dW  {nm} g[{M}x{C}]^T x[{M}x{K}]", timeit(lambda: ops.linear_bwd_weight(d, g32, xin, dw, db)), M * C * 4 + M * K * ES, 2.0 * M * C * K)

for tag, M, C in STAGES:
    if ONLY and ONLY not in ("ln", tag):
        continue
    x, gam, bet = rnd(M, C), rnd(C), rnd(C)
    y, st = ops.layernorm_fwd(x, gam, bet, CT)
    report(f"{tag} ln_fwd [{M}x{C}]", timeit(lambda: ops.layernorm_fwd(x, gam, bet, CT)), M * C * (4 + ES))
    dy, dx = rnd(M, C, dtype=CT), rnd(M, C)
    dg, dbt = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    report(f"{tag} ln_bwd [{M}x{C}]", timeit(lambda: ops.layernorm_bwd(dy, x, st, gam, dx, True, dg, dbt)), M * C * (ES + 4 + 8))

if not ONLY or ONLY == "lnmerge":  # PatchMerging LayerNorm backward (gather: rows of 4 C_in channels from the 2 x 2 token neighbourhood)
    for tag, Bq, H, W, Cin in (("m01a", B, 12, 48, 64), ("m12a", B, 6, 24, 128), ("m01s", B, 12, 24, 64), ("m12s", B, 6, 12, 128)):
        rows, C4 = Bq * (H // 2) * (W // 2), 4 * Cin
        x = rnd(Bq * H * W, Cin)
        gam, bet = rnd(C4), rnd(C4)
        y, st = ops.layernorm_fwd(x, gam, bet, CT, gather=(Bq, H, W, Cin))
        dy, dx = rnd(rows, C4, dtype=CT), torch.empty_like(x)
        dxm = torch.empty(x.shape, dtype=CT, device=DEV)
        dg, dbt = torch.zeros(C4, device=DEV), torch.zeros(C4, device=DEV)
        report(f"{tag} ln_bwd gather [{rows}x{C4}]", timeit(lambda: ops.layernorm_bwd(dy, x, st, gam, dx, False, dg, dbt, gather=(Bq, H, W, Cin), dx_masked=dxm)),
               rows * C4 * (ES + 4 + 4 + ES))

GEO = [("s0a", 12, 48, 64), ("s0s", 12, 24, 64), ("s1a", 6, 24, 128), ("s1s", 6, 12, 128), ("s2a", 3, 12, 256), ("s2s", 3, 6, 256)]
for tag, H, W, C in GEO:
    if ONLY and ONLY not in ("attn", tag):
        continue
    M = B * H * W
    qkv, table = rnd(M, 3 * C, dtype=CT), rnd(25, 4)
    o = torch.empty(M, C, dtype=CT, device=DEV)
    for sh in (0, 1):
        d = ops.attn_desc(c, B, H, W, C, 4, 3, 3, sh, sh)
        report(f"{tag} attn_fwd shift={sh} [{M}x{3 * C}]", timeit(lambda: ops.window_attn_fwd(d, qkv, table, o)), M * 4 * C * ES, 4.0 * M * 9 * C)
        do, dqkv, dt = rnd(M, C, dtype=CT), torch.empty_like(qkv), torch.zeros_like(table)
        report(f"{tag} attn_bwd shift={sh}", timeit(lambda: ops.window_attn_bwd(d, qkv, table, do, dqkv, dt)), M * 8 * C * ES, 10.0 * M * 9 * C)

if not ONLY or ONLY == "fft":
    for nm, shape in (("audio", (256, 1, 10, 1600)), ("seismic", (256, 1, 10, 20))):
        xt = rnd(*shape)
        n_el = xt.numel()
        report(f"fft_realpack {nm} {list(shape)}", timeit(lambda: ops.fft_realpack(xt)), n_el * 4 * 3)

if ONLY == "gru":
    Bq, T, H = int(os.environ.get("FOCAL_MB_GRU_B", "256")), int(os.environ.get("FOCAL_MB_GRU_T", "10")), 256
    gd = ops.GRUDesc(Bq, T, H)
    gi = [rnd(Bq * T, 3 * H) for _ in range(2)]
    w16 = [rnd(3 * H, H, dtype=torch.bfloat16) * 0.06 for _ in range(2)]
    bh = [rnd(3 * H) * 0.1 for _ in range(2)]
    hs = [torch.zeros(T + 1, Bq, H, device=DEV) for _ in range(2)]
    sv = [torch.empty(T, 4, Bq, H, device=DEV) for _ in range(2)]
    out = torch.empty(Bq, T, 2 * H, device=DEV)
    report("gru_seq_fwd B=256 T=10 H=256 x2 dirs", timeit(lambda: ops.gru_seq_fwd(gd, gi, w16, bh, hs, sv, out)), 0)
    wt = [rnd(H, 3 * H, dtype=torch.bfloat16) * 0.06 for _ in range(2)]
    dout = rnd(Bq * T, 2 * H)
    dgi = [torch.empty(Bq * T, 3 * H, device=DEV) for _ in range(2)]
    dgh = [torch.empty(T, Bq, 3 * H, device=DEV) for _ in range(2)]
    report("gru_seq_bwd B=256 T=10 H=256 x2 dirs", timeit(lambda: ops.gru_seq_bwd(gd, dout, T * 2 * H, 2 * H, 1.0, wt, hs, sv, dgi, dgh)), 0)
