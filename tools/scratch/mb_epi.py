#!/usr/bin/env python3
"""Per-kernel microbenchmark at the shapes of the SW_Transformer step (B=256 windows, two views in one pass) (HIP-event timed, GPU box only).
Prints one line per (op, shape): microseconds per launch, algorithmic GB/s and TFLOP/s."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
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

B = 512
c, f32 = ops.code(CT), ops.code(torch.float32)
for tag, M, C in (("s1a", B * 144, 128), ("s2a", B * 36, 256), ("s2s", B * 18, 256)):
    a = rnd(M, C, dtype=CT)
    N = 4 * C
    w, b = rnd(N, C, dtype=CT), rnd(N)
    y, y2 = torch.empty(M, N, dtype=CT, device=DEV), torch.empty(M, N, dtype=CT, device=DEV)
    d0 = ops.linear_desc(c, M, N, C, c, c)
    d1 = ops.linear_desc(c, M, N, C, c, c, ACT_NONE, EPI_GELU)
    report(f"{tag} fc1 plain store [{M}x{C}]->{N}", timeit(lambda: ops.linear_fwd(d0, a, w, b, None, y)), (M * C + M * N) * ES, 2.0 * M * N * C)
    report(f"{tag} fc1 + gelu (2 outputs)", timeit(lambda: ops.linear_fwd(d1, a, w, b, None, y, y2)), (M * C + 2 * M * N) * ES, 2.0 * M * N * C)
    # dX fc2: gm [M, C] bf16 (pre-masked) x W2 [C, 4C] -> du [M, 4C] (x saved gelu')
    gm = rnd(M, C, dtype=CT)
    w2 = rnd(C, N, dtype=CT)
    du = torch.empty(M, N, dtype=CT, device=DEV)
    e0 = ops.linear_desc(c, M, C, N, c, c)
    e1 = ops.linear_desc(c, M, C, N, c, c, ACT_GELU)
    report(f"{tag} dX fc2 plain g[{M}x{C}]->{N}", timeit(lambda: ops.linear_bwd_data(e0, gm, w2, None, du)), (M * C + M * N) * ES, 2.0 * M * N * C)
    report(f"{tag} dX fc2 x gelu' (aux read)", timeit(lambda: ops.linear_bwd_data(e1, gm, w2, y2, du)), (M * C + 2 * M * N) * ES, 2.0 * M * N * C)
