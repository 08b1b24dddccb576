#!/usr/bin/env python3
"""Two-stream timeline of an eager pretraining step: HIP events around every libfocal_hip launch, recorded on the stream the launch
goes to, all measured against one base event -- where each stream is busy, where it waits, what runs beside what.  (rocprofv3
serialises the streams, a hipGraph replay cannot be instrumented; the GPU is kept behind the host by a queue of large fills so the
event pairs bracket kernel time, not host launch gaps.)

  python tools/timeline.py [bench args, e.g. --model DeepSense | --dataset HAR4]      -> gpurun_out/timeline_<model>_<dataset>.txt
"""
import collections
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "focal_amd", "src"))
import torch  # noqa: E402

import bench  # noqa: E402
from focal_amd import ops  # noqa: E402

SKIP = {"code", "torch_dtype", "zero_pool_reset", "pool_zeros", "zeros", "drop_desc", "new_rng_state", "linear_desc", "ln_desc", "mlp_desc",
        "attn_desc", "conv_desc", "conv_in_desc", "bn_desc", "mlp_supported", "dw_group_supported", "dw_group_kind", "resid_ln_supported", "bwd_data_ln_supported", "check",
        "linear_bwd_weight_group_workgroups", "linear"}
rec = []


def wrap(name, fn):
    def traced(*a, **kw):
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        out = fn(*a, **kw)
        e1.record(st)
        shp = "x".join(str(tuple(t.shape)) for t in a if isinstance(t, torch.Tensor))[:60]
        rec.append((name, st.cuda_stream, e0, e1, shp))
        return out
    return traced


def main():
    a = bench.parse()
    dev = torch.device("cuda", 0)
    step = bench.Step(a, dev)
    for _ in range(3):
        step.run()
    torch.cuda.synchronize()
    saved = {}
    for name in dir(ops):
        fn = getattr(ops, name)
        if callable(fn) and not name.startswith("_") and name not in SKIP and getattr(fn, "__module__", "") == ops.__name__ and not isinstance(fn, type):
            saved[name] = fn
            setattr(ops, name, wrap(name, fn))
    pad = torch.empty(256 << 20, dtype=torch.float32, device=dev)
    for _ in range(int(os.environ.get("TIMELINE_PAD_FILLS", "100"))):  # ~0.2 ms each: the whole step is enqueued before the GPU gets to it
        pad.fill_(1.0)
    base = torch.cuda.Event(enable_timing=True)
    base.record(torch.cuda.current_stream())
    step.run()
    end = torch.cuda.Event(enable_timing=True)
    end.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    for k, v in saved.items():
        setattr(ops, k, v)
    rows = sorted(((base.elapsed_time(e0) * 1e3, base.elapsed_time(e1) * 1e3, n, s, shp) for n, s, e0, e1, shp in rec), key=lambda r: r[0])
    total = base.elapsed_time(end) * 1e3
    streams = sorted({r[3] for r in rows})
    sid = {s: i for i, s in enumerate(streams)}
    out = [f"# eager step with per-launch events: {len(rows)} traced launches on {len(streams)} streams, {total:.0f} us from the first enqueue to the end",
           "# per stream: busy = sum of its launches' durations, span = first start .. last end"]
    for s in streams:
        rs = [r for r in rows if r[3] == s]
        busy = sum(r[1] - r[0] for r in rs)
        out.append(f"#   stream {sid[s]}: {len(rs):4d} launches, busy {busy:7.0f} us, span {rs[0][0]:7.0f} .. {max(r[1] for r in rs):7.0f} us")
    # overlap: time covered by >= 2 streams, by exactly 1, by none
    ev = []
    for t0, t1, *_ in rows:
        ev.append((t0, 1)); ev.append((t1, -1))
    ev.sort()
    cover = collections.Counter()
    depth, last = 0, 0.0
    for t, d in ev:
        cover[min(depth, 2)] += t - last
        depth += d
        last = t
    out.append(f"#   time with no traced kernel {cover[0]:.0f} us, exactly one stream busy {cover[1]:.0f} us, two or more {cover[2]:.0f} us")
    fam = collections.defaultdict(lambda: [0, 0.0])
    for t0, t1, n, s, shp in rows:
        fam[n][0] += 1
        fam[n][1] += t1 - t0
    out.append("# by op (sum of durations as seen with the other stream running beside it):")
    for n, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        out.append(f"#   {n:28s} {c:4d} launches {t:8.0f} us")
    out.append("# start_us  end_us  dur_us  stream  op  shapes")
    for t0, t1, n, s, shp in rows:
        out.append(f"{t0:9.1f} {t1:9.1f} {t1 - t0:7.1f}  s{sid[s]}  {n}  {shp}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", f"timeline_{a.model}_{a.dataset}.txt")
    open(path, "w").write("\n".join(out) + "\n")
    print("\n".join(out[:60]))


if __name__ == "__main__":
    main()
