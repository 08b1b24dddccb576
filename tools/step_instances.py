#!/usr/bin/env python3
"""Per-launch table of a step from `bench.py --trace-dump FILE`: every launch position with its operand description, duration (median over the
traced steps), algorithmic GB/s and TFLOP/s -- grouped by (kernel, instance), sorted by time per step.
  python tools/step_instances.py FILE [kernel-substring]"""
import collections
import json
import sys


def main():
    d = json.load(open(sys.argv[1]))
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = collections.OrderedDict()
    for r in d["launches"]:
        if pat and pat not in r["kernel"]:
            continue
        k = (r["kernel"], r["inst"], r["wgs"], r["threads"])
        e = rows.setdefault(k, [0, 0.0, r["bytes"], r["flops"]])
        e[0] += 1
        e[1] += r["us"]
    tot = sum(e[1] for e in rows.values())
    print(f"{len(d['launches'])} launches per step; {tot / 1e3:.3f} ms in the {len(rows)} instances shown")
    print(f"{'us/step':>8s} {'calls':>5s} {'avg us':>7s} {'GB/s':>7s} {'TF/s':>6s} {'wgs':>5s}  kernel | instance")
    for (kern, inst, wgs, thr), (n, us, b, f) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        avg = us / n
        print(f"{us:8.1f} {n:5d} {avg:7.1f} {b / avg / 1e3:7.0f} {f / avg / 1e6:6.0f} {wgs:5d}  {kern[:62]} | {inst}")


if __name__ == "__main__":
    main()
