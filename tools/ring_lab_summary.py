#!/usr/bin/env python3
"""Condense the output of tools/gemm_ring_lab.hip: per case the pipe kernel's time, the best ring configuration and all variants."""
import re
import sys

pat = re.compile(r'(.*?)\s+([\d.]+) \(([\d.]+) TB/s\)')
for line in open(sys.argv[1]):
    line = line.rstrip()
    if '!!' in line:
        print(line)
        continue
    m = re.match(r'(.{0,64}?)\s+([\d.]+) MB :(.*)', line)
    if not m:
        print(line[:200])
        continue
    parts = [pat.match(x.strip()) for x in m.group(3).split('|') if x.strip()]
    parts = [(q.group(1), float(q.group(2))) for q in parts if q]
    base = parts[0][1]
    rest = parts[1:]
    best = min(rest, key=lambda t: t[1]) if rest else ("", 0.0)
    mb = float(m.group(2))
    print(f"{m.group(1).strip():44s} {mb:6.1f}MB old {base:6.1f} ({mb / base:4.2f} TB/s) best {best[1]:6.1f} x{best[1] / base:.2f} {best[0]:20s} | " +
          " | ".join(f"{t} {u:.1f}" for t, u in rest))
