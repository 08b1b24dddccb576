#!/usr/bin/env python3
"""AMD_LOG_LEVEL=3 log of a graph capture -> per-stream sequence of captured operations (kernel launches collapsed), to see which
stream's last captured node is not followed by an event record that somebody waits for."""
import re
import sys
ops = []
for l in open(sys.argv[1], errors="ignore"):
    m = re.search(r"\[hipGraph\] Current capture node (\w+) on stream : (0x[0-9a-f]+)(?:, Event (0x[0-9a-f]+))?", l)
    if m:
        ops.append((m.group(1), m.group(2), m.group(3)))
    elif "hipStreamBeginCapture" in l and "Returned" not in l:
        ops.append(("BEGIN", "", None))
    elif "hipStreamEndCapture (" in l:
        ops.append(("END", "", None))
# take the last capture
last_begin = max(i for i, o in enumerate(ops) if o[0] == "BEGIN")
ops = ops[last_begin + 1:]
streams = {}
order = []
pending_event = {}
for kind, st, ev in ops:
    if kind == "END":
        break
    if st not in streams:
        streams[st] = []
        order.append(st)
    seq = streams[st]
    if kind == "LaunchKernel":
        if seq and seq[-1][0] == "K":
            seq[-1] = ("K", seq[-1][1] + 1)
        else:
            seq.append(("K", 1))
    elif kind == "EventRecord":
        pending_event[ev] = st
        seq.append(("REC", ev[-5:]))
    elif kind == "StreamWaitEvent":
        seq.append(("WAIT<-" + str(order.index(pending_event[ev]) if ev in pending_event else "ext"), ev[-5:]))
    else:
        seq.append((kind, ""))
for i, st in enumerate(order):
    seq = streams[st]
    print(f"stream {i} {st}: {len(seq)} entries; tail:", " ".join(f"{a}:{b}" for a, b in seq[-8:]))
