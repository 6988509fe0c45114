#!/usr/bin/env python3
"""Timeline of the fx_* kernels in a rocprofv3 kernel trace: per launch of the dominant kernel its duration, what ran between it and the next
one (names, durations) and the idle gaps -- where a step's time goes beyond its dominant kernel.   python tools/trace_gaps.py <kernel_trace.csv>"""
import csv
import statistics
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r.get("Kernel_Name", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
fx = [(s, e, n) for s, e, n in rows if "fx_" in n]
tot = {}
for s, e, n in fx:
    tot[n] = tot.get(n, 0) + (e - s)
dom = max(tot, key=tot.get)
print("dominant:", dom[:100])
idx = [i for i, (s, e, n) in enumerate(rows) if n == dom]
groups = {}   # names between two launches of the dominant kernel -> list of (period, dominant duration, other fx busy time, idle)
for a, b in zip(idx, idx[1:]):
    s0, e0, _ = rows[a]
    s1, _, _ = rows[b]
    between = rows[a + 1:b]
    if any("fx_" not in n for _, _, n in between):   # (something else of the process ran in between: not a back-to-back step)
        continue
    busy = sum(e - s for s, e, _ in between)
    key = tuple(n.split("(")[0][:70] for _, _, n in between)
    groups.setdefault(key, []).append((s1 - s0, e0 - s0, busy, (s1 - s0) - (e0 - s0) - busy))
for key, v in groups.items():
    v = v[len(v) // 2:]   # the later half: settled clocks
    med = lambda i: statistics.median(x[i] for x in v) / 1e3
    print("%d pairs with %s between (later half, us, medians): period %.2f = dominant kernel %.2f + other fx kernels %.2f + idle %.2f" % (
        len(v), list(key) if key else "nothing", med(0), med(1), med(2), med(3)))
