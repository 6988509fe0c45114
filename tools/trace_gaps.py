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
periods, durs, others, idles = [], [], [], []
for a, b in zip(idx, idx[1:]):
    s0, e0, _ = rows[a]
    s1, _, _ = rows[b]
    between = rows[a + 1:b]
    if any("fx_" not in n for _, _, n in between):   # (something else of the process ran in between: not a back-to-back step)
        continue
    busy = sum(e - s for s, e, _ in between)
    periods.append(s1 - s0)
    durs.append(e0 - s0)
    others.append(busy)
    idles.append((s1 - s0) - (e0 - s0) - busy)
    last_between = [n[:60] for _, _, n in between]
def med(x):
    return statistics.median(x) / 1e3 if x else float("nan")
n = len(periods)
half = periods[n // 2:], durs[n // 2:], others[n // 2:], idles[n // 2:]   # the later half: settled clocks
print("back-to-back pairs: %d; later half (us, medians): period %.2f = dominant kernel %.2f + other fx kernels %.2f + idle %.2f" % (
    n, med(half[0]), med(half[1]), med(half[2]), med(half[3])))
print("between two launches:", last_between if n else None)
