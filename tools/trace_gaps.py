#!/usr/bin/env python3
"""Where one iteration's wall time goes, from a rocprofv3 --kernel-trace CSV: per kernel of ONE steady-state iteration (the
window between the last two rollout launches) its duration and the idle gap in front of it on the device, overlapping streams
merged.  usage: trace_gaps.py <*_kernel_trace.csv> [rollout-kernel-substring]"""
import csv
import sys

path = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "rollout"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if key in r[2]]
if len(starts) < 3:
    sys.exit(f"fewer than 3 launches matching {key!r}")
# step-wise rollouts launch the key kernel T times per iteration: iteration boundary = a gap of other kernels between two runs
bounds = [starts[0]] + [b for a, b in zip(starts, starts[1:]) if any(key not in rows[i][2] for i in range(a + 1, b)) and
                        sum(1 for i in range(a + 1, b) if key not in rows[i][2]) > 8]
lo, hi = bounds[-2], bounds[-1]
it = rows[lo:hi]
t0, t1 = it[0][0], rows[hi][0]
busy_end = t0
busy = 0
agg = {}
print(f"iteration window: {(t1 - t0) / 1e3:.1f} us, {len(it)} kernel launches")
for s, e, name in it:
    gap = max(0, s - busy_end)
    short = name.split("(")[0].replace("void ", "")[:48]
    a = agg.setdefault(short, [0, 0, 0])
    a[0] += 1
    a[1] += e - s
    a[2] += gap
    if e > busy_end:
        busy += e - max(s, busy_end)
        busy_end = e
print(f"device busy {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us ({100 * (t1 - t0 - busy) / (t1 - t0):.1f} %)")
print(f"{'kernel':50s} {'calls':>6s} {'total us':>10s} {'avg us':>8s} {'gap before, total us':>22s}")
for k, (n, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:50s} {n:6d} {d / 1e3:10.1f} {d / 1e3 / n:8.2f} {g / 1e3:22.1f}")
