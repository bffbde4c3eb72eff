#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 kernel_stats.csv (and the bench line's headline numbers)."""
import csv
import json
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms %.1f" % (tot / 1e6))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 12]:
    print("%-60s calls %7s tot %9.2f ms avg %8.1f us %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                               float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
if len(sys.argv) > 2:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("value %.4g %s, %.2f ms/step, rollout %.2f ms" % (d["value"], d["unit"], d["ms_per_step"], d["rollout_ms"]))
