#!/usr/bin/env python3
"""Per HIP queue: median duration, gap to the next kernel and launch period of one kernel's dispatches in a rocprofv3
--kernel-trace run.   python tools/kernel_gaps.py <trace dir> <kernel substring>"""
import collections
import csv
import glob
import statistics
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"]]
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for q, v in sorted(byq.items()):
    v.sort()
    v = v[len(v) // 4:]  # steady part
    dur = [b - a for a, b in v]
    gap = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    per = [v[i + 1][0] - v[i][0] for i in range(len(v) - 1)]
    print("queue %s: %d dispatches, duration median %.2f us (min %.2f), gap median %.2f us, period median %.2f us"
          % (q, len(v), statistics.median(dur) / 1e3, min(dur) / 1e3, statistics.median(gap) / 1e3, statistics.median(per) / 1e3))
