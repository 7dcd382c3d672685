#!/usr/bin/env python3
"""Idle gaps between consecutive kernels of one queue in a rocprofv3 kernel trace of the bench's timed loop (graph replay,
64 windows, three sub-batch streams).  usage: stream_gaps.py p_kernel_trace.csv"""
import collections, csv, sys
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith(("ba::", "void ba::"))]
byq = collections.defaultdict(list)
for r in rows:
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ba::", "")
    gx = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]); gy = int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"])
    win = gx if "solve" in nm else gy
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm, win))
for q, v in byq.items():
    v.sort()
    v = [x for x in v if x[3] in (21, 22)]          # the timed loop's launches only
    if len(v) < 100:
        continue
    gaps = collections.defaultdict(list)
    for a, b in zip(v[:-1], v[1:]):
        gaps[(a[2][:9], b[2][:9])].append((b[0] - a[1]) / 1e3)
    dur = collections.defaultdict(list)
    for x in v:
        dur[x[2][:9]].append((x[1] - x[0]) / 1e3)
    it = [b[0] - a[0] for a, b in zip(v[:-3], v[3:]) if a[2] == b[2] and (b[0] - a[0]) < 4e5]
    print("queue", q, "launches", len(v), "start-to-start of the same kernel one iteration later: median %.1f us" % (np.median(it) / 1e3))
    for k, g in sorted(gaps.items()):
        g = np.array(g)
        g = g[g < 400]                                   # (pauses between the timed regions are not gaps)
        print("   gap %-10s -> %-10s p25 %5.1f  median %5.1f  p75 %5.1f  p95 %5.1f  mean %5.1f us  n %d" % (k[0], k[1], *np.percentile(g, [25, 50, 75, 95]), g.mean(), len(g)))
    for k, d in sorted(dur.items()):
        print("   duration %-10s median %6.1f us mean %6.1f" % (k, np.median(d), np.mean(d)))
