#!/usr/bin/env python3
"""Split a rocprofv3 kernel trace (p_kernel_trace.csv) by launch shape.

bench.py launches the same kernels in three shapes (1 window: `single_window`; windows/2 per stream: the timed
loop; all windows: `okvis_ba_profile_iterations`, the eager per-kernel pass behind `roofline`), so the averages of
`--stats` mix them.  This prints one row per (kernel, windows in the launch) so that the `roofline.avg_launch_us`
of the bench line can be compared with the profiler's average for the same launch shape.

usage: kernel_trace_by_shape.py p_kernel_trace.csv > by_shape.csv
"""
import collections
import csv
import sys

rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not name.startswith("ba::"):
        continue
    gx = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    gy = int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"])
    windows = gy if gy > 1 or "solve" not in name else gx       # solve: one workgroup per window in x
    if "solve" in name:
        windows = gx
    rows[(name, windows)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "windows_in_launch", "calls", "avg_us", "median_us", "min_us", "max_us"])
for (name, windows), v in sorted(rows.items()):
    v.sort()
    w.writerow([name, windows, len(v), "%.2f" % (sum(v) / len(v) / 1e3), "%.2f" % (v[len(v) // 2] / 1e3),
                "%.2f" % (v[0] / 1e3), "%.2f" % (v[-1] / 1e3)])
