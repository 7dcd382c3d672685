#!/usr/bin/env python3
"""The bench line's `dogleg` record under a kernel trace: one warm-up batch and one traced batch of 64 fresh windows, optimize(10) in
the reference's mode from the perturbed start.

  run:    rocprofv3 --kernel-trace --output-format csv -d DIR -o p -- python scripts/r04_dogleg_trace.py run
  digest: python scripts/r04_dogleg_trace.py digest DIR/.../p_kernel_trace.csv  -> the kernels of the last optimize call in start
          order: start (us after the first), duration, queue, workgroups, name; then per queue the busy time and the gaps
"""
import csv
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import numpy as np
    from okvis_amd import solver, synthetic
    from okvis_amd.window import default_options
    n = int(os.environ.get("DOGLEG_WINDOWS", "64"))
    bd = None
    for bi in range(3):
        fresh = [synthetic.make_window(10, 400, 1.0, 7_000_000 + 1000 * bi + i) for i in range(n)]
        if bd is None:
            bd = solver.WindowBatch(fresh, device=0, options=default_options())
        else:
            bd.upload(fresh)
        bd.synchronize()
        time.sleep(0.05)   # a visible gap in the trace in front of every call
        t0 = time.perf_counter()
        sm = bd.optimize(10)
        bd.synchronize()
        dt = time.perf_counter() - t0
        redo = int(sum(bd.array("IMU_REDO_COUNT", w).sum() for w in range(n)))
        print(f"batch {bi}: optimize(10) {dt * 1e3:.3f} ms, {sum(x['iterations'] for x in sm)} counted iterations, "
              f"{int(bd.array('SLOTS')[0])} slots, {redo} re-preintegrations", flush=True)
    bd.close()


def digest(path):
    rows = []
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"),
                     (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) * (int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"])), name))
    rows.sort()
    # the last call: everything after the last gap of more than 20 ms
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - rows[i - 1][1] > 20_000_000:
            cut = i
    rows = rows[cut:]
    t0 = rows[0][0]
    print(f"{len(rows)} kernels, {(max(r[1] for r in rows) - t0) / 1e3:.1f} us from the first start to the last end")
    for s, e, q, wg, name in rows:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q:>3} {wg:6d}  {name[:70]}")
    by_q = {}
    for s, e, q, wg, name in rows:
        by_q.setdefault(q, []).append((s, e))
    for q, v in sorted(by_q.items()):
        busy = sum(e - s for s, e in v)
        span = v[-1][1] - v[0][0]
        print(f"queue {q}: {len(v)} kernels, busy {busy / 1e3:.1f} us of {span / 1e3:.1f} us")
    by_name = {}
    for s, e, q, wg, name in rows:
        by_name.setdefault((name, wg), []).append(e - s)
    for (name, wg), v in sorted(by_name.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        print(f"{sum(v) / 1e3:9.1f} us  {len(v):4d} x  median {v[len(v) // 2] / 1e3:7.1f}  max {v[-1] / 1e3:7.1f}  wg {wg:6d}  {name[:60]}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else digest(sys.argv[2])
