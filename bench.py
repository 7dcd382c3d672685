#!/usr/bin/env python3
"""bench.py — Gauss-Newton iterations/s of the MI355X sliding-window BA backend.

A "step" is one full trust-region iteration (Schur reduce -> reduced solve -> back-substitute + (+)update
+ re-linearise, BASELINE.md §"Path under measurement") on every window resident on this GPU.  Workload
per GPU: `--windows` (default 64) independent synthetic windows of BASELINE.json configs[1] (10 keyframes /
2 cams / 400 landmarks / 100-sample IMU factors, fp64); `--windows 8` is configs[3]'s per-GPU share (64
windows over 8 GPUs) and `single_window` in the output is configs[1] alone (latency).
Weak scaling: every rank owns its own windows (seeds 20240923 + rank*windows + i), there is no data-path
collective; the only exchange is the gather of per-rank timings.  All convergence tolerances are disabled
in the timed region so that every step performs the full work (accepted or rejected steps launch the
same kernels).

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector peak (MI355X_MICROARCH.md); the linearise kernel issues plain v_fma_f64


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--windows", type=int, default=64, help="independent windows per GPU (weak scaling)")
    p.add_argument("--total-windows", type=int, default=0,
                   help="BASELINE configs[3]: this many windows IN TOTAL, window i on rank i mod N (strong scaling); "
                        "overrides --windows")
    p.add_argument("--repeats", type=int, default=15, help="timed regions of --steps iterations each (median / p10 / p90)")
    p.add_argument("--streams", type=int, default=0, help="sub-batch streams (0 = auto)")
    p.add_argument("--keyframes", type=int, default=10)
    p.add_argument("--landmarks", type=int, default=400)
    p.add_argument("--visibility", type=float, default=1.0)
    p.add_argument("--no-graph", action="store_true", help="eager launches (for rocprofv3 kernel traces)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-iters", type=int, default=0, help="oracle iterations for the CPU baseline (0 = auto ~12 s)")
    p.add_argument("--profile-steps", type=int, default=20, help="eager per-kernel HIP-event pass for the roofline")
    p.add_argument("--fp32", action="store_true", help="BASELINE configs[4]: fp32 Jacobian/Hessian build, fp64 solve (dtype f32+f64)")
    p.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 FETCH_SIZE/WRITE_SIZE passes (roofline.traffic = null)")
    p.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    return p.parse_args()


KERNEL_SYMBOL = {"schur": "schur_kernel", "solve": "solve_kernel", "linearize": "linearize_kernel"}


def pmc_traffic(a, kernel):
    """HBM bytes per launch of `kernel` from the TCC counters, collected the way MI355X_MICROARCH.md
    prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (they do not fit one pass) over
    an eager re-run of the same workload in a child process; FETCH_SIZE (KB) is doubled (gfx950 tallies
    128-B requests at 64 B), WRITE_SIZE (KB) is taken as reported (uncalibrated).  Returns None when
    rocprofv3 is unavailable or a pass fails — never fatal for the bench line."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    raw = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="okvis_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", "--windows", str(a.windows),
               "--keyframes", str(a.keyframes), "--landmarks", str(a.landmarks), "--visibility", str(a.visibility),
               "--steps", "12", "--warmup", "4", "--no-graph"] + (["--fp32"] if a.fp32 else [])
        try:
            subprocess.run(cmd, timeout=120, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"},
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == ctr and KERNEL_SYMBOL[kernel] in r["Kernel_Name"]:
                        vals.append(float(r["Counter_Value"]))
            if not vals:
                return None
            vals = vals[len(vals) // 4:]          # skip the first launches (first-touch / cold L2)
            raw[ctr] = sum(vals) / len(vals)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch_b = 2.0 * raw["FETCH_SIZE"] * 1024.0
    write_b = raw["WRITE_SIZE"] * 1024.0
    return {"bytes": fetch_b + write_b, "fetch_bytes": fetch_b, "write_bytes": write_b,
            "raw_kb": raw, "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE x1 (uncalibrated); separate --pmc passes, eager launches"}


def main():
    a = parse()
    from okvis_amd import dist as D
    rk = D.Rank.from_env()
    rank, world, local_rank = rk.rank, rk.world, D.local_device(rk)
    dist = D.init()   # "nccl" (= RCCL over xGMI) on the GPU node

    from okvis_amd import solver, synthetic
    from okvis_amd.window import default_options

    if a.total_windows > 0:     # strong scaling: a fixed set of windows sharded over the ranks (configs[3])
        ids = D.shard_windows(a.total_windows, rank, world)
        seeds = [20240923 + i for i in ids]
        a.windows = len(seeds)
    else:                       # weak scaling: every rank owns --windows windows of its own
        seeds = D.shard_seeds(rank, world, a.windows)
        ids = [rank * a.windows + i for i in range(a.windows)]
    wins = [synthetic.make_window(a.keyframes, a.landmarks, a.visibility, s) for s in seeds]
    opt = default_options()
    opt.function_tolerance = 0.0
    opt.gradient_tolerance = 0.0
    opt.parameter_tolerance = 0.0
    opt.use_graph = 0 if a.no_graph else 1
    opt.n_streams = a.streams
    opt.fp32_linearize = 1 if a.fp32 else 0
    opt.gauss_newton = 1  # every timed iteration does identical full work (no trust-region collapse at the optimum)
    batch = solver.WindowBatch(wins, device=local_rank, options=opt)

    def barrier():
        batch.synchronize()
        if dist is not None:
            import torch
            dist.barrier()
            if torch.cuda.is_available():
                torch.cuda.synchronize()

    batch.begin()
    if a.pmc_child:   # counter-collection child of pmc_traffic(): a short eager loop, no output
        batch.iterate(a.warmup + a.steps)
        batch.finish()
        batch.close()
        return
    if a.warmup > 0:
        batch.iterate(a.warmup)
    # build the graph of the timed call outside the timed region
    if not a.no_graph and a.steps != a.warmup:
        batch.iterate(a.steps)
    # R timed regions of EXACTLY --steps iterations each, every one bracketed by barrier + synchronize on both sides
    # and MAX-reduced over the ranks; the reported value is the MEDIAN region (median + p10/p90)
    walls, evs = [], []
    for _ in range(max(1, a.repeats)):
        barrier()
        t0 = time.perf_counter()
        batch.iterate(a.steps)
        barrier()
        t1 = time.perf_counter()
        walls.append(D.max_over_ranks(dist, t1 - t0))
        evs.append(batch.last_iterate_ms())
    wall = float(np.median(walls))
    ev_ms = float(np.median(evs))
    # the one collective of the design: all-gather of the timing records, one record per window
    # {window_id, iterations, final_cost (filled after finish), seconds}; ranks with fewer windows pad with id -1
    summaries = None

    # ---- per-kernel attribution for the roofline (eager launches bracketed by HIP events) ----
    roofline = None
    if rank == 0 and a.profile_steps > 0:
        # per-launch durations of an eager pass bracketed by HIP events on the solver's stream; the MEDIAN launch is what
        # the roofline uses.  Launches in which an IMU factor re-preintegrates (the slowest workgroup of the linearise
        # launch, ~100 us) are reported separately: they depend on how far the biases move, not on the kernel.
        pl = batch.profile_launches(max(a.profile_steps, 30))
        prof = {k: float(np.median(v)) for k, v in pl.items()}          # ms per launch
        slow = pl["linearize"] > 1.5 * prof["linearize"]
        nbytes = batch.algorithmic_bytes()
        nbytes["linearize"] += nbytes.pop("small")
        # dominant kernel = most GPU time, i.e. launch time x the share of the 256 CUs the launch fills (the solve
        # kernel runs ONE workgroup per window on one CU each: at 64 windows it lasts as long as the linearise launch
        # but occupies a quarter of the device)
        st = solver.check_window(wins[0])
        n_cu, wg_per_cu = 256, 2
        share = {"solve": min(1.0, a.windows / n_cu),
                 "linearize": min(1.0, (st["n_group"] + a.keyframes) * a.windows / (n_cu * wg_per_cu)),
                 "schur": min(1.0, st["n_chunk"] * a.windows / (n_cu * wg_per_cu))}
        dom = max(prof, key=lambda k: prof[k] * share[k])
        per_launch_s = prof[dom] * 1e-3
        achieved = nbytes[dom] / per_launch_s / 1e9 if per_launch_s > 0 else 0.0
        pmc = None if (a.no_pmc or world > 1) else pmc_traffic(a, dom)
        # fp64 work of the linearise launch (~1.25 kflop per observation for residual + Jacobian +
        # J^T J / J^T r, + 0.25 kflop for the cost): what the vector ALUs have to issue, against the fp64 vector peak
        lin_flops = 1.5e3 * sum(w.n_obs for w in wins)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None if pmc is None else pmc["bytes"],
                    "traffic_detail": pmc,
                    "measured_traffic_frac": None if pmc is None else pmc["bytes"] / per_launch_s / 1e9 / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": nbytes[dom], "avg_launch_us": per_launch_s * 1e6,
                    "launch_us": {k: {"median": float(np.median(v)) * 1e3, "p10": float(np.percentile(v, 10)) * 1e3,
                                      "p90": float(np.percentile(v, 90)) * 1e3, "n": int(v.size)} for k, v in pl.items()},
                    "linearize_launches_with_imu_redo": {"count": int(slow.sum()),
                                                         "mean_us": float(pl["linearize"][slow].mean() * 1e3) if slow.any() else None},
                    "fp64": {"kernel": "linearize", "flops_per_launch": lin_flops,
                             "achieved_tflops": lin_flops / (prof["linearize"] * 1e-3) / 1e12, "peak_tflops": FP64_PEAK_TFLOPS,
                             "frac": lin_flops / (prof["linearize"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS},
                    "per_kernel_us": {k: v * 1e3 for k, v in prof.items()},
                    "per_kernel_algorithmic_bytes": nbytes, "per_kernel_cu_share": share,
                    "note": "dominant kernel = largest MEDIAN launch time x share of the CUs it fills; achieved = its "
                            "algorithmic (requested) bytes / median launch time.  Counters (profiles/) show the launch "
                            "latency-bound, neither HBM- nor FMA-bound: measured_traffic_frac (TCC bytes / 8 TB/s) and fp64.frac "
                            "say how far from either roof; W / V / b round trips between launches are served by L2 / "
                            "Infinity Cache, so measured traffic is below the algorithmic bytes"}
        # the timed loop does not launch all windows at once: it runs n_streams sub-batches side by side.  Launch medians at
        # THAT shape (one sub-batch alone on the device) and which kernel the stream time goes to.
        nst = a.streams if a.streams > 0 else (3 if a.windows >= 48 else 2 if a.windows >= 16 else 1)
        sub = (a.windows + nst - 1) // nst
        bs = solver.WindowBatch(wins[:sub], device=local_rank, options=opt)
        bs.begin()
        pls = {k: float(np.median(v)) * 1e3 for k, v in bs.profile_launches(30).items()}
        sb_bytes = bs.algorithmic_bytes()
        bs.finish()
        bs.close()
        chain = sum(pls.values())
        tdom = max(pls, key=pls.get)
        roofline["timed_loop_shape"] = {
            "streams": nst, "windows_per_launch": sub, "launch_us": pls, "stream_chain_us": chain,
            "measured_us_per_step": wall * 1e6 / a.steps, "time_dominant_kernel": tdom,
            "time_dominant_share_of_chain": pls[tdom] / chain,
            "time_dominant_algorithmic_GBps": sb_bytes[tdom] / (pls[tdom] * 1e-6) / 1e9,
            "note": "one iteration of a sub-batch is the dependent chain schur -> solve -> linearise on its stream; the solve "
                    "kernel (one 1024-thread workgroup per window, a 25-step block Cholesky in LDS) takes the same time for 1 "
                    "window or 64: it is bound by the latency of its dependent phases, not by HBM or the FMA rate"}
    summaries = batch.finish()
    n_rec = (a.total_windows + world - 1) // world if a.total_windows > 0 else a.windows
    rec = []
    for k in range(n_rec):
        if k < len(wins):
            rec += [float(ids[k]), float(summaries[k]["iterations"]), float(summaries[k]["final_cost"]), ev_ms * 1e-3]
        else:
            rec += [-1.0, 0.0, 0.0, 0.0]
    gathered = D.gather_records(dist, rec)
    records = [r[4 * k:4 * k + 4] for r in gathered for k in range(n_rec) if r[4 * k] >= 0]
    per_rank_ms = [r[3] * 1e3 for r in gathered]

    single = None
    if rank == 0:
        # latency of ONE window (configs[1] exactly), graph replay
        b1 = solver.WindowBatch(wins[:1], device=local_rank, options=opt)
        b1.begin()
        b1.iterate(a.steps)
        b1.synchronize()
        b1.iterate(a.steps)
        ms1 = b1.last_iterate_ms()
        b1.finish()
        b1.close()
        single = {"iterations_per_s": a.steps / (ms1 * 1e-3), "ms_per_iteration": ms1 / a.steps}

    cpu = cpu_mt = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:   # reported at N=1 only
        # CPU restatement (oracle/: same algorithm, same policy, g++ -O3), bounded sample of the same workload, on this
        # box's host cores: one thread, and OpenMP over the observation sweep + landmark Schur reduction at all cores.
        # Test infrastructure used here ONLY as the reported baseline, never in the measured path.  It is NOT Ceres
        # (not installable here); the reference runs Ceres with 2 threads (ThreadedKFVio.cpp:736).
        from tests import oracle_lib
        L = oracle_lib.lib()

        def cpu_rate(threads, budget_s):
            L.orc_set_threads(threads)
            ow = oracle_lib.OracleWindow(wins[0])
            n_probe = 10
            tp = ow.time_iterations(n_probe, opt)
            n = a.cpu_iters if a.cpu_iters > 0 else max(30, int(budget_s / max(tp / n_probe, 1e-6)))
            ow = oracle_lib.OracleWindow(wins[0])
            tc = ow.time_iterations(n, opt)
            L.orc_set_threads(1)
            return n / tc, n, tc

        v1, n1, t1 = cpu_rate(1, 8.0)
        cpu = {"value": v1, "unit": "iterations/s", "cores": 1, "kind": "port",
               "sample": f"1 window (configs[1] shape) x {n1} iterations of the CPU restatement (oracle/, g++ -O3 -fopenmp, "
                         f"same DOGLEG/Gauss-Newton mode as the GPU run; not Ceres), {t1:.1f} s"}
        # all cores: the windows of a batch are independent, so a CPU deployment runs one window per thread (OpenMP inside
        # one small window does not pay: measured 0.2x at 64 threads).  C threads, one window of the batch each.
        import threading
        def cpu_rate_windows(threads, n_each):
            ows = [oracle_lib.OracleWindow(wins[i % len(wins)]) for i in range(threads)]
            th = [threading.Thread(target=ows[i].time_iterations, args=(n_each, opt)) for i in range(threads)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            dt = time.perf_counter() - t0
            return threads * n_each / dt, dt

        ncpu = max(1, min(os.cpu_count() or 1, 256))
        if ncpu > 1:
            # short probes at all logical CPUs and at half of them (SMT), the better one gets the ~8 s sample
            probes = {c: cpu_rate_windows(c, 10)[0] for c in sorted({ncpu, max(1, ncpu // 2)})}
            ncore = max(probes, key=probes.get)
            n_each = max(10, int(8.0 * probes[ncore] / ncore))
            vm, tm = cpu_rate_windows(ncore, n_each)
            cpu_mt = {"value": vm, "unit": "iterations/s", "cores": ncore, "kind": "port",
                      "sample": f"{ncore} host threads (os.cpu_count() = {os.cpu_count()}; 10-iteration probes: "
                                + ", ".join(f"{c} threads {r:.0f} it/s" for c, r in probes.items()) +
                                f"), one window of the batch each, {n_each} iterations per window, {tm:.1f} s",
                      "speedup_over_1_core": vm / v1}

    if rank == 0:
        n_windows_total = a.total_windows if a.total_windows > 0 else world * a.windows
        total_iters = n_windows_total * a.steps
        value = total_iters / wall
        out = {
            "metric": "Gauss-Newton iterations/sec on 10-KF x 2-cam x 400-landmark windows (batch throughput: window-iterations/s)",
            "value": value, "unit": "iterations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": wall * 1e3 / a.steps, "higher_is_better": True,
            "scaling": "strong" if a.total_windows > 0 else "weak", "vs_baseline": None,
            "dtype": "f32+f64" if a.fp32 else "f64", "data": "synthetic",
            "config": {"workload": (f"{a.total_windows} windows in total over {world} GPU(s) (configs[3]) of BASELINE configs[1] "
                                    if a.total_windows > 0 else f"{a.windows} independent windows per GPU of BASELINE configs[1] ") +
                                   f"({a.keyframes} KF / 2 cam / {a.landmarks} landmarks / {wins[0].n_obs} obs / "
                                   f"{wins[0].n_imu} IMU factors x ~100 samples, fp64), Gauss-Newton mode, tolerances off",
                       "windows_per_gpu": a.windows, "observations_per_window": wins[0].n_obs,
                       "reduced_dim": wins[0].reduced_dim(), "graph": not a.no_graph, "parallelism": f"windows x{world}"},
            "hip_event_ms_per_step": max(per_rank_ms) / a.steps,
            "timed_regions": {"n": len(walls), "steps_each": a.steps, "statistic": "median",
                              "ms_per_step": {"median": wall * 1e3 / a.steps, "p10": float(np.percentile(walls, 10)) * 1e3 / a.steps,
                                              "p90": float(np.percentile(walls, 90)) * 1e3 / a.steps,
                                              "first": walls[0] * 1e3 / a.steps, "min": min(walls) * 1e3 / a.steps,
                                              "max": max(walls) * 1e3 / a.steps}},
            "window_records": {"fields": ["window_id", "iterations", "final_cost", "seconds"], "n": len(records),
                               "first": records[:2], "collective": (f"one all_gather, backend {dist.get_backend()}" + (" (= RCCL)" if dist.get_backend() == "nccl" else "")) if dist is not None else "none (1 rank)"},
            "single_window": single, "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_mt,
            "speedup_vs_cpu": None if cpu is None else {
                "single_window_vs_1_core": single["iterations_per_s"] / cpu["value"],
                "batch_vs_all_cores": None if cpu_mt is None else value / cpu_mt["value"],
                "note": "against this repository's CPU restatement, not against Ceres (north_star's 40x refers to Ceres)"},
            "final_cost_window0": summaries[0]["final_cost"],
        }
        print(json.dumps(out), flush=True)
    batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
