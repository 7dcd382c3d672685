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
    p.add_argument("--repeats", type=int, default=50, help="timed regions of --steps iterations each (median / p10 / p90); at least this many")
    p.add_argument("--min-timed-s", type=float, default=3.0,
                   help="the timed regions together cover at least this much wall time (more regions of EXACTLY --steps iterations each are "
                        "added: a 20-step region of the default batch is 3 ms, too short for a 1 Hz utilisation sampler to see)")
    p.add_argument("--streams", type=int, default=0, help="sub-batch streams (0 = auto)")
    p.add_argument("--tune", action="append", default=[], metavar="FIELD=VALUE",
                   help="okvis_ba_options::tuning field for the timed batch (launch-shape experiments: fused_max_windows=64, group_lm=16, ...); "
                        "recorded under config.tuning")
    p.add_argument("--keyframes", type=int, default=10)
    p.add_argument("--landmarks", type=int, default=400)
    p.add_argument("--visibility", type=float, default=1.0)
    p.add_argument("--no-graph", action="store_true", help="eager launches (for rocprofv3 kernel traces)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-iters", type=int, default=0, help="oracle iterations for the CPU baseline (0 = auto ~12 s)")
    p.add_argument("--profile-steps", type=int, default=20, help="eager per-kernel HIP-event pass for the roofline")
    p.add_argument("--fp32", action="store_true", help="BASELINE configs[4]: fp32 Jacobian/Hessian build, fp64 solve (dtype f32+f64)")
    p.add_argument("--no-extras", action="store_true", help="skip the dogleg / configs[2] / strong-scaling sub-records")
    p.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 FETCH_SIZE/WRITE_SIZE passes (roofline.traffic = null)")
    p.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--pmc-separate", action="store_true", help=argparse.SUPPRESS)   # (child: the kernels of a sub-batch of a larger upload — not fused)
    return p.parse_args()


KERNEL_SYMBOL = {"schur": "schur_", "solve": "solve_kernel", "linearize": "linearize"}   # (schur_kernel | schur_mfma_kernel, linearize_kernel | linearize2_kernel)


def pmc_traffic(a, kernel, windows=None, separate=False):
    """HBM bytes per launch of `kernel` (launches of `windows` windows; default: all of the GPU's) from the TCC counters, collected the way MI355X_MICROARCH.md
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
               sys.executable, os.path.abspath(__file__), "--pmc-child", "--windows", str(windows or a.windows),
               "--keyframes", str(a.keyframes), "--landmarks", str(a.landmarks), "--visibility", str(a.visibility),
               "--steps", "12", "--warmup", "4", "--no-graph"] + (["--fp32"] if a.fp32 else []) + (["--pmc-separate"] if separate else [])
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
    for kv in a.tune:
        k, v = kv.split("=")
        setattr(opt.tuning, k, int(v, 0))
    opt.fp32_linearize = 1 if a.fp32 else 0
    opt.gauss_newton = 1  # every timed iteration does identical full work (no trust-region collapse at the optimum)
    if a.pmc_separate:
        opt.reserved0 = opt.reserved0 | 4
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
    n_regions = max(1, a.repeats)
    k = 0
    while k < n_regions:
        barrier()
        t0 = time.perf_counter()
        batch.iterate(a.steps)
        barrier()
        t1 = time.perf_counter()
        walls.append(D.max_over_ranks(dist, t1 - t0))
        evs.append(batch.last_iterate_ms())
        k += 1
        if k == 5 and a.min_timed_s > 0:   # (every rank computes the same count from the max-reduced times)
            n_regions = max(n_regions, min(20000, int(np.ceil(a.min_timed_s / max(float(np.median(walls)), 1e-6)))))
    wall = float(np.median(walls))
    ev_ms = float(np.median(evs))
    # the one collective of the design: all-gather of the timing records, one record per window
    # {window_id, iterations, final_cost (filled after finish), seconds}; ranks with fewer windows pad with id -1
    summaries = None

    # ---- per-kernel attribution for the roofline (eager launches bracketed by HIP events on the solver's stream) ----
    roofline = None
    if rank == 0 and a.profile_steps > 0:
        st = solver.check_window(wins[0])
        D_red = int(st["D"])
        n_cu, wg_per_cu = 256, 3

        def flops_per_launch(nw):
            # ~1.5 kflop per observation (residual + 2x15 Jacobian + J^T J / J^T r + cost) in the linearise
            # launch; landmark Schur complement 3 E_l^2 per landmark (E_l = 6 x blocks seeing it); reduced solve D^3 / 3 + 2 D^2
            obs = sum(w.n_obs for w in wins[:nw])
            lm = sum(w.n_lm for w in wins[:nw])
            E = 6.0 * a.keyframes
            return {"linearize": 1.5e3 * obs, "schur": 3.0 * E * E * lm, "solve": nw * (D_red ** 3 / 3.0 + 2.0 * D_red ** 2)}

        def kernel_table(b, nw):
            """median launch time (us) per kernel of batch b (nw windows per launch), with both roofline fractions"""
            pl = b.profile_launches(max(a.profile_steps, 30))
            nbytes = b.algorithmic_bytes()
            nbytes["linearize"] += nbytes.pop("small")
            fl = flops_per_launch(nw)
            share = {"solve": min(1.0, nw / n_cu),
                     "linearize": min(1.0, (st["n_group"] + a.keyframes) * nw / (n_cu * wg_per_cu)),
                     "schur": min(1.0, st["n_chunk"] * nw / (n_cu * wg_per_cu))}
            out = {}
            for k, v in pl.items():
                us = float(np.median(v)) * 1e3
                gbs = nbytes[k] / (us * 1e-6) / 1e9
                tf = fl[k] / (us * 1e-6) / 1e12
                out[k] = {"launch_us": us, "p10_us": float(np.percentile(v, 10)) * 1e3, "p90_us": float(np.percentile(v, 90)) * 1e3,
                          "n": int(v.size), "algorithmic_bytes": nbytes[k], "achieved_GBps": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
                          "flops": fl[k], "achieved_tflops": tf, "fp64_frac": tf / FP64_PEAK_TFLOPS, "cu_share": share[k]}
            slow = pl["linearize"] > 1.5 * float(np.median(pl["linearize"]))
            return out, {"count": int(slow.sum()), "mean_us": float(pl["linearize"][slow].mean() * 1e3) if slow.any() else None}

        # (a) the shape the timed loop launches: n_streams sub-batches side by side, each a dependent chain schur -> solve ->
        #     linearise.  The roofline entry names the kernel that takes most of that chain.
        nst = a.streams if a.streams > 0 else (2 if a.windows >= 128 else 3 if a.windows >= 56 else 2 if a.windows >= 8 else 1)   # the library's rule (okvis_ba_upload)
        sub = (a.windows + nst - 1) // nst
        # a batch of `sub` windows on its own would run in fused mode (up to 48 windows); the sub-batches of a
        # larger upload do not: profile the kernels the timed loop launches
        import copy
        opt_sub = copy.copy(opt)
        if a.windows > 48:
            opt_sub.reserved0 = opt_sub.reserved0 | 4
        bs = solver.WindowBatch(wins[:sub], device=local_rank, options=opt_sub)
        bs.begin()
        loop_tab, _ = kernel_table(bs, sub)
        bs.finish()
        bs.close()
        chain = sum(v["launch_us"] for v in loop_tab.values())
        dom = max(loop_tab, key=lambda k: loop_tab[k]["launch_us"])
        # (b) all windows of the GPU in one launch (the launch that fills the device: the linearise kernel)
        full_tab, redo = kernel_table(batch, a.windows)
        # HBM traffic of the dominant kernel at the SAME launch shape as its algorithmic bytes (a sub-batch of the timed loop),
        # and of the device-filling kernel at its own (all windows in one launch)
        pmc = None if (a.no_pmc or world > 1) else pmc_traffic(a, dom, windows=sub, separate=a.windows > 48)
        pmc_lin = None if (a.no_pmc or world > 1 or dom == "linearize") else pmc_traffic(a, "linearize")
        d = loop_tab[dom]
        roofline = {
            # the time-dominant kernel at the launch shape of the timed loop.  "latency": the SQ counters (profiles/) show its
            # waves waiting (dependent LDS / flag round trips of the block factorisation), neither roof is close; achieved /
            # peak / frac are its algorithmic bytes against the HBM roof, fp64 its flops against the fp64 (vector = matrix) peak
            "bound": "latency", "kernel": dom, "launch_shape": {"streams": nst, "windows_per_launch": sub},
            "achieved": d["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["hbm_frac"],
            "traffic": None if pmc is None else pmc["bytes"], "traffic_detail": pmc,
            "traffic_note": f"PMC passes over eager launches of {sub} windows: the launch shape of algorithmic_bytes_per_launch" if pmc else None,
            "traffic_over_algorithmic": None if pmc is None else pmc["bytes"] / d["algorithmic_bytes"],
            "avg_launch_us": d["launch_us"], "algorithmic_bytes_per_launch": d["algorithmic_bytes"],
            "fp64": {"kernel": dom, "flops_per_launch": d["flops"], "achieved_tflops": d["achieved_tflops"],
                     "peak_tflops": FP64_PEAK_TFLOPS, "frac": d["fp64_frac"]},
            "share_of_stream_chain": d["launch_us"] / chain, "stream_chain_us": chain,
            "measured_us_per_step": wall * 1e6 / a.steps,
            "kernels_timed_loop_shape": loop_tab,
            "kernels_all_windows_one_launch": full_tab,
            "device_filling_kernel": {"kernel": "linearize", "windows_per_launch": a.windows, **full_tab["linearize"],
                                      "traffic": None if pmc_lin is None else pmc_lin["bytes"],
                                      "measured_traffic_frac": None if pmc_lin is None else
                                      pmc_lin["bytes"] / (full_tab["linearize"]["launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS},
            "launch_us": {k: {"median": v["launch_us"], "p10": v["p10_us"], "p90": v["p90_us"], "n": v["n"]} for k, v in full_tab.items()},
            "linearize_launches_with_imu_redo": redo,
            "note": "one iteration of a sub-batch is the dependent chain schur -> solve -> linearise on its stream; `kernel` is the "
                    "longest link of that chain.  The solve kernel is one 1024-thread workgroup per window (decision, assembly, "
                    "blocked LDL^T on the fp64 matrix core with a one-wave diagonal chain, back-substitution, trial states); the "
                    "linearise launch is the one that fills all CUs (device_filling_kernel).  Algorithmic bytes / flops per "
                    "SURVEY.md section 8d; W / V / b round trips between launches are served by L2 / Infinity Cache"}
    def bytes_8d(w):
        """SURVEY.md section 8(d), "ALGORITHMIC bytes per iteration" (compulsory traffic: states in LDS / registers, W NOT materialised),
        from the actual O, K, C, L, M, D, D_m of window w: three observation sweeps 3*32*O; states 8*(7K + 9K + 7C + 4L) read three
        times and written once; reduced system 8*D^2 written + read; per-landmark V^-1, b_l 8*9*L written + read; IMU summaries
        8*290*M twice; prior 8*D_m^2 + 8*D_m.  Returns the total and the share this implementation's solve stage owns (reduced system
        + IMU summaries + prior + the pose / speed-bias states it writes)."""
        O, K, L, M = w.n_obs, w.n_sb, w.n_lm, w.n_imu           # (one speed/bias block per keyframe; n_pose also counts the
        Cc = w.n_pose - K                                       #  camera extrinsics blocks)
        Dr, Dm = w.reduced_dim(), int(np.asarray(w.marg_e0).size)
        obs, states = 3 * 32 * O, 4 * 8 * (7 * K + 9 * K + 7 * Cc + 4 * L)
        red, lmk, imu, prior = 2 * 8 * Dr * Dr, 2 * 8 * 9 * L, 2 * 8 * 290 * M, 8 * Dm * Dm + 8 * Dm
        return {"total": obs + states + red + lmk + imu + prior, "solve": red + imu + prior + 8 * (7 * K + 9 * K),
                "terms": {"observation_sweeps": obs, "states": states, "reduced_system": red, "landmark_blocks": lmk, "imu": imu, "prior": prior}}

    if roofline is not None:
        # the WHOLE step against the roofs (algorithmic bytes and flops of one iteration of one window x the
        # windows, over the measured time of a step) — the number that describes the headline, next to its longest link above
        def step_record(nw, ms_per_step, nbytes, fl, ws):
            B, F = float(sum(nbytes.values())), float(sum(fl.values()))
            B8 = float(sum(bytes_8d(w)["total"] for w in ws))
            return {"windows": nw, "ms_per_step": ms_per_step, "flops_per_step": F, "flops_per_window_iteration": F / nw,
                    # two byte counts, both per window-iteration.  bytes_8d: SURVEY.md section 8(d)'s compulsory traffic (W not
                    # materialised).  bytes_impl: what THIS implementation's launches exchange through HBM by design — it
                    # materialises W (144 B per (landmark, block) pair, written once by the linearise launch and read by the Schur
                    # launch and by the next linearise launch's back-substitution) and the per-chunk Schur partials.
                    "bytes_8d_per_window_iteration": B8 / nw, "bytes_impl_per_window_iteration": B / nw,
                    "impl_over_8d": B / B8,
                    "achieved_GBps_8d": B8 / (ms_per_step * 1e-3) / 1e9, "frac_8d": B8 / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_step": B, "bytes_per_window_iteration": B / nw,
                    "achieved_GBps": B / (ms_per_step * 1e-3) / 1e9, "hbm_frac": B / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "achieved_tflops": F / (ms_per_step * 1e-3) / 1e12, "fp64_frac": F / (ms_per_step * 1e-3) / 1e12 / FP64_PEAK_TFLOPS}
        nb_all = batch.algorithmic_bytes()
        roofline["step"] = step_record(a.windows, wall * 1e3 / a.steps, nb_all, flops_per_launch(a.windows), wins)
        roofline["step"]["note"] = ("frac_8d = SURVEY.md section 8(d) bytes / measured step time / 8 TB/s; hbm_frac = the same with bytes_impl, "
                                    "which adds the materialised W and the Schur partials (impl_over_8d x the compulsory traffic)")
        # the dominant kernel against section 8(d)'s bytes of ITS stage (reduced system written + read, IMU summaries, prior, states written)
        d8 = float(sum(bytes_8d(w)["solve" if dom == "solve" else "total"] for w in wins[:sub]))
        roofline["bytes_8d_per_launch"] = d8
        roofline["achieved_8d"] = d8 / (d["launch_us"] * 1e-6) / 1e9
        roofline["frac_8d"] = roofline["achieved_8d"] / HBM_PEAK_GBS
        roofline["step"]["bound"] = "latency"
        split_small = bool(batch.launch_route()["split_small"])
        roofline["step"]["chain"] = ("one iteration of a sub-batch = " + ("4" if split_small else "3") +
                                     " dependent launches on its stream (Schur, solve, linearise" + (", IMU / prior factors" if split_small else "") +
                                     f"), {nst} sub-batches side by side; the solve launch occupies {sub} of 256 CUs")
        if not a.no_extras and world == 1:
            # the saturated shape: 256 windows per GPU (4 x the headline's batch), same windows repeated with fresh seeds
            nsat = 256
            wsat = wins + [synthetic.make_window(a.keyframes, a.landmarks, a.visibility, 20250000 + i) for i in range(nsat - len(wins))]
            bsat = solver.WindowBatch(wsat[:nsat], device=local_rank, options=opt)
            bsat.begin()
            bsat.iterate(a.warmup)
            bsat.iterate(a.steps)
            bsat.synchronize()
            bsat.iterate(a.steps)
            ms_sat = bsat.last_iterate_ms() / a.steps
            nb_sat = bsat.algorithmic_bytes()
            bsat.finish()
            bsat.close()
            obs_s, lm_s = sum(w.n_obs for w in wsat[:nsat]), sum(w.n_lm for w in wsat[:nsat])
            fl_sat = {"linearize": 1.5e3 * obs_s, "schur": 3.0 * (6.0 * a.keyframes) ** 2 * lm_s, "solve": nsat * (D_red ** 3 / 3.0 + 2.0 * D_red ** 2)}
            roofline["step_saturated"] = step_record(nsat, ms_sat, nb_sat, fl_sat, wsat[:nsat])
            roofline["step_saturated"]["iterations_per_s"] = nsat / (ms_sat * 1e-3)
    summaries = batch.finish()
    # ---- the timed batch against the oracle, AFTER the timed region (the checker, never the thing measured).  (i) the states the
    #      timed iterations ended in: thousands of Gauss-Newton iterations sit at the fixed point, which the oracle reaches from the
    #      same start (it iterates until its cost is stationary to 1e-14); (ii) the same solver, the same options, the same windows
    #      uploaded again (shapes unchanged: the captured graphs of the timed region are replayed) for optimize(10) from the start —
    #      an iteration-for-iteration comparison.  Four windows sampled over the three sub-batches.
    oracle_check = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not a.pmc_child and a.total_windows == 0:
        from tests import oracle_lib as _ol
        sample = sorted({0, a.windows // 3, (2 * a.windows) // 3, a.windows - 1})
        dev_fix = 0.0
        for i in sample:
            ow = _ol.OracleWindow(wins[i])
            prev = None
            for _ in range(40):
                r = ow.optimize(5, opt)
                if prev is not None and abs(r["final_cost"] - prev) <= 1e-14 * prev:
                    break
                prev = r["final_cost"]
            dev_fix = max(dev_fix, abs(summaries[i]["final_cost"] - r["final_cost"]) / r["final_cost"])
        batch.upload(wins)
        s10 = batch.optimize(10)
        dev_10, same_book = 0.0, True
        for i in sample:
            r = _ol.OracleWindow(wins[i]).optimize(10, opt)
            dev_10 = max(dev_10, abs(s10[i]["final_cost"] - r["final_cost"]) / r["final_cost"])
            same_book = same_book and (s10[i]["iterations"], s10[i]["successful_steps"]) == (r["iterations"], r["successful_steps"])
        oracle_check = {"windows_checked": sample, "max_rel_cost_dev_vs_oracle": max(dev_fix, dev_10),
                        "after_timed_region_fixed_point": dev_fix, "optimize_10_from_the_start": dev_10,
                        "identical_iteration_bookkeeping": bool(same_book), "route": batch.launch_route(),
                        "note": "the timed solver object after the timed region, bench options, against oracle/ (CPU restatement pinned "
                                "to the reference's sources, DESIGN.md section 3); tests/test_gpu_batch64.py checks all 64 windows"}
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
        late = b1.helper_timeouts()   # solve launches whose helper workgroups were late (summed in the solving workgroup then)
        b1.close()
        single = {"iterations_per_s": a.steps / (ms1 * 1e-3), "ms_per_iteration": ms1 / a.steps, "helper_timeouts": late}

    cpu = cpu_mt = cpu_2t = None
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
        v2, n2, t2 = cpu_rate(2, 4.0)
        cpu = {"value": v1, "unit": "iterations/s", "cores": 1, "kind": "port",
               "sample": f"1 window (configs[1] shape) x {n1} iterations of the CPU restatement (oracle/, g++ -O3 -fopenmp, "
                         f"same DOGLEG/Gauss-Newton mode as the GPU run; not Ceres), {t1:.1f} s"}
        cpu_2t = {"value": v2, "unit": "iterations/s", "cores": 2, "kind": "port",
                  "sample": f"the same window x {n2} iterations with 2 OpenMP threads over the observation sweep and the landmark Schur "
                            f"reduction — the thread count the application hands to Ceres (ThreadedKFVio.cpp:736: optimize(..., 2, false)); {t2:.1f} s",
                  "speedup_over_1_core": v2 / v1}
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

    # ---- the reference's configured mode: DOGLEG with default tolerances from the perturbed start, optimize(10) per window
    #      (Estimator::optimize, Estimator.cpp:843-906: first-linearisation redoPreintegration included) over fresh windows
    dogleg = None
    config_c = None
    strong = None
    frontend = None
    if rank == 0 and not a.no_extras and not a.pmc_child:
        dopt = default_options()
        dopt.use_graph = 0 if a.no_graph else 1
        dopt.n_streams = a.streams
        for kv in a.tune:
            k, v = kv.split("=")
            setattr(dopt.tuning, k, int(v, 0))
        n_batches, per_batch = 4, max(16, min(64, a.windows))
        it_total = slot_total = redo_total = 0
        t_total = 0.0
        term = {}
        bd = None
        for bi in range(n_batches + 1):   # the first batch warms the solver up (allocations, launch graphs) and is not counted
            fresh = [synthetic.make_window(a.keyframes, a.landmarks, a.visibility, 7_000_000 + 1000 * bi + i) for i in range(per_batch)]
            if bd is None:
                bd = solver.WindowBatch(fresh, device=local_rank, options=dopt)
            else:
                bd.upload(fresh)      # one solver serves batch after batch, like a deployment would
            bd.synchronize()
            t0 = time.perf_counter()
            sm = bd.optimize(10)
            bd.synchronize()
            dt = time.perf_counter() - t0
            if bi > 0:
                t_total += dt
                it_total += sum(x["iterations"] for x in sm)
                slot_total += int(bd.array("SLOTS")[0]) * per_batch
                redo_total += int(sum(bd.array("IMU_REDO_COUNT", w).sum() for w in range(per_batch)))
                for x in sm:
                    term[x["termination"]] = term.get(x["termination"], 0) + 1
        bd.close()
        dogleg = {"mode": "DOGLEG, Jacobi scaling, default tolerances (Estimator.cpp:854-873), optimize(10) from the perturbed start",
                  "windows": n_batches * per_batch, "windows_per_call": per_batch,
                  "counted_iterations": it_total, "iterations_per_s": it_total / t_total,
                  "windows_per_s": n_batches * per_batch / t_total, "ms_per_optimize_call": t_total / n_batches * 1e3,
                  "launch_slots": slot_total, "launch_slots_per_counted_iteration": slot_total / max(1, it_total),
                  "imu_repreintegrations": redo_total, "imu_repreintegrations_per_factor": redo_total / max(1, n_batches * per_batch * fresh[0].n_imu),
                  "termination_histogram": {str(k): v for k, v in sorted(term.items())},
                  "note": "wall time of okvis_ba_optimize (begin + first linearisation incl. IMU re-preintegration + 10 iteration "
                          "slots + top-up slots + final decision + landmark quality), upload excluded; a launch slot is one schur + "
                          "solve + linearise triple for the whole batch; termination 0 = iteration cap, 1 = function tolerance"}
        # ---- frontend pieces (include/okvis_amd_frontend.h): stereo triangulation + uncertainty for the candidate matches of one
        #      frame pair, host buffers in and out (the PCIe-inclusive rate a matcher sees)
        from okvis_amd import frontend as FE
        rng = np.random.default_rng(7)
        n_kp, n_pairs = 2000, 4096
        fcam = FE.camera(synthetic.TEST_INTR_RADTAN, 1)
        T_AB = np.array([0.11, 0.005, -0.002, 0.0, 0.01, 0.0, 1.0]); T_AB[3:] /= np.linalg.norm(T_AB[3:])
        depth = rng.uniform(0.8, 20.0, n_kp)
        pA = np.c_[rng.uniform(-0.5, 0.5, n_kp) * depth, rng.uniform(-0.35, 0.35, n_kp) * depth, depth]
        uvA, _ = synthetic.project_points(synthetic.TEST_INTR_RADTAN, 1, pA)
        uvB, _ = synthetic.project_points(synthetic.TEST_INTR_RADTAN, 1, pA - T_AB[:3])
        kpA = np.c_[np.nan_to_num(uvA, nan=100.0), np.full(n_kp, 8.0)].astype(np.float32)
        kpB = np.c_[np.nan_to_num(uvB, nan=100.0) + rng.normal(size=(n_kp, 2)) * 0.4, np.full(n_kp, 8.0)].astype(np.float32)
        fpairs = np.c_[rng.integers(0, n_kp, n_pairs), rng.integers(0, n_kp, n_pairs)].astype(np.int32)
        fpairs[: n_kp] = np.c_[np.arange(n_kp), np.arange(n_kp)]
        fctx = FE.Frontend(local_rank)
        UO = np.diag([1e-2] * 3 + [1e-8] * 3)
        for _ in range(3):
            fctx.stereo_triangulate(fcam, fcam, T_AB, UO, kpA, kpB, fpairs)
        tf0 = time.perf_counter()
        n_calls = 50
        for _ in range(n_calls):
            _, _, fflags = fctx.stereo_triangulate(fcam, fcam, T_AB, UO, kpA, kpB, fpairs)
        tf1 = time.perf_counter()
        fctx.close()
        frontend = {"entry": "okvis_fe_stereo_triangulate with uncertainty (ProbabilisticStereoTriangulator.cpp:178-355)",
                    "candidates_per_call": n_pairs, "keypoints_per_image": n_kp, "ms_per_call": (tf1 - tf0) / n_calls * 1e3,
                    "candidates_per_s": n_pairs * n_calls / (tf1 - tf0), "valid_fraction": float((fflags & 1).mean()),
                    "note": "host buffers in and out (one H2D, one kernel, one D2H per call), ctypes overhead included"}
        # ---- BASELINE configs[2]: 50 keyframes / 2000 landmarks / 200 000 observations (HBM-resident tiled solve), one window
        wc = synthetic.config_C(seed=20240923, visibility=a.visibility)
        bc = solver.WindowBatch([wc], device=local_rank, options=opt)
        bc.begin()
        bc.iterate(5)
        bc.synchronize()
        msc = []
        for _ in range(7):
            bc.iterate(20)
            bc.synchronize()
            msc.append(bc.last_iterate_ms() / 20)
        plc = {k: float(np.median(v)) * 1e3 for k, v in bc.profile_launches(12).items()}
        bytes_c = bc.algorithmic_bytes()
        bc.finish()
        bc.close()
        # flops of one iteration by SURVEY.md section 8d (as in roofline.step): 1.5 kflop per observation, 3 E_l^2 per landmark with
        # E_l = 6 x the poses that see it (all 50 here), D^3 / 3 + 2 D^2 for the reduced solve; against the fp64 peak (matrix = vector)
        Dc = wc.reduced_dim()
        fl_c = {"linearize": 1.5e3 * wc.n_obs, "schur": 3.0 * (6.0 * 50) ** 2 * wc.n_lm, "solve": Dc ** 3 / 3.0 + 2.0 * Dc ** 2}
        t_c = float(np.median(msc)) * 1e-3
        B_c = float(sum(bytes_c.values()))
        config_c = {"workload": f"1 window, 50 KF / 2 cam / 2000 landmarks / {wc.n_obs} observations, D = {Dc}, Gauss-Newton mode",
                    "ms_per_iteration": float(np.median(msc)), "iterations_per_s": 1e3 / float(np.median(msc)),
                    "launch_us": plc, "note": "launch_us.solve = assembly + tile export + tiled fp64-MFMA Cholesky + tail (four launches)",
                    "roofline": {"bound": "latency (one window: the tiled Cholesky is a chain of 16 tile columns, the Schur launch 2000 landmarks x 1275 block pairs on fp64 FMA)",
                                 "flops_per_iteration": fl_c, "achieved_tflops": sum(fl_c.values()) / t_c / 1e12,
                                 "fp64_peak_tflops": FP64_PEAK_TFLOPS, "fp64_frac": sum(fl_c.values()) / t_c / 1e12 / FP64_PEAK_TFLOPS,
                                 "per_launch_tflops": {k: fl_c[k] / (plc[k] * 1e-6) / 1e12 for k in fl_c if plc.get(k)},
                                 "algorithmic_bytes_per_iteration": B_c, "achieved_GBps": B_c / t_c / 1e9, "hbm_frac": B_c / t_c / 1e9 / HBM_PEAK_GBS}}
    fp32 = None
    if rank == 0 and not a.no_extras and not a.pmc_child and not a.fp32:
        # ---- BASELINE configs[4]: fp32 Jacobian / Hessian build, fp64 Schur complement and reduced solve (okvis_ba_options.
        #      fp32_linearize), on the windows and in the mode of the headline line; and what it does to the result in the
        #      reference's mode (DOGLEG, default tolerances, optimize(10) from the perturbed start) on window 0
        import copy
        o32 = copy.copy(opt)
        o32.fp32_linearize = 1
        b32 = solver.WindowBatch(wins, device=local_rank, options=o32)
        b32.begin()
        b32.iterate(a.warmup)
        b32.synchronize()
        ms32 = []
        for _ in range(9):
            b32.iterate(a.steps)
            b32.synchronize()
            ms32.append(b32.last_iterate_ms() / a.steps)
        pl32 = {k: float(np.median(v)) * 1e3 for k, v in b32.profile_launches(20).items()}
        b32.finish()
        b32.close()
        dev = {}
        for name, f32 in (("fp64", 0), ("mixed", 1)):
            od = default_options()
            od.fp32_linearize = f32
            bdv = solver.WindowBatch([wins[0]], device=local_rank, options=od)
            dev[name] = (bdv.optimize(10)[0], bdv.get_state(0))
            bdv.close()
        s64, s32 = dev["fp64"][0], dev["mixed"][0]
        fp32 = {"mode": "fp32_linearize = 1: residual, Jacobians and the J^T J / J^T r products of the linearise launch in fp32; landmark "
                        "Schur complement, IMU / prior factors and reduced solve in fp64",
                "iterations_per_s": len(wins) / (float(np.median(ms32)) * 1e-3), "ms_per_step": float(np.median(ms32)),
                "fp64_iterations_per_s": len(wins) * a.steps / wall, "speedup_over_fp64": (wall / a.steps * 1e3) / float(np.median(ms32)),
                "launch_us": pl32,
                "dogleg_window_0": {"final_cost_fp64": s64["final_cost"], "final_cost_mixed": s32["final_cost"],
                                    "final_cost_rel_dev": abs(s32["final_cost"] - s64["final_cost"]) / s64["final_cost"],
                                    "iterations_fp64": s64["iterations"], "iterations_mixed": s32["iterations"],
                                    "max_position_dev_m": float(np.abs(dev["mixed"][1][0][:, :3] - dev["fp64"][1][0][:, :3]).max()),
                                    "max_landmark_dev_m": float(np.abs(dev["mixed"][1][2][:, :3] - dev["fp64"][1][2][:, :3]).max())},
                "study": "profiles/r04_mixed_precision.json (8 seeds, 10 and 30 iterations)"}
    frame_host = None
    if rank == 0 and not a.no_extras and not a.pmc_child:
        # ---- what surrounds the iterations of one frame of the estimator (one window, 8 frames, replay-sized): structure upload,
        #      the same change as an okvis_ba_patch (newest frame replaced), one marginalisation call
        try:
            from okvis_amd.window import Patch
            wf = synthetic.make_window(8, 430, 0.5, seed=20240924)
            K, npz = 8, wf.n_pose
            newest = np.flatnonzero(np.asarray(wf.obs_pose) == K - 1)
            pf = Patch(remove_pose=[K - 1], remove_sb=[K - 1], add_pose=wf.pose[K - 1:K], add_pose_fixed=[0], add_sb=wf.sb[K - 1:K],
                       add_sb_fixed=[0], add_obs_lm=wf.obs_lm[newest], add_obs_pose=np.full(newest.size, npz - 1),
                       add_obs_ext=np.asarray(wf.obs_ext)[newest] - 1, add_obs_cam=wf.obs_cam[newest], add_obs_uv=wf.obs_uv[newest],
                       add_obs_sqrtw=wf.obs_sqrtw[newest], add_imu_pose0=[K - 2], add_imu_sb0=[K - 2], add_imu_pose1=[npz - 1],
                       add_imu_sb1=[K - 1], add_imu_t0=wf.imu_t0[-1:], add_imu_t1=wf.imu_t1[-1:], add_imu_s_begin=[0],
                       add_imu_s_count=wf.imu_s_count[-1:],
                       add_imu_s_t=wf.imu_s_t[wf.imu_s_begin[-1]:wf.imu_s_begin[-1] + wf.imu_s_count[-1]],
                       add_imu_s_gyr=wf.imu_s_gyr[wf.imu_s_begin[-1]:wf.imu_s_begin[-1] + wf.imu_s_count[-1]],
                       add_imu_s_acc=wf.imu_s_acc[wf.imu_s_begin[-1]:wf.imu_s_begin[-1] + wf.imu_s_count[-1]])
            fopt = default_options()
            bf = solver.WindowBatch([wf], device=local_rank, options=fopt, patchable=True)
            bf.optimize(3)
            # (like okvis_amd::Estimator, the caller hands every IMU term's preintegration back with the window: okvis_ba_fetch_imu_caches)
            res0 = bf.fetch_results(0)
            wf.imu_sb_ref, wf.imu_cache, wf.imu_sb_ref_valid = res0["imu_sb_ref"], bf.fetch_imu_caches(0), np.full(wf.n_imu, 2, np.uint8)
            t_up, t_patch, t_opt = [], [], []
            import ctypes as C
            pc, keep_p = pf.as_c()
            from okvis_amd.window import WindowC
            wfc, keep_w = wf.as_c()
            arr_w = (WindowC * 1)(wfc)
            for rep in range(12):
                bf.synchronize()
                t0 = time.perf_counter()
                rcu = bf._L.okvis_ba_upload(bf._h, 1, arr_w)   # (the C call itself: no Python marshalling in the timed part)
                t1 = time.perf_counter()
                assert rcu == 0, rcu
                bf.optimize(10)
                t2 = time.perf_counter()
                rcp = bf._L.okvis_ba_patch_window(bf._h, 0, C.byref(pc))
                t3 = time.perf_counter()
                assert rcp == 0, rcp
                if rep >= 2:
                    t_up.append(t1 - t0); t_opt.append(t2 - t1); t_patch.append(t3 - t2)
            n_obs_after = bf.patched_view(0).n_obs
            bf.close()
            wm = synthetic.small_window(seed=9, K=6, L=150, visibility=0.8)
            pmm = np.zeros(wm.n_pose, np.uint8); smm = np.zeros(wm.n_sb, np.uint8); pmm[0] = 1; smm[[0, 1]] = 1
            bm = solver.WindowBatch([wm], device=local_rank, options=default_options())
            t_marg = []
            for rep in range(14):
                t0 = time.perf_counter()
                gm = bm.marginalize(0, pmm, smm)
                if rep >= 2:
                    t_marg.append(time.perf_counter() - t0)
            bm.close()
            frame_host = {"window": f"8 frames / {wf.n_lm} landmarks / {wf.n_obs} observations (the replay's size), one-window solver",
                          "upload_ms": float(np.median(t_up)) * 1e3, "optimize10_ms": float(np.median(t_opt)) * 1e3,
                          "patch_newest_frame_ms": float(np.median(t_patch)) * 1e3, "observations_after_patch": int(n_obs_after),
                          "marginalize_ms": float(np.median(t_marg)) * 1e3, "marginalize_kept_dim": int(gm["dim"]),
                          "note": "host wall clock through ctypes; upload = okvis_ba_upload (index build + arena fill; the IMU terms carry their preintegration records), patch = "
                                  "okvis_ba_patch_window replacing the newest frame with its observations and IMU term (container edit + "
                                  "the same index build; the blocks that stay keep the device's values), marginalize = one "
                                  "okvis_ba_marginalize call on a 6-frame sub-window (150 landmarks, one pose and two speed/bias blocks eliminated)"}
            # the same frame inside okvis_amd::Estimator: the C++ replay of a synthetic ASL recording (80 frames, windows of 8 frames),
            # once with the window patched between frames (the default) and once flattened + uploaded every frame
            try:
                import re, subprocess, tempfile
                from okvis_amd import recording
                exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "okvis_amd", "lib", "okvis_amd_replay")
                with tempfile.TemporaryDirectory() as dsyn:
                    recording.write_synthetic_recording(dsyn, duration_s=8.0)
                    rep = {}
                    for name, extra in (("patch", []), ("flatten_upload", ["--no-patch"])):
                        out = subprocess.run([exe, dsyn] + extra, capture_output=True, text=True, timeout=120).stdout
                        m = re.search(r"medians per frame: optimize ([\d.]+) \(window ([\d.]+) \+ hand-over ([\d.]+) \+ iterations ([\d.]+) \+ "
                                      r"download ([\d.]+)\) \+ marginalise ([\d.]+) ms", out)
                        if m:
                            v = [float(x) for x in m.groups()]
                            rep[name] = {"optimize_ms": v[0], "window_description_ms": v[1], "hand_over_ms": v[2], "iterations_ms": v[3],
                                         "download_ms": v[4], "marginalise_ms": v[5]}
                    frame_host["estimator_replay"] = dict(rep, note="medians per frame of okvis_amd_replay (okvis_amd::Estimator, optimize(10) + "
                                                                   "applyMarginalizationStrategy per frame): window description = the edits since "
                                                                   "the last frame as one okvis_ba_patch, or a full flatten, plus the wait for the numbers of the marginalisation the previous "
                                                                   "frame enqueued (okvis_ba_marginalize_begin / _end); hand-over = "
                                                                   "okvis_ba_patch_window, or okvis_ba_upload")
            except Exception as e:
                frame_host["estimator_replay"] = {"error": repr(e)}
        except Exception as e:   # a diagnostic record must never take the bench line with it
            frame_host = {"error": repr(e)}
    if not a.no_extras and not a.pmc_child and world > 1 and a.total_windows == 0:
        # ---- BASELINE configs[3] as written: 64 windows IN TOTAL over the ranks (strong scaling), next to the weak line
        tw = 64
        ids_s = D.shard_windows(tw, rank, world)
        wins_s = [synthetic.make_window(a.keyframes, a.landmarks, a.visibility, 20240923 + i) for i in ids_s]
        bsx = solver.WindowBatch(wins_s, device=local_rank, options=opt)
        bsx.begin()
        bsx.iterate(a.warmup)
        bsx.iterate(a.steps)
        ws = []
        for _ in range(max(5, a.repeats // 5)):
            barrier()
            t0 = time.perf_counter()
            bsx.iterate(a.steps)
            bsx.synchronize()
            if dist is not None:
                dist.barrier()
            ws.append(D.max_over_ranks(dist, time.perf_counter() - t0))
        bsx.finish()
        bsx.close()
        if rank == 0:
            wmed = float(np.median(ws))
            strong = {"total_windows": tw, "windows_on_rank0": len(wins_s), "ranks": world,
                      "backend": dist.get_backend() if dist is not None else None,
                      "value": tw * a.steps / wmed, "unit": "iterations/s", "ms_per_step": wmed * 1e3 / a.steps,
                      "scaling": "strong", "note": "BASELINE configs[3]: 64 independent windows sharded over the GPUs (8 per GPU at 8 GPUs)"}
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
                       "reduced_dim": wins[0].reduced_dim(), "graph": not a.no_graph, "parallelism": f"windows x{world}",
                       **({"tuning": a.tune} if a.tune else {})},
            "hip_event_ms_per_step": max(per_rank_ms) / a.steps,
            "timed_regions": {"n": len(walls), "steps_each": a.steps, "statistic": "median",
                              "ms_per_step": {"median": wall * 1e3 / a.steps, "p10": float(np.percentile(walls, 10)) * 1e3 / a.steps,
                                              "p90": float(np.percentile(walls, 90)) * 1e3 / a.steps,
                                              "first": walls[0] * 1e3 / a.steps, "min": min(walls) * 1e3 / a.steps,
                                              "max": max(walls) * 1e3 / a.steps}},
            "window_records": {"fields": ["window_id", "iterations", "final_cost", "seconds"], "n": len(records),
                               "first": records[:2], "collective": (f"one all_gather, backend {dist.get_backend()}" + (" (= RCCL)" if dist.get_backend() == "nccl" else "")) if dist is not None else "none (1 rank)"},
            "single_window": single, "roofline": roofline, "dogleg": dogleg, "config_C": config_c, "fp32": fp32, "frontend": frontend, "frame_host": frame_host, "strong_scaling_64_windows": strong,
            "ranks_seen_by_collective": world if dist is None else dist.get_world_size(),
            "cpu_baseline": cpu, "cpu_baseline_2_threads": cpu_2t, "cpu_baseline_all_cores": cpu_mt,
            "oracle_check": oracle_check,
            "max_rel_cost_dev_vs_oracle": None if oracle_check is None else oracle_check["max_rel_cost_dev_vs_oracle"],
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
