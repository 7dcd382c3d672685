#!/bin/bash
# round 5: the whole GPU suite, the default bench line (PMC passes + CPU baselines + DOGLEG / configs[2] / fp32 / saturated records),
# configs[3] as written, the batch sweep, rocprofv3 kernel traces of the bench command (eager and graph) with the per-launch-shape
# split, SQ + MFMA counters of an eager 64-window run, configs[2] trace, replay timing, phase stamps of the solve kernel and of the
# linearise / IMU workgroups, the LDL^T micro-benchmark.  -> gpurun_out/prof_r05/ (copied into profiles/r05_* afterwards)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r05
mkdir -p $O
cd $R
export TMPDIR=/tmp
OKVIS_AMD_CHECK_PATCH=1 timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
head -c 300 $O/bench_default.json; echo
python bench.py --total-windows 64 --no-pmc --no-cpu-baseline --no-extras > $O/bench_total64.json 2> $O/bench_total64.err
for n in 1 2 4 8 16 32 64 128 256; do
  python bench.py --windows $n --no-pmc --no-cpu-baseline --no-extras --profile-steps 0 --repeats 15 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'windows': $n, 'iterations_per_s': d['value'], 'us_per_step': d['ms_per_step']*1e3}))"
done > $O/batch_sweep.jsonl
cat $O/batch_sweep.jsonl
for n in 1 8 22 64; do timeout 120 python tools/gpu_solve_stamps.py $n > $O/solve_stamps_$n.txt 2>&1; done
timeout 120 python tools/gpu_prof_stamps.py > $O/linearize_stamps.txt 2>&1
timeout 60 tests/micro/bin/ldl16 > $O/ldl16_micro.txt 2>&1
for b in branch_lat lds_lat icache rcp_acc mfma_lat; do echo "== $b"; timeout 60 tests/micro/bin/$b; done > $O/micro_costs.txt 2>&1
cd /tmp
for mode in eager graph; do
  flag=""; [ $mode = eager ] && flag="--no-graph"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$mode -o p -- \
    python $R/bench.py $flag --no-pmc --no-cpu-baseline --no-extras --repeats 5 > $O/bench_$mode.json 2> $O/bench_$mode.err
  f=$(find $O/$mode -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats_$mode.csv
  t=$(find $O/$mode -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python $R/scripts/kernel_trace_by_shape.py $t > $O/kernel_by_shape_$mode.csv
  rm -rf $O/$mode
done
summarise() {   # counter_collection.csv -> per kernel mean per launch
python - "$1" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not k.startswith("ba::"): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-30s %16.0f per launch" % (c, v / max(1, n[(k, c)])))
PY
}
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc -o p -- \
  python $R/bench.py --no-graph --no-pmc --no-cpu-baseline --no-extras --repeats 2 --steps 20 --warmup 4 --profile-steps 0 > /dev/null 2> $O/pmc.err
c=$(find $O/pmc -name "*counter_collection.csv" | head -1); [ -n "$c" ] && summarise "$c" > $O/pmc_sq_64windows.txt
rm -rf $O/pmc
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc2 -o p -- \
  python $R/bench.py --no-graph --no-pmc --no-cpu-baseline --no-extras --repeats 2 --steps 20 --warmup 4 --profile-steps 0 > /dev/null 2> $O/pmc2.err
c=$(find $O/pmc2 -name "*counter_collection.csv" | head -1); [ -n "$c" ] && summarise "$c" > $O/pmc_mfma_64windows.txt
rm -rf $O/pmc2
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c_trace -o p -- python $R/scripts/bench_config_c.py > $O/bench_config_C.json 2>/dev/null
f=$(find $O/c_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_config_C.csv
rm -rf $O/c_trace
cd $R
# the long double referee: spread of the far-start DOGLEG case, the random sweep's seeds, two D = 300 windows
timeout 300 python tools/gpu_referee_spread.py > $O/referee_spread41.txt 2>&1
timeout 300 python tools/gpu_sweep_gaps.py > $O/referee_sweep_gaps.txt 2>&1
timeout 300 python tools/gpu_referee_large.py > $O/referee_large.txt 2>&1
timeout 300 python tools/gpu_referee_marg.py > $O/referee_marg.txt 2>&1
timeout 300 python tools/gpu_tolerance_audit.py > $O/tolerance_audit.txt 2>&1
timeout 200 python tools/gpu_referee_dogleg_iters.py 2 30 > $O/referee_dogleg_iters.txt 2>&1
timeout 200 python tools/gpu_replay_timing.py > $O/replay_timing.txt 2>&1
grep -E "medians|route" $O/replay_timing.txt | head
timeout 200 python scripts/bench_marginalize.py --config-c > $O/bench_marginalize.json 2> $O/bench_marginalize.err
echo done
