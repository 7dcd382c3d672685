cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r04_marg
timeout 900 python -m pytest tests/test_gpu_marginalization.py -m gpu -q -x > gpurun_out/r04_marg/pytest.log 2>&1
tail -25 gpurun_out/r04_marg/pytest.log
timeout 300 python scripts/bench_marginalize.py --config-c > gpurun_out/r04_marg/bench_marginalize.json 2> gpurun_out/r04_marg/bench_marginalize.err
tail -c 1500 gpurun_out/r04_marg/bench_marginalize.json; tail -3 gpurun_out/r04_marg/bench_marginalize.err
