cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r04_marg
OKVIS_BA_H0_CHECK=1 timeout 900 python -m pytest tests/test_gpu_marginalization.py -m gpu -q -x -s -k "large_prior_product" 2>&1 | grep "H0 check" | head
