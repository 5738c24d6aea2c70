cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemv" 2>&1 | tail -3
VCLA_GEMV1X=0 python tools/bench_kernels.py gemv1 2>&1 | grep gemv1
VCLA_GEMV1X=1 python tools/bench_kernels.py gemv1 2>&1 | grep gemv1
