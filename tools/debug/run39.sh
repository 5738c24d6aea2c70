cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "dstream or deferred" 2>&1 | tail -2
