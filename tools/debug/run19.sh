cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -4
timeout 600 python tools/bench_kernels.py vit 2>&1 | grep "vit " 
timeout 600 python tools/prof_vision.py 64 5 2>&1 | grep "vision stack"
timeout 600 python tools/bench_kernels.py fp8mfma 2>&1 | grep "llama\|vit" | head -5
