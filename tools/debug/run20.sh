cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -3
for pg in 1 0; do echo "VCLA_GEMM_PERSIST=$pg"; VCLA_GEMM_PERSIST=$pg timeout 600 python tools/bench_kernels.py vit 2>&1 | grep "auto\|256 " ; VCLA_GEMM_PERSIST=$pg timeout 600 python tools/prof_vision.py 64 5 2>&1 | grep "vision stack"; VCLA_GEMM_PERSIST=$pg timeout 600 python tools/bench_kernels.py fp8mfma 2>&1 | grep "llama" | head -4; done
