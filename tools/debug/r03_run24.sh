#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run24.log) 2>&1
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm or rope_kv" 2>&1 | tail -4
timeout 600 python tools/bench_kernels.py vit 2>&1 | grep -E "auto" | grep -v amdgpu
echo "== done"
