#!/bin/bash
mkdir -p gpurun_out
echo "== ring parity"; timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "ring" 2>&1 | tail -5
echo "== ring microbench (areg form = default for 256x96 bf16)"; VCLA_BENCH_MS=256 VCLA_BENCH_FKS=11 timeout 200 python tools/bench_kernels.py ring 2>&1 | grep -v amdgpu.ids | grep -E "bf16 row-major|layer sums" | tee gpurun_out/r05_ring_areg.txt
echo "== all-LDS form"; VCLA_RING_AREG=0 VCLA_BENCH_MS=256 VCLA_BENCH_FKS=11 timeout 200 python tools/bench_kernels.py ring 2>&1 | grep -v amdgpu.ids | grep -E "gate-up  k11 bf16 row-major|qkv      k11 bf16 row" | tee -a gpurun_out/r05_ring_areg.txt
