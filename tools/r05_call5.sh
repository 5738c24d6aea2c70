#!/bin/bash
mkdir -p gpurun_out
echo "== slab parity"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "slab_major" 2>&1 | tail -8
echo "== ring microbench, slab-major vs row-major"; VCLA_BENCH_MS=256 VCLA_BENCH_FKS=1,11 timeout 400 python tools/bench_kernels.py ring 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_ring_slab_microbench.txt
