#!/bin/bash
mkdir -p gpurun_out
echo "== slab256"; timeout 500 python tools/bench_kernels.py slab256 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_slab256_microbench.txt
echo "== ring (graph-timed)"; VCLA_BENCH_MS=256 VCLA_BENCH_FKS=1,11 timeout 400 python tools/bench_kernels.py ring 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_ring_slab_microbench.txt
