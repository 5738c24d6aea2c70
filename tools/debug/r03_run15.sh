#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run15.log) 2>&1
echo "== product .so (with the seam code)"
VCLA_BENCH_MS=64 timeout 600 python tools/bench_kernels.py dstream 2>&1 | grep -E "^M=" | cut -c1-150
echo "== tools/libvcla_oldds.so (streaming GEMM of commit 7079b1d, before the seam)"
VCLA_LIB=$PWD/tools/libvcla_oldds.so VCLA_BENCH_MS=64 timeout 600 python tools/bench_kernels.py dstream 2>&1 | grep -E "^M=" | cut -c1-150
echo "== product again"
VCLA_BENCH_MS=64 timeout 600 python tools/bench_kernels.py dstream 2>&1 | grep -E "^M=" | cut -c1-150
echo "== done"
