#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run32.log) 2>&1
for cfg in "0 0" "-175 0" "-175 1" "-350 1" "0 1"; do
set -- $cfg
echo "== VCLA_GEMM_STAGGER=$1 VCLA_GEMM_EPI_LDS=$2"
VCLA_GEMM_STAGGER=$1 VCLA_GEMM_EPI_LDS=$2 timeout 300 python tools/bench_kernels.py vit 2>&1 | grep -E "auto" | grep -v amdgpu
done
echo "== timeline stagger -175 lds 1"
VCLA_GEMM_STAGGER=-175 VCLA_GEMM_EPI_LDS=1 VCLA_LIB=$PWD/tools/libvcla_timeline.so timeout 300 python tools/debug/gemm256_timeline.py 0 2>&1 | grep -v amdgpu | cut -c1-230 | head -12
echo "== timeline stagger -175 lds 0"
VCLA_GEMM_STAGGER=-175 VCLA_GEMM_EPI_LDS=0 VCLA_LIB=$PWD/tools/libvcla_timeline.so timeout 300 python tools/debug/gemm256_timeline.py 0 2>&1 | grep -v amdgpu | cut -c1-230 | head -12
echo "== done"
