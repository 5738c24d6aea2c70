#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run28.log) 2>&1
for pf in 1 0; do
VCLA_GEMM_PF=$pf VCLA_LIB=$PWD/tools/libvcla_timeline.so timeout 300 python tools/debug/gemm256_timeline.py 0
done
echo "== done"
