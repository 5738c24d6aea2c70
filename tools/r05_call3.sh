#!/bin/bash
mkdir -p gpurun_out
VCLA_LIB=$PWD/tools/libvcla_ringx.so timeout 600 python tools/bench_kernels.py ringx 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_ringx.txt
