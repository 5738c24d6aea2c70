#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run33.log) 2>&1
for ab in 0 1 2; do
echo "== epilogue ablation $ab (0 = full, 1 = no stores, 2 = raw accumulators: no bias / activation)"
VCLA_LIB=$PWD/tools/libvcla_tl_epi$ab.so timeout 300 python tools/debug/gemm256_timeline.py 0 2>&1 | grep -v amdgpu | cut -c1-230 | grep -E "^==|round 0"
done
echo "== stagger within the XCD (-175) and by XCD (175), full kernel"
for st in -175 175; do
VCLA_GEMM_STAGGER=$st VCLA_LIB=$PWD/tools/libvcla_tl_epi0.so timeout 300 python tools/debug/gemm256_timeline.py 0 2>&1 | grep -v amdgpu | cut -c1-230 | grep -E "^== vit (qkv|fc1:)|round 1" | head -4
done
echo "== done"
