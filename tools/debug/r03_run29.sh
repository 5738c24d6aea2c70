#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run29.log) 2>&1
for st in 0 100 175 250; do
echo "== VCLA_GEMM_STAGGER=$st"
VCLA_GEMM_STAGGER=$st timeout 300 python tools/bench_kernels.py vit 2>&1 | grep -E "auto" | grep -v amdgpu
VCLA_GEMM_STAGGER=$st timeout 300 python tools/bench_kernels.py gemm 2>&1 | grep -E "llama (qkv|o|gate-up swiglu|down) " | head -4
done
echo "== timeline with stagger 175"
VCLA_GEMM_STAGGER=175 VCLA_LIB=$PWD/tools/libvcla_timeline.so timeout 300 python tools/debug/gemm256_timeline.py 0 2>&1 | grep -v amdgpu | cut -c1-230
for st in 0 175 0 175; do
  echo "== VCLA_GEMM_STAGGER=$st bench B=64"
  VCLA_GEMM_STAGGER=$st timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"
done
echo "== done"
