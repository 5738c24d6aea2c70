#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run25.log) 2>&1
for rep in 1 2; do
for lib in product pffull; do
  if [ $lib = product ]; then unset VCLA_LIB; else export VCLA_LIB=$PWD/tools/libvcla_pffull.so; fi
  echo "== $lib: B=64 bench"
  timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"
done; done
for lib in product pffull; do
  if [ $lib = product ]; then unset VCLA_LIB; else export VCLA_LIB=$PWD/tools/libvcla_pffull.so; fi
  echo "== $lib: microbench"
  timeout 300 python tools/bench_kernels.py vit 2>&1 | grep -E "auto" | grep -v amdgpu
  timeout 300 python tools/bench_kernels.py gemm 2>&1 | grep -E "llama (qkv|o|gate-up swiglu|down) " | head -4
done
echo "== done"
