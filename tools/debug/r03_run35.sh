#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run35.log) 2>&1
for rep in 1 2; do
for lib in fast general; do
  if [ $lib = fast ]; then unset VCLA_LIB; else export VCLA_LIB=$PWD/tools/libvcla_nofast.so; fi
  echo "== $lib epilogue: B=64 bench"
  timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"
done; done
for lib in fast general; do
  if [ $lib = fast ]; then unset VCLA_LIB; else export VCLA_LIB=$PWD/tools/libvcla_nofast.so; fi
  echo "== $lib epilogue: microbench"
  timeout 300 python tools/bench_kernels.py vit 2>&1 | grep -E "auto" | grep -v amdgpu
  timeout 300 python tools/bench_kernels.py gemm 2>&1 | grep -E "llama (qkv|o|gate-up swiglu|down) |vit fc1  \(B=1\)" | head -5
  echo "-- B=1 bench"
  timeout 600 python bench.py --steps 3 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['breakdown_ms'])"
done
echo "== done"
