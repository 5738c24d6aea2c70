#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run39.log) 2>&1
for i in 1 2 3 4; do
echo "== GEMM parity, pass $i"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm and not dstream and not gemv" 2>&1 | grep -v amdgpu | tail -6
done
echo "== microbench"
timeout 300 python tools/bench_kernels.py vit 2>&1 | grep -E "auto" | grep -v amdgpu
for rep in 1 2; do
for lib in product general; do
  if [ $lib = product ]; then unset VCLA_LIB; else export VCLA_LIB=$PWD/tools/libvcla_nofast.so; fi
  echo "== $lib epilogue: B=64 bench"
  timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"
done; done
echo "== done"
