#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run36.log) 2>&1
echo "== GEMM parity"
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm and not dstream and not gemv" 2>&1 | tail -4
echo "== microbench (bias preloaded before the K loop, residual loads hoisted)"
timeout 300 python tools/bench_kernels.py vit 2>&1 | grep -E "auto" | grep -v amdgpu
timeout 300 python tools/bench_kernels.py gemm 2>&1 | grep -E "llama (qkv|o|gate-up swiglu|down) " | head -4
for rep in 1 2; do
for lib in product general; do
  if [ $lib = product ]; then unset VCLA_LIB; else export VCLA_LIB=$PWD/tools/libvcla_nofast.so; fi
  echo "== $lib epilogue: B=64 bench"
  timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"
done; done
unset VCLA_LIB
echo "== model parity"
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4
echo "== done"
