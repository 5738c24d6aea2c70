#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run31.log) 2>&1
echo "== GEMM parity"
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm and not dstream and not gemv" 2>&1 | tail -4
for st in 0 175; do
echo "== VCLA_GEMM_STAGGER=$st"
VCLA_GEMM_STAGGER=$st timeout 300 python tools/bench_kernels.py vit 2>&1 | grep -E "auto" | grep -v amdgpu
VCLA_GEMM_STAGGER=$st timeout 300 python tools/bench_kernels.py gemm 2>&1 | grep -E "llama (qkv|o|down) " | head -3
done
for st in 0 175 0 175; do
  echo "== VCLA_GEMM_STAGGER=$st bench B=64"
  VCLA_GEMM_STAGGER=$st timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"
done
echo "== done"
