#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run18.log) 2>&1
echo "== GEMM parity (tile257 + mfma256)"
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "tile257 or mfma256 or full_size" 2>&1 | tail -3
for v in 1 0 1 0; do
  echo "== VCLA_GEMM_XR=$v: ViT GEMM microbench"
  VCLA_GEMM_XR=$v timeout 300 python tools/bench_kernels.py vit 2>&1 | grep -E "auto" | grep -v amdgpu
done
for v in 1 0; do
  echo "== VCLA_GEMM_XR=$v: B=64 bench"
  VCLA_GEMM_XR=$v timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"
done
echo "== done"
