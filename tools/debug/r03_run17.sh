#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run17.log) 2>&1
echo "== GEMM parity (PF default, 257-row tiles)"
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm and not dstream and not gemv" 2>&1 | tail -5
for v in 1 0; do
  echo "== VCLA_GEMM_XR=$v: ViT GEMM microbench + B=64 bench"
  VCLA_GEMM_XR=$v timeout 300 python tools/bench_kernels.py vit 2>&1 | grep -E "auto" | grep -v amdgpu
  VCLA_GEMM_XR=$v timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"
done
echo "== model parity (7B taps etc.)"
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -5
echo "== done"
