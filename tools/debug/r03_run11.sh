#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run11.log) 2>&1
echo "== parity with VCLA_GEMM_PF=1"; VCLA_GEMM_PF=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm and not dstream and not gemv" 2>&1 | tail -5
for v in 0 1; do
  echo "== VCLA_GEMM_PF=$v: ViT + LLaMA prefill GEMM shapes"
  VCLA_GEMM_PF=$v timeout 300 python tools/bench_kernels.py vit 2>&1 | grep -E "auto|256 " | grep -v amdgpu
  VCLA_GEMM_PF=$v timeout 300 python tools/bench_kernels.py gemm 2>&1 | grep -v amdgpu | head -14
done
echo "== done"
