#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run34.log) 2>&1
echo "== GEMM parity"
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm and not dstream and not gemv" 2>&1 | tail -4
echo "== timeline, fast epilogue"
VCLA_LIB=$PWD/tools/libvcla_tl_epi0.so timeout 300 python tools/debug/gemm256_timeline.py 0 2>&1 | grep -v amdgpu | cut -c1-230 | grep -E "^==|round 0"
echo "== microbench"
timeout 300 python tools/bench_kernels.py vit 2>&1 | grep -E "auto" | grep -v amdgpu
timeout 300 python tools/bench_kernels.py gemm 2>&1 | grep -E "llama (qkv|o|gate-up swiglu|down) " | head -4
echo "== VCLA_GEMM_EPI_LDS=1 microbench"
VCLA_GEMM_EPI_LDS=1 timeout 300 python tools/bench_kernels.py vit 2>&1 | grep -E "auto" | grep -v amdgpu
for rep in 1 2; do
  echo "== bench B=64"
  timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"
done
echo "== done"
