#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run40.log) 2>&1
for i in 1 2; do
echo "== kernel parity pass $i (all GEMM families incl. fp8 MFMA and streaming)"
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm" 2>&1 | grep -v amdgpu | tail -5
done
echo "== fp8 benches"
timeout 600 python bench.py --fp8 --batch 64 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"
timeout 600 python bench.py --fp8 --image-size 336 --batch 32 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"
echo "== model parity (fp8 tests)"
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -k "fp8" 2>&1 | grep -v amdgpu | tail -4
echo "== done"
