#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run43.log) 2>&1
echo "== rope / attention kernel tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "rope or attn_decode" 2>&1 | grep -v amdgpu | tail -6
echo "== model tests"
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_dropin.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v amdgpu | tail -4
echo "== bench B=64"
timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"
echo "== bench fp8 B=64"
timeout 600 python bench.py --fp8 --batch 64 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"
echo "== done"
