#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run16.log) 2>&1
echo "== tests: dstream + set_image_size + env switches"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "dstream" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -x -k "set_image_size or prefix_allowed" 2>&1 | tail -3
for v in 0 1 0 1; do
  echo "== VCLA_GEMM_PF=$v bench B=64 (vision / prefill)"
  VCLA_GEMM_PF=$v timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['breakdown_ms'])"
done
echo "== default bench (with config4 leg), no cpu"
timeout 900 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/r03_bench_try.json; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_try.json')); print(d['value'], d['breakdown_ms']); print(d['config2']); print(d.get('config4'))"
echo "== done"
