#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run42.log) 2>&1
echo "== model tests at B=64"
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -x -k "batch64 or fp8 or decode_matches" 2>&1 | grep -v amdgpu | tail -4
for v in 1 0 1 0; do
  echo "== VCLA_LMHEAD_PANEL=$v bench B=64"
  VCLA_LMHEAD_PANEL=$v timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"
done
echo "== done"
