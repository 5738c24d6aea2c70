#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run12.log) 2>&1
echo "== split-K parity (two-launch + fused)"; timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "dstream" 2>&1 | tail -8
echo "== model tests"; timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -x -k "batch64 or prefix_allowed or single_token or decode_matches" 2>&1 | tail -8
for v in 0 1; do
  echo "== VCLA_DS_FUSED=$v bench B=64"
  VCLA_DS_FUSED=$v timeout 600 python bench.py --batch 64 --steps 3 --warmup 1 --steps-b64 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['breakdown_ms'])"
done
echo "== done"
