#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run13.log) 2>&1
for v in 0 1; do
echo "== VCLA_DS_SEAM_L2=$v: split-K parity incl. fused"; VCLA_DS_SEAM_L2=$v timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "dstream_splitk" 2>&1 | tail -4
echo "== VCLA_DS_SEAM_L2=$v microbench"; VCLA_DS_SEAM_L2=$v VCLA_BENCH_MS=64 timeout 600 python tools/bench_kernels.py dstream 2>&1 | grep -E "split-K"
done
for env in "VCLA_DS_FUSED=0" "VCLA_DS_FUSED=1" "VCLA_DS_FUSED=1 VCLA_DS_SEAM_L2=1"; do
  echo "== $env bench B=64"
  env $env timeout 600 python bench.py --batch 64 --steps 3 --warmup 1 --steps-b64 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['breakdown_ms'])"
done
echo "== done"
