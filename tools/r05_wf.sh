#!/bin/bash
# round 5: ring kernel with fragment-major weight pieces -- parity, microbench, and the B = 256 decode leg with / without
mkdir -p gpurun_out
echo "== ring tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "ring" 2>&1 | tail -5
echo "== microbench"; VCLA_BENCH_MS=256,192 timeout 400 python tools/bench_kernels.py ringwf 2>&1 | tee gpurun_out/r05_ringwf.txt
for wf in 0 1; do
  echo "== gb256 VCLA_RING_WF=$wf"; VCLA_RING_WF=$wf timeout 400 python bench.py --global-batch 256 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r05_gb256_wf$wf.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('breakdown_ms'))"
done
echo "== model tests at 129-256 rows"; timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -k "batch_rows or beam or fp8_kernels" 2>&1 | tail -5
echo "== done"
