#!/bin/bash
# round 5, GPU call 2: ring kernel parity + microbench, new edge-case tests, B = 256 model test, gb256 bench lines
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
echo "== ring kernel + loss tests"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "ring or causal_lm_loss" 2>&1 | tail -15
echo "== ring microbench"; VCLA_BENCH_MS=256,160 timeout 400 python tools/bench_kernels.py ring 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_ring_microbench.txt
echo "== model tests (edge cases, interior masks, B=256 rows)"; timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -k "edge or interior or batch_rows or fp8" 2>&1 | tail -15
echo "== bench gb256 bf16"; timeout 600 python bench.py --gpus 1 --global-batch 256 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r05_gb256_bf16.json 2> gpurun_out/r05_gb256_bf16.err; echo rc=$?; python - <<'PY'
import json
for f in ("gpurun_out/r05_gb256_bf16.json",):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f, d["value"], d.get("images_per_sec"), d["breakdown_ms"])
    except Exception as e: print(f, "failed", e)
PY
echo "== bench gb256 fp8 336"; timeout 600 python bench.py --gpus 1 --fp8 --image-size 336 --global-batch 256 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r05_c4_gb256_after.json 2> gpurun_out/r05_c4_gb256_after.err; echo rc=$?; python - <<'PY'
import json
for f in ("gpurun_out/r05_c4_gb256_after.json",):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f, d["value"], d.get("images_per_sec"), d["breakdown_ms"])
    except Exception as e: print(f, "failed", e)
PY
tail -3 gpurun_out/*.err
