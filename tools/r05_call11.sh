#!/bin/bash
mkdir -p gpurun_out
echo "== ring parity (bf16, fp8, slab)"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "ring" 2>&1 | tail -6
echo "== ring microbench"; VCLA_BENCH_MS=256 VCLA_BENCH_FKS=1,11 timeout 400 python tools/bench_kernels.py ring 2>&1 | grep -v amdgpu.ids | grep -v slab-major | tee gpurun_out/r05_ring_microbench2.txt
echo "== 7B fp8 B=256 parity"; timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -k "fp8_kernels_match and 256" 2>&1 | tail -4
echo "== bench gb256 fp8 336"; timeout 600 python bench.py --gpus 1 --fp8 --image-size 336 --global-batch 256 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05_bench_strong_gb256_fp8_336.json 2> gpurun_out/r05_c4.err; echo rc=$?
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05_bench_strong_gb256_fp8_336.json") if l.startswith("{")][-1]); print(d["value"], d.get("images_per_sec"), d.get("images_per_sec_prefill"), d["breakdown_ms"])
PY
./tools/l2_intake f > gpurun_out/r05_l2_intake_full.txt 2>&1; tail -3 gpurun_out/r05_l2_intake_full.txt
