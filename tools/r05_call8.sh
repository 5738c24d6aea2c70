#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
echo "== attention tests"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "attention_vit or attention_mfma" 2>&1 | tail -8
echo "== W8A8 layer test + 336px vision"; timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -k "fp8_mfma_prefill_matches or 336px or set_image_size" 2>&1 | tail -12
grep -E "W8A8" gpurun_out/parity_report.txt | cut -c1-400
echo "== attention microbench"; timeout 300 python tools/bench_kernels.py attn577 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_attn577.txt
echo "== bench config4 share (B=32, 336px, fp8)"; timeout 600 python bench.py --fp8 --image-size 336 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05_c4_b32.json 2> gpurun_out/r05_c4_b32.err; echo rc=$?
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05_c4_b32.json") if l.startswith("{")][-1]); print(d["value"], d.get("images_per_sec"), d.get("images_per_sec_prefill"), d["breakdown_ms"])
PY
