#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run20.log) 2>&1
echo "== kernel tests: decode attention (bf16 + fp8 cache), rope append"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "attn_decode or rope_kv" 2>&1 | tail -15
echo "== model: fp8 kv cache 7B"
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -x -k "fp8_kv_cache or fp8_decode_weights or batch64_decode" 2>&1 | tail -15
grep "fp8 K/V" gpurun_out/parity_report.txt | tail -8
echo "== done"
