#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
echo "== 7B fp8 parity vs oracle on dequantised weights"; timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -k "fp8_kernels_match or fp8_mfma_prefill_matches" --durations=5 2>&1 | tail -25
grep -E "W8A16|W8A8" gpurun_out/parity_report.txt | cut -c1-330
