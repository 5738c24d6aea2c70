#!/bin/bash
# round 3, GPU call 4: whole-sequence MFMA attention A/B, macro graphs (vision + prefill), full suites, bench
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.txt
exec > >(tee gpurun_out/r03_run4.log) 2>&1
echo "== vit attention A/B"
for v in 0 1; do echo "-- VCLA_ATTN_MFMA_WHOLE=$v"; VCLA_ATTN_MFMA_WHOLE=$v timeout 300 python tools/bench_kernels.py vitattn 2>&1 | grep -v amdgpu.ids; done
echo "== full GPU suite"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -12 | cut -c1-300
echo "== bench (macro graphs on)"; timeout 900 python bench.py --steps 3 --warmup 2 --steps-b64 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r03_run4_bench.json | cut -c1-1400
echo "== bench (macro graphs off)"; VCLA_MACRO_GRAPH=0 timeout 900 python bench.py --steps 3 --warmup 2 --steps-b64 0 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r03_run4_bench_nomacro.json | cut -c1-900
echo "== done"
