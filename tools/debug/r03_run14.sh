#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run14.log) 2>&1
for nb in 4 1; do
echo "== VCLA_BENCH_NB=$nb (1 = same matrix every launch: Infinity-Cache resident)"; VCLA_BENCH_NB=$nb VCLA_BENCH_MS=64 timeout 600 python tools/bench_kernels.py dstream 2>&1 | grep -E "^M="
done
echo "== done"
