#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run19.log) 2>&1
for nb in 4 1; do for fl in 1 0; do
echo "== VCLA_BENCH_NBUF=$nb VCLA_ATTN_FLASH=$fl"
VCLA_BENCH_NBUF=$nb VCLA_ATTN_FLASH=$fl timeout 300 python tools/bench_kernels.py attndec 2>&1 | grep -E "^attndec B= (64|32)"
done; done
echo "== done"
