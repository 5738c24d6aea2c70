#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run27.log) 2>&1
echo "== kernel tests: split-qkv attention, raw partial GEMM"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "attn_decode or raw_partials or rope_kv" 2>&1 | tail -12
echo "== qkv GEMM: unsplit vs raw split"
VCLA_BENCH_MS=64,48 timeout 600 python tools/bench_kernels.py dstream 2>&1 | grep -E "qkv" | cut -c1-170
for qp in 0 1; do
echo "== attention VCLA_BENCH_QP=$qp"
VCLA_BENCH_QP=$qp timeout 300 python tools/bench_kernels.py attndec 2>&1 | grep -E "^attndec B= (64|32)"
done
echo "== done"
