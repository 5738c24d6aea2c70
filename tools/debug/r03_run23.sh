#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run23.log) 2>&1
echo "== dstream parity"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "dstream or rope_kv" 2>&1 | tail -3
echo "== dstream microbench (lane id laundered before the epilogue)"
VCLA_BENCH_MS=64,32 timeout 600 python tools/bench_kernels.py dstream 2>&1 | grep -E "^M=" | cut -c1-160
echo "== bench B=64"
timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"
echo "== done"
