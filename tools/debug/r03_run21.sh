#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run21.log) 2>&1
echo "== kernel tests: decode attention (bf16 + fp8 cache), rope append"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "attn_decode or rope_kv" 2>&1 | tail -5
for kv in 0 1; do
echo "== decode attention microbench VCLA_BENCH_KV8=$kv"
VCLA_BENCH_KV8=$kv timeout 300 python tools/bench_kernels.py attndec 2>&1 | grep -E "^attndec"
done
for kv in 0 1; do
echo "== bench --fp8 --batch 64 --fp8-kv $kv"
timeout 600 python bench.py --fp8 --fp8-kv $kv --batch 64 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"
echo "== bench --fp8 --image-size 336 --batch 32 --fp8-kv $kv"
timeout 600 python bench.py --fp8 --fp8-kv $kv --image-size 336 --batch 32 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"
echo "== bench --fp8 (B=1) --fp8-kv $kv"
timeout 600 python bench.py --fp8 --fp8-kv $kv --steps 3 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['breakdown_ms'])"
done
echo "== done"
