#!/bin/bash
# round 3, GPU call 2: DS2 with the full ring, 16-wave o/down variant, MFMA attention (double buffer + remainder key), tr16 probe
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run2.log) 2>&1
echo "== tr16 probe"; timeout 120 python tools/debug/probe_tr16.py 2>&1 | grep -v amdgpu.ids
echo "== kernel parity (attention, streaming GEMM)"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "attention or attn or dstream or deferred" 2>&1 | tail -8
echo "== W16 parity"; VCLA_DS2=0 VCLA_DS_W16=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "dstream or deferred" 2>&1 | tail -5
echo "== dstream variants (M=64)"
echo "-- VCLA_DS2=1 (ring 144 KiB, 60 DMA in flight)"; VCLA_DS2=1 VCLA_BENCH_MS=64 timeout 300 python tools/bench_kernels.py dstream 2>&1 | grep -v amdgpu.ids | grep -v "fp8\b" | cut -c1-110
echo "-- VCLA_DS2=0 VCLA_DS_W16=1"; VCLA_DS2=0 VCLA_DS_W16=1 VCLA_BENCH_MS=64 timeout 300 python tools/bench_kernels.py dstream 2>&1 | grep -v amdgpu.ids | cut -c1-110
echo "== vit attention A/B"
for v in 0 1; do echo "-- VCLA_ATTN_MFMA_DB=$v"; VCLA_ATTN_MFMA_DB=$v timeout 300 python tools/bench_kernels.py vitattn 2>&1 | grep -v amdgpu.ids; done
echo "== bench B=64 (DS2=0, flash decode attention)"; VCLA_DS2=0 timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1500 | tee gpurun_out/r03_run2_bench_b64.json
echo "== bench B=64 (DS2=0, W16)"; VCLA_DS2=0 VCLA_DS_W16=1 VCLA_DS_SPLITK_O=1 timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1500 | tee gpurun_out/r03_run2_bench_b64_w16.json
echo "== done"
