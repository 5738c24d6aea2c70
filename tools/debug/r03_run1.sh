#!/bin/bash
# round 3, GPU call 1: kernel parity, A/B microbenchmarks of the new decode kernels, model parity, in-model A/B
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.txt
exec > >(tee gpurun_out/r03_run1.log) 2>&1
echo "== kernel parity"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -25
echo "== dstream A/B"
for v in 0 1; do echo "-- VCLA_DS2=$v"; VCLA_DS2=$v VCLA_BENCH_MS=64,32 timeout 300 python tools/bench_kernels.py dstream 2>&1 | grep -v amdgpu.ids; done
echo "== attndec A/B"
for v in 0 1; do echo "-- VCLA_ATTN_FLASH=$v"; VCLA_ATTN_FLASH=$v timeout 300 python tools/bench_kernels.py attndec 2>&1 | grep -v amdgpu.ids; done
echo "== bench new defaults"; timeout 600 python bench.py --steps 2 --warmup 1 --steps-b64 2 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r03_run1_bench_new.json
echo "== bench old kernels"; VCLA_DS2=0 VCLA_ATTN_FLASH=0 timeout 600 python bench.py --steps 2 --warmup 1 --steps-b64 2 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r03_run1_bench_old.json
echo "== bench DS2 only"; VCLA_ATTN_FLASH=0 timeout 600 python bench.py --steps 2 --warmup 1 --steps-b64 2 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r03_run1_bench_ds2.json
echo "== model parity"; timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_dropin.py tests/test_gpu_preprocess.py tests/test_gpu_sampling.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -40 | cut -c1-300
echo "== done"
