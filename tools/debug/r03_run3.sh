#!/bin/bash
# round 3, GPU call 3: tr-read V in the MFMA attention, decode attention with both batches up front, ViT tail kernel A/B, bench
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run3.log) 2>&1
echo "== kernel parity (attention, streaming GEMM)"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "attention or attn or dstream or deferred" 2>&1 | tail -8
echo "== vit attention"; timeout 300 python tools/bench_kernels.py vitattn attndec vittail 2>&1 | grep -v amdgpu.ids
echo "== vit GEMMs, tail on the skinny kernel"; VCLA_TAIL_KERNEL=7 timeout 300 python tools/bench_kernels.py vit 2>&1 | grep -v amdgpu.ids | grep auto
echo "== vit GEMMs, tail on panel + reduce"; timeout 300 python tools/bench_kernels.py vit 2>&1 | grep -v amdgpu.ids | grep auto
echo "== model parity (7B + small)"; timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -6 | cut -c1-300
echo "== bench"; timeout 900 python bench.py --steps 3 --warmup 1 --steps-b64 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r03_run3_bench.json | cut -c1-1800
echo "== bench, skinny tails"; VCLA_TAIL_KERNEL=7 timeout 900 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r03_run3_bench_b64_tail7.json | cut -c1-1200
echo "== done"
