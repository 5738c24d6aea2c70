cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/bench_kernels.py fp8mfma 2>&1 | grep -v amdgpu | tee gpurun_out/r02_fp8mfma_microbench.txt
timeout 600 python bench.py --fp8 --batch 64 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r02_bench_fp8_b64.json | cut -c1-1200
