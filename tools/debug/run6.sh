cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "dstream or fragment" 2>&1 | tail -3
VCLA_BENCH_MS=64 timeout 600 python tools/bench_kernels.py dstream 2>&1 | grep "M=" | tee gpurun_out/r02_dstream_b.txt
timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r02_b64_defer.json | cut -c1-900
VCLA_DS_DEFER=0 timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
