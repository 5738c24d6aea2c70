set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "dstream or rmsnorm_pack or fragment_major or attn_decode_fused" 2>&1 | tail -15
timeout 600 python tools/bench_kernels.py dstream 2>&1 | tee gpurun_out/r02_dstream_microbench.txt | tail -30
timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 | tee gpurun_out/r02_b64_a.json
VCLA_DSTREAM=0 timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 | tee gpurun_out/r02_b64_old.json
