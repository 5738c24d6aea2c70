cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "dstream" 2>&1 | tail -3
VCLA_BENCH_MS=64 VCLA_DS_ROT=1 timeout 600 python tools/bench_kernels.py dstream 2>&1 | grep "M=" | tee gpurun_out/r02_dstream_rot1.txt
VCLA_BENCH_MS=64 VCLA_DS_ROT=0 timeout 600 python tools/bench_kernels.py dstream 2>&1 | grep "M=" | tee gpurun_out/r02_dstream_rot0.txt
