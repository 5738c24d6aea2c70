cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_preprocess.py -x -q 2>&1 | tail -3
timeout 600 python tools/bench_preprocess.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02_preprocess_throughput.txt
