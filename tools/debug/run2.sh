cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "dstream or rmsnorm_pack or fragment_major" 2>&1 | tail -5
timeout 600 python tools/bench_kernels.py dstream 2>&1 | tee gpurun_out/r02_dstream_microbench.txt | tail -30
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof64n -o b64 -- python $GRAFT_REPO_ROOT/bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof64n.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof64n -name "*kernel_stats.csv" | head -1); head -25 "$f" | cut -c1-160
find gpurun_out/prof64n -name "*.csv" -size +3M -delete
