cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
echo "== new tests"
timeout 1500 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "edge or fp32_mode or rccl or saturate or golden or fp8_cache or decode_fused" 2>&1 | tail -15
cp gpurun_out/parity_report.txt gpurun_out/r04a_parity_report.txt
echo "== bench global-batch 256"
timeout 900 python bench.py --global-batch 256 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-1500 | tee gpurun_out/r04a_gb256.txt
echo "== bench B=128"
timeout 900 python bench.py --batch 128 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-1200 | tee gpurun_out/r04a_b128.txt
echo "== rocprof gb256"
R=$GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_256 -o bench -- python $R/bench.py --global-batch 256 --steps 1 --warmup 1 --no-cpu-baseline --new-tokens 16 2>&1 | tail -1 | cut -c1-300)
f=$(find gpurun_out/prof_256 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/prof_by_grid.py $f 40 | tee gpurun_out/r04a_gb256_by_grid.txt
rm -rf gpurun_out/prof_256
