cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.txt
( time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) 2>&1 | tee gpurun_out/r02_gpu_tests.txt
( time timeout 900 python bench.py --steps 5 --warmup 2 2>&1 | tail -1 | tee gpurun_out/r02_bench_default.json | cut -c1-3000 ) 2>&1
