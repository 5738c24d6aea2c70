cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "dstream or deferred" 2>&1 | grep -E "^FAILED|Error|assert" | head -5
rm -rf gpurun_out/pg
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pg -o t -- python $R/bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --new-tokens 16 2>&1 | tail -1 | cut -c1-100)
f=$(find gpurun_out/pg -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/prof_by_grid.py $f 14
rm -rf gpurun_out/pg
