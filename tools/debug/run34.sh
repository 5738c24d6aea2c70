cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -2
python tools/prof_vision.py 64 5 2>&1 | tail -1
rm -rf gpurun_out/pv
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pv -o t -- python $R/tools/prof_vision.py 64 5 2>&1 | grep "vision stack")
f=$(find gpurun_out/pv -name "*kernel_trace.csv" | head -1)
python tools/prof_by_grid.py $f 30 | grep attn
rm -rf gpurun_out/pv
timeout 600 python bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['config'])"
