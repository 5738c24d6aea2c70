cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -k "vision or tiny or small" 2>&1 | tail -2
VCLA_ATTN_MFMA_NW=4 python tools/prof_vision.py 64 5 2>&1 | tail -1
python tools/prof_vision.py 64 5 2>&1 | tail -1
rm -rf gpurun_out/pv
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pv -o t -- python $R/tools/prof_vision.py 64 5 2>&1 | grep "vision stack")
f=$(find gpurun_out/pv -name "*kernel_trace.csv" | head -1)
python tools/prof_by_grid.py $f 12
rm -rf gpurun_out/pv
