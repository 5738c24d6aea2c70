cd $GRAFT_REPO_ROOT
for k in 0 2048 8192; do VCLA_TAIL_K=$k python tools/prof_vision.py 64 8 2>&1 | tail -1 | awk -v k=$k '{print "tailK" k, $0}'; done
for k in 0 2048; do VCLA_TAIL_K=$k python tools/prof_vision.py 64 8 2>&1 | tail -1 | awk -v k=$k '{print "tailK" k, $0}'; done
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_shapes or gemm_bf16 or ragged" 2>&1 | tail -2
