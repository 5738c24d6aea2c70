cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "fp8" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "fp8" 2>&1 | tail -8
grep -n "fp8_mfma\|fp8 MFMA" gpurun_out/parity_report.txt | cut -c1-260 | tail -12
