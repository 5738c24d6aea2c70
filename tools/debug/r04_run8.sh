cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -8 | tee gpurun_out/r04g_gpu_tests.txt
cp gpurun_out/parity_report.txt gpurun_out/r04g_parity_report.txt
