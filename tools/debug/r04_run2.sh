cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
echo "== dec256 microbench (default dispatch)"
python tools/bench_kernels.py dec256 2>&1 | grep -E "^M=|^==" | tee gpurun_out/r04b_dec256.txt
for S in 1 2 3 4 6 8; do echo "-- VCLA_MFMA128_S=$S"; VCLA_BENCH_MS=256 VCLA_MFMA128_S=$S python tools/bench_kernels.py dec256 2>&1 | grep -E "^M=" ; done | tee -a gpurun_out/r04b_dec256.txt
echo "-- 256x256 kernel forced"; VCLA_BENCH_MS=256 VCLA_BENCH_FK=4 python tools/bench_kernels.py dec256 2>&1 | grep -E "^M=" | tee -a gpurun_out/r04b_dec256.txt
echo "== tests"
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider -k "batch_rows or fp32_mode" 2>&1 | tail -5
cp gpurun_out/parity_report.txt gpurun_out/r04b_parity_report.txt
echo "== bench global-batch 256"
timeout 900 python bench.py --global-batch 256 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1100 | tee gpurun_out/r04b_gb256.txt
