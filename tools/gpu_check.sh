#!/bin/bash
# One gpurun call = the whole measurement loop: env info, per-kernel parity, model parity, bench, rocprof stats.
# Usage (on the GPU box, from the repo root): bash tools/gpu_check.sh [quick|full]
mode=${1:-full}
mkdir -p gpurun_out
exec > >(tee gpurun_out/gpu_check.log) 2>&1
echo "== host"; nproc; free -g | head -2; rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8
python -c "import torch;print(torch.__version__, torch.cuda.device_count(), torch.cuda.get_device_name(0))"
rm -f gpurun_out/parity_report.txt
echo "== kernel parity"; timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -60
echo "== model parity"; timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_dropin.py tests/test_gpu_preprocess.py tests/test_gpu_sampling.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/model_parity.log 2>&1; tail -40 gpurun_out/model_parity.log | cut -c1-300
if [ "$mode" != "quick" ]; then
  echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -5
  echo "== bench"; timeout 900 python bench.py --steps 3 --warmup 1 2>&1 | tail -5 | tee gpurun_out/bench.log
  echo "== bench fp8 B=1"; timeout 600 python bench.py --fp8 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_fp8.log
  echo "== bench sampled B=1"; timeout 600 python bench.py --sample --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_sample.log
  echo "== bench B=64"; timeout 600 python bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 | tee gpurun_out/bench_b64.log
  echo "== kernel microbench"; (timeout 900 python tools/bench_kernels.py gemv1 vit 2>&1; VCLA_BENCH_MS=64 timeout 600 python tools/bench_kernels.py dstream 2>&1) | grep -v amdgpu.ids | tee gpurun_out/kernels.log
  echo "== rocprof B=64"; rm -rf gpurun_out/prof64; (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof64 -o bench -- python $GRAFT_REPO_ROOT/bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1); f=$(find gpurun_out/prof64 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" | cut -c1-200; find gpurun_out/prof64 -name "*kernel_trace.csv" -delete
  echo "== rocprof"; rm -rf gpurun_out/prof; cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -3
  cd $GRAFT_REPO_ROOT; find gpurun_out/prof -name "*.csv" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"; find gpurun_out/prof -name "*kernel_trace.csv" -delete
fi
if [ "$mode" = "pmc" ]; then
  echo "== PMC (HBM traffic of the dominant kernel; separate passes, counters only)"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$c
    (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_gemv.py 2>&1 | tail -2)
    f=$(find gpurun_out/pmc_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" $c gemv1p_kernel | tee gpurun_out/pmc_$c.txt
    find gpurun_out/pmc_$c -name "*.csv" -size +2M -delete
  done
  echo "== PMC MFMA utilisation of the 256x256 GEMM (LLaMA gate/up prefill shape)"
  rocprofv3 -L 2>/dev/null | grep -i -E "MFMA|GRBM_GUI_ACTIVE|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES" | head -20 > gpurun_out/pmc_counters_available.txt; cat gpurun_out/pmc_counters_available.txt | cut -c1-160 | head -12
  for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES; do
    rm -rf gpurun_out/pmcg_$c
    (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcg_$c -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_gemm.py 2>&1 | tail -1)
    f=$(find gpurun_out/pmcg_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" $c gemm_mfma256_kernel | tee gpurun_out/pmcg_$c.txt
    find gpurun_out/pmcg_$c -name "*.csv" -size +2M -delete
  done
fi
echo "== done"
