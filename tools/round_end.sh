#!/bin/bash
# Round-end measurement set (one gpurun call): full GPU test suite, smoke, the bench lines of every configuration, rocprofv3
# kernel stats of the two main bench commands (grouped by (kernel, grid)), PMC passes (separate, counters only), kernel
# microbenchmarks.  Artifacts: gpurun_out/<tag>_*; copy the ones to be judged into profiles/.   usage: bash tools/round_end.sh r03
tag=${1:-rXX}
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
echo "== tests"; timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/${tag}_gpu_tests.txt
cp gpurun_out/parity_report.txt gpurun_out/${tag}_parity_report.txt 2>/dev/null
echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 | tee -a gpurun_out/${tag}_gpu_tests.txt
echo "== bench (default)"; timeout 900 python bench.py --steps 5 --warmup 2 2>&1 | tail -1 > gpurun_out/${tag}_bench.json; cut -c1-400 gpurun_out/${tag}_bench.json
echo "== bench B=64"; timeout 900 python bench.py --batch 64 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${tag}_bench_b64.json; cut -c1-300 gpurun_out/${tag}_bench_b64.json
echo "== bench fp8"; timeout 600 python bench.py --fp8 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${tag}_bench_fp8.json; cut -c1-200 gpurun_out/${tag}_bench_fp8.json
echo "== bench fp8 B=64"; timeout 900 python bench.py --fp8 --batch 64 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${tag}_bench_fp8_b64.json; cut -c1-200 gpurun_out/${tag}_bench_fp8_b64.json
echo "== bench fp8 336px B=32 (per-GPU share of configs[4])"; timeout 900 python bench.py --fp8 --image-size 336 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${tag}_bench_336px_fp8_b32.json; cut -c1-200 gpurun_out/${tag}_bench_336px_fp8_b32.json
echo "== bench sampled"; timeout 600 python bench.py --sample --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${tag}_bench_sample.json; cut -c1-200 gpurun_out/${tag}_bench_sample.json
echo "== bench strong scaling mode on ONE GPU: global batch 256 (the N = 1 leg of north_star's '>= 6x images/sec 1 -> 8 GPUs at batch 256')"; timeout 900 python bench.py --gpus 1 --global-batch 256 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${tag}_bench_strong_gb256.json; cut -c1-300 gpurun_out/${tag}_bench_strong_gb256.json
echo "== bench B=128"; timeout 900 python bench.py --batch 128 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${tag}_bench_b128.json; cut -c1-200 gpurun_out/${tag}_bench_b128.json
for cfg in "b1:--steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --no-cpu-baseline" "b64:--batch 64 --steps 1 --warmup 1 --steps-c4 0 --no-cpu-baseline" "gb256:--global-batch 256 --steps 1 --warmup 1 --no-cpu-baseline"; do
  nm=${cfg%%:*}; args=${cfg#*:}
  echo "== rocprofv3 --kernel-trace --stats: bench.py $args"
  rm -rf gpurun_out/prof_$nm
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$nm -o bench -- python $R/bench.py $args 2>&1 | tail -1 | cut -c1-200)
  f=$(find gpurun_out/prof_$nm -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_bench_${nm}_kernel_stats.csv && head -12 $f | cut -c1-160
  f=$(find gpurun_out/prof_$nm -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/prof_by_grid.py $f 24 | tee gpurun_out/${tag}_bench_${nm}_by_grid.txt
  rm -rf gpurun_out/prof_$nm
done
echo "== PMC passes (counters only, one counter per run)"
pmc() { # counter workload-script kernel-filter outfile [env]
  rm -rf gpurun_out/pmcx
  (cd /tmp && export TMPDIR=/tmp && env $5 timeout 600 rocprofv3 --pmc $1 --output-format csv -d $R/gpurun_out/pmcx -o pmc -- python $R/$2 2>&1 | tail -1 | cut -c1-80)
  f=$(find gpurun_out/pmcx -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" $1 $3 | tee -a gpurun_out/$4
  rm -rf gpurun_out/pmcx
}
rm -f gpurun_out/${tag}_pmc_*.txt
for c in FETCH_SIZE WRITE_SIZE; do pmc $c tools/pmc_engine.py decode_engine_kernel ${tag}_pmc_engine_$(echo $c | tr A-Z a-z).txt; done
for c in FETCH_SIZE WRITE_SIZE; do pmc $c tools/pmc_gemv.py gemv1p_kernel ${tag}_pmc_gemv1p_$(echo $c | tr A-Z a-z).txt; done
for c in FETCH_SIZE WRITE_SIZE; do pmc $c tools/pmc_dstream.py gemm_dstream_kernel ${tag}_pmc_dstream_$(echo $c | tr A-Z a-z).txt; done
for c in FETCH_SIZE WRITE_SIZE; do pmc $c tools/pmc_attn_decode.py attn_decode_flash_kernel ${tag}_pmc_attn_decode_b64_$(echo $c | tr A-Z a-z).txt; done
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES; do pmc $c tools/pmc_gemm.py gemm_mfma256_kernel ${tag}_pmc_vit_fc1_mfma.txt VCLA_PMC_SHAPE=vit; done
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE FETCH_SIZE WRITE_SIZE; do pmc $c tools/pmc_vit_attn.py attn_vit_dma_kernel ${tag}_pmc_vit_attn.txt; done
echo "== engine A/B + timeline (full 7B)"; timeout 900 python tools/engine_probe.py --layers 32 --vocab 49958 --steps 3 --time 64 --timeline 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${tag}_engine_vs_launches.txt
echo "== bench with the engine off (VCLA_ENGINE=0: the round-5 launch path, same box)"; VCLA_ENGINE=0 timeout 600 python bench.py --steps 3 --warmup 1 --steps-b64 0 --steps-c4 0 --steps-strong 0 --steps-strong-c4 0 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${tag}_bench_engine_off.json; cut -c1-300 gpurun_out/${tag}_bench_engine_off.json
echo "== B = 1 time to first token (one image, T = 128): wall clock, then the kernels of one request (rocprofv3 --kernel-trace of the same script)"
(timeout 300 python tools/prof_ttft.py 20 2>&1 | grep "^B="; echo "-- VCLA_RING_VIT=0 (round-5 dispatch of the one-image ViT / resampler GEMMs)"; VCLA_RING_VIT=0 timeout 300 python tools/prof_ttft.py 20 2>&1 | grep "^B=") | tee gpurun_out/${tag}_ttft_b1.txt
rm -rf gpurun_out/prof_ttft
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_ttft -o t -- python $R/tools/prof_ttft.py 10 2>&1 | grep "^B=")
f=$(find gpurun_out/prof_ttft -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/prof_by_grid.py $f 40 | grep -v "at::native" | tee -a gpurun_out/${tag}_ttft_b1.txt
rm -rf gpurun_out/prof_ttft
echo "== microbench"
(python tools/bench_kernels.py gemv1 2>&1 | grep "^gemv1"; VCLA_BENCH_MS=64 python tools/bench_kernels.py dstream 2>&1 | grep -E "^M=|^=="; python tools/bench_kernels.py dec256 2>&1 | grep -E "^M=|^=="; python tools/bench_kernels.py vit vittail vitattn attndec vit1 2>&1 | grep -E "^vit|^attn"; echo "-- sustained (400 launches per figure)"; VCLA_BENCH_REPS=400 python tools/bench_kernels.py vit 2>&1 | grep "^vit") | tee gpurun_out/${tag}_kernel_microbench.txt
echo "== done"
