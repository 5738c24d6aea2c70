#!/bin/bash
# copy the judged summaries of the last `tools/gpu_check.sh [full|pmc]` run from gpurun_out/ (scratch) into profiles/ (tracked)
# usage: bash tools/collect_profiles.sh r01 run18
r=${1:-r01}; tag=${2:-run}
o=gpurun_out; p=profiles
mkdir -p $p
grep '^{' $o/bench.log        | tail -1 > $p/${r}_bench_${tag}.json
grep '^{' $o/bench_fp8.log    | tail -1 > $p/${r}_bench_fp8_${tag}.json
grep '^{' $o/bench_sample.log | tail -1 > $p/${r}_bench_sample_${tag}.json
grep '^{' $o/bench_b64.log    | tail -1 > $p/${r}_bench_b64_${tag}.json
f=$(find $o/prof -name "*kernel_stats.csv" | head -1);   [ -n "$f" ] && cp $f $p/${r}_bench_b1_kernel_stats.csv
f=$(find $o/prof64 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $p/${r}_bench_b64_kernel_stats.csv
[ -f $o/kernels.log ] && cp $o/kernels.log $p/${r}_kernel_microbench_${tag}_vit.txt
[ -f $o/parity_report.txt ] && cp $o/parity_report.txt $p/${r}_parity_report_${tag}.txt
[ -f $o/pmc_FETCH_SIZE.txt ] && cp $o/pmc_FETCH_SIZE.txt $p/${r}_pmc_gemv1_fetch_size.txt
[ -f $o/pmc_WRITE_SIZE.txt ] && cp $o/pmc_WRITE_SIZE.txt $p/${r}_pmc_gemv1_write_size.txt
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES; do [ -f $o/pmcg_$c.txt ] && cat $o/pmcg_$c.txt; done > $p/${r}_pmc_gemm256_mfma.txt
grep -E "passed|failed|smoke ok" $o/gpu_check.log > $p/${r}_gpu_tests_${tag}.txt
ls -la $p | tail -30
