#!/bin/bash
# round 5: the whole GPU suite + smoke + the bench lines + rocprofv3 per-kernel tables (one gpurun call)
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$PWD}
rm -f gpurun_out/parity_report.txt
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -12 | tee gpurun_out/r05_gpu_tests.txt
echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== bench default"; timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; echo rc=$?; tail -c 600 gpurun_out/r05_bench.json
by_grid() {  # $1 = tag, rest = bench args
  tag=$1; shift
  rm -rf gpurun_out/prof_$tag
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o bench -- python $R/bench.py "$@" --no-cpu-baseline > $R/gpurun_out/r05_prof_$tag.json 2> /dev/null)
  f=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/prof_by_grid.py "$f" 30 > gpurun_out/r05_bench_${tag}_by_grid.txt
  s=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$s" ] && head -40 "$s" > gpurun_out/r05_bench_${tag}_kernel_stats.csv
  rm -rf gpurun_out/prof_$tag
  head -12 gpurun_out/r05_bench_${tag}_by_grid.txt
}
echo "== rocprof B=1"; by_grid b1 --steps 2 --warmup 1 --steps-b64 0 --steps-c4 0 --steps-strong 0
echo "== rocprof B=64"; by_grid b64 --batch 64 --steps 1 --warmup 1
echo "== rocprof gb256"; by_grid gb256 --global-batch 256 --steps 1 --warmup 1
echo "== rocprof config4 share"; by_grid fp8_336px_b32 --fp8 --image-size 336 --batch 32 --steps 1 --warmup 1
echo "== done"
