#!/bin/bash
# rocprofv3 in-model tables (B = 1, B = 64) of the current build
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run6.log) 2>&1
for cfg in "b1:--steps 2 --warmup 1 --steps-b64 0 --no-cpu-baseline" "b64:--batch 64 --steps 1 --warmup 1 --no-cpu-baseline"; do
  nm=${cfg%%:*}; args=${cfg#*:}
  echo "== rocprofv3 --kernel-trace --stats: bench.py $args"
  rm -rf gpurun_out/prof_$nm
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$nm -o bench -- python $R/bench.py $args 2>&1 | tail -1 | cut -c1-200)
  f=$(find gpurun_out/prof_$nm -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/prof_by_grid.py $f 22 | tee gpurun_out/r03mid_bench_${nm}_by_grid.txt
  rm -rf gpurun_out/prof_$nm
done
echo "== done"
