cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for d in 1 0; do
  (cd /tmp && export TMPDIR=/tmp && VCLA_DS_DEFER=$d timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof64d$d -o b64 -- python $GRAFT_REPO_ROOT/bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof64d$d.log 2>&1)
  f=$(find gpurun_out/prof64d$d -name "*.db" | head -1)
  python tools/prof_stats.py $f 14 | cut -c1-150 | tee gpurun_out/r02_b64_defer${d}_kernel_stats.csv
  rm -rf gpurun_out/prof64d$d
done
