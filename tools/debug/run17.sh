cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcd_$c
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcd_$c -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_dstream.py 2>&1 | tail -1)
  f=$(find gpurun_out/pmcd_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" $c gemm_dstream_kernel | tee gpurun_out/r02_pmc_dstream_$(echo $c | tr A-Z a-z).txt
  rm -rf gpurun_out/pmcd_$c
done
