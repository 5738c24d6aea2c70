cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcg_$c
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcg_$c -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_gemv.py 2>&1 | tail -1)
  f=$(find gpurun_out/pmcg_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" $c gemv1p_kernel | tee gpurun_out/r02_pmc_gemv1p_$(echo $c | tr A-Z a-z).txt
  rm -rf gpurun_out/pmcg_$c
done
