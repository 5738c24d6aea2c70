cd $GRAFT_REPO_ROOT
python tools/pmc_attn_decode.py 2>&1 | tail -1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmca_$c
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmca_$c -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_attn_decode.py 2>&1 | tail -1)
  f=$(find gpurun_out/pmca_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" $c attn_decode_kernel | tee gpurun_out/r02_pmc_attn_decode_$(echo $c | tr A-Z a-z).txt
  rm -rf gpurun_out/pmca_$c
done
