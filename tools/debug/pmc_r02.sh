cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
pmc() { # counter workload-script kernel-filter outfile [env]
  rm -rf gpurun_out/pmcx
  (cd /tmp && export TMPDIR=/tmp && env $5 timeout 600 rocprofv3 --pmc $1 --output-format csv -d $R/gpurun_out/pmcx -o pmc -- python $R/$2 2>&1 | tail -1 | cut -c1-80)
  f=$(find gpurun_out/pmcx -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" $1 $3 | tee -a gpurun_out/$4
  rm -rf gpurun_out/pmcx
}
rm -f gpurun_out/r02_pmc_*.txt
for c in FETCH_SIZE WRITE_SIZE; do pmc $c tools/pmc_dstream.py gemm_dstream_kernel r02_pmc_dstream_$(echo $c | tr A-Z a-z).txt; done
for c in FETCH_SIZE WRITE_SIZE; do pmc $c tools/pmc_attn_decode.py attn_decode_kernel r02_pmc_attn_decode_b64_$(echo $c | tr A-Z a-z).txt; done
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES; do pmc $c tools/pmc_gemm.py gemm_mfma256_kernel r02_pmc_vit_fc1_mfma.txt VCLA_PMC_SHAPE=vit; done
