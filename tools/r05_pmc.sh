#!/bin/bash
# round 5 PMC passes (counters only, one per run): HBM traffic of the ring gate/up GEMM, MFMA-busy of the ring GEMM and of the 577-token ViT attention
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$PWD}
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rm -rf gpurun_out/pmc5_$c
  (cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc5_$c -o pmc -- python $R/tools/pmc_ring.py 2>&1 | tail -1)
  f=$(find gpurun_out/pmc5_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python tools/pmc_summary.py "$f" $c gemm_ring_kernel | sed 's/^/ring gate-up M=256: /' | tee -a gpurun_out/r05_pmc_ring.txt
    python tools/pmc_summary.py "$f" $c attn_vit_long_kernel | sed 's/^/attn_vit_long B=32: /' | tee -a gpurun_out/r05_pmc_ring.txt
  fi
  rm -rf gpurun_out/pmc5_$c
done
