#!/bin/bash
# PMC passes over the M=64 panel microbench (legacy grid); one counter group per pass
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|name)|^gpu|Counter_Name" | head -5
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters_all.txt 2>&1
grep -c . $GRAFT_REPO_ROOT/gpurun_out/counters_all.txt
for grp in "FETCH_SIZE WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "MemUnitStalled MemUnitBusy" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmc_$tag
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py panel > /tmp/pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $GRAFT_REPO_ROOT/tools/debug/pmc_group.py "$f" gemm_panel_kernel; else echo "no output for $grp"; tail -3 /tmp/pmc_$tag.log; fi
done
