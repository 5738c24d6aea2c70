cd /tmp && export TMPDIR=/tmp
for cfg in "4 320 8" "4 512 16" "4 768 16" "4 1024 32" "4 1536 32" "4 2048 64" "2 320 8" "2 768 16"; do
  set -- $cfg
  rm -rf /tmp/pd
  VCLA_PANEL_DIAG=$1 VCLA_PANEL_WGS=$2 VCLA_PANEL_SMAX=$3 rocprofv3 --kernel-trace --output-format csv -d /tmp/pd -o t -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py panel > /dev/null 2>&1
  echo "== DIAG=$1 WGS=$2 SMAX=$3"
  python $GRAFT_REPO_ROOT/tools/debug/trace_summary.py $(find /tmp/pd -name "*kernel_trace.csv" | head -1) gemm_panel_kernel | cut -c60-200
done
