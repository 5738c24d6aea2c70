cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
i=0
for cfg in "VCLA_ATTN_PF_WGS=0" "VCLA_ATTN_PF_WGS=224" "VCLA_ATTN_PF_WGS=224 VCLA_ATTN_PF_ROWS=3072" "VCLA_ATTN_PF_WGS=224 VCLA_ATTN_PF_ROWS=2048" "VCLA_ATTN_PF_WGS=480 VCLA_ATTN_PF_ROWS=3072" "VCLA_ATTN_PF_WGS=992 VCLA_ATTN_PF_ROWS=3072"; do
  i=$((i+1)); echo "== $cfg"
  rm -rf gpurun_out/pg$i
  (cd /tmp && export TMPDIR=/tmp && env $cfg timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pg$i -o t -- python $R/bench.py --steps 1 --warmup 1 --steps-b64 0 --no-cpu-baseline --new-tokens 64 2>&1 | tail -1 | cut -c1-120)
  f=$(find gpurun_out/pg$i -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/prof_by_grid.py $f 7
  rm -rf gpurun_out/pg$i
done
