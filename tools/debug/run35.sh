cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -3
for cfg in "VCLA_FUSE_ATTN_O=0" "VCLA_FUSE_ATTN_O=1" "VCLA_FUSE_ATTN_O=1 VCLA_FUSE_GEMV_WGS=256" "VCLA_FUSE_ATTN_O=1 VCLA_FUSE_GEMV_WGS=480"; do
  echo "== $cfg"
  env $cfg timeout 600 python bench.py --steps 2 --warmup 1 --steps-b64 0 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['unit'], d['ms_per_step'], d['breakdown_ms'])"
done
rm -rf gpurun_out/pg
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pg -o t -- python $R/bench.py --steps 1 --warmup 1 --steps-b64 0 --no-cpu-baseline --new-tokens 64 2>&1 | tail -1 | cut -c1-100)
f=$(find gpurun_out/pg -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/prof_by_grid.py $f 7
rm -rf gpurun_out/pg
