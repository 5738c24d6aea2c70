cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemv or attn_decode" 2>&1 | tail -3
for cfg in "VCLA_GEMV1X=0" "VCLA_GEMV1X=1 VCLA_PF_WGS=0 VCLA_ATTN_PF_WGS=0" "VCLA_GEMV1X=1 VCLA_PF_WGS=512 VCLA_ATTN_PF_WGS=0" "VCLA_GEMV1X=1 VCLA_PF_WGS=0 VCLA_ATTN_PF_WGS=224" "VCLA_GEMV1X=1 VCLA_PF_WGS=512 VCLA_ATTN_PF_WGS=224" "VCLA_GEMV1X=1 VCLA_PF_WGS=256 VCLA_ATTN_PF_WGS=480" "VCLA_GEMV1X=1 VCLA_PF_WGS=1024 VCLA_ATTN_PF_WGS=992"; do
  echo "== $cfg"
  env $cfg timeout 600 python bench.py --steps 2 --warmup 1 --steps-b64 0 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['unit'], d['ms_per_step'], {k:v for k,v in d.get('config',{}).items() if 'ms' in k or 'tok' in k})"
done
