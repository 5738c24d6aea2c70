cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemv or fp8" 2>&1 | tail -3
for cfg in "VCLA_GEMV1X=0" "VCLA_GEMV1X=1"; do
  echo "== fp8 $cfg"
  env $cfg timeout 600 python bench.py --fp8 --steps 2 --warmup 1 --steps-b64 0 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['unit'], d['ms_per_step'])"
done
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "fp8" 2>&1 | tail -3
