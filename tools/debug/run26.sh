cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemv or gemm_shapes" 2>&1 | tail -3
for g in 256 512 768; do VCLA_GEMV_GRID=$g python tools/bench_kernels.py gemv1 2>&1 | grep "gemv1 " | awk -v g=$g '{print "grid" g, $0}' | cut -c1-90; done
for cfg in "VCLA_GEMV1X=0" "VCLA_GEMV_GRID=512" "VCLA_GEMV_GRID=768"; do
  echo "== $cfg"
  env $cfg timeout 600 python bench.py --steps 2 --warmup 1 --steps-b64 0 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['unit'], d['ms_per_step'])"
done
