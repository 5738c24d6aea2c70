cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "dstream or deferred" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -k "batch64 or 7b_decode" 2>&1 | tail -2
for i in 1 2; do timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['breakdown_ms'])"; done
