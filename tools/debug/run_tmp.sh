cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "test_gemm" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider -k "golden or 7b_prefill or batch64 or vision_stack" 2>&1 | tail -2
for i in 1 2; do timeout 600 python bench.py --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --steps-c4 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])"; done
