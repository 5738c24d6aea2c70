cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "dstream or fragment" 2>&1 | tail -4
VCLA_BENCH_MS=64,32 timeout 600 python tools/bench_kernels.py dstream 2>&1 | grep "M=" | tee gpurun_out/r02_dstream_c.txt
for sk in 4 1 2; do VCLA_DS_SPLITK=$sk timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('splitk=$sk', d['value'], d['breakdown_ms'])"; done
