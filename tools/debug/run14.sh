cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "attn_decode" 2>&1 | tail -4
for nw in 8 4; do VCLA_ATTN_NW=$nw timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('attn_nw=$nw', d['value'], d['breakdown_ms'])"; done
