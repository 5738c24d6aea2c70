cd $GRAFT_REPO_ROOT
for pad in 8 24 0; do VCLA_CTX_PAD=$pad timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ctx_pad=$pad', d['value'], d['breakdown_ms'])"; done
