timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attn_decode" -p no:cacheprovider 2>&1 | tail -3
for c in 0 1; do echo "VCLA_ATTN_COOP=$c"; VCLA_ATTN_COOP=$c python bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"decode_ms_per_token_step": [0-9.]*'; done
echo "B=1 coop forced"; VCLA_ATTN_COOP=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"decode_ms_per_token_step": [0-9.]*'
