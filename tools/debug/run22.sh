cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "attn_decode" 2>&1 | tail -2
for nw in 4 8; do VCLA_ATTN_NW=$nw python tools/pmc_attn_decode.py 2>&1 | tail -1; done
