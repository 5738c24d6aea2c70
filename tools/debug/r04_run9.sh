cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "attention" 2>&1 | tail -3
for f in 2 1; do echo "-- VCLA_ATTN_VIT=$f"; VCLA_ATTN_VIT=$f python tools/bench_kernels.py vitattn 2>&1 | grep -E "whole-seq"; done | tee gpurun_out/r04h_vitattn.txt
