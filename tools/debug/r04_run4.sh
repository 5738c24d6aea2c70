cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== vit attention tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "attention_vit" 2>&1 | tail -3
for abl in 0 1 2 3 4; do echo "-- VCLA_ATTN_VIT_ABL=$abl"; VCLA_ATTN_VIT_ABL=$abl python tools/bench_kernels.py vitattn 2>&1 | grep -E "whole-seq"; done | tee gpurun_out/r04d_vitattn_abl.txt
