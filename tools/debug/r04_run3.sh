cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== vit attention tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "attention" 2>&1 | tail -5
echo "== vitattn microbench"
python tools/bench_kernels.py vitattn 2>&1 | grep -E "^attn|^==" | tee gpurun_out/r04c_vitattn.txt
echo "== model tests (vision)"
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider -k "golden or vision_stack or bf16_path" 2>&1 | tail -5
echo "== bench B=64"
timeout 900 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --steps-c4 0 2>&1 | tail -1 | cut -c1-900 | tee gpurun_out/r04c_b64.txt
