cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== vit attention tests (DMA form default)"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "attention" 2>&1 | tail -3
for f in 2 1; do echo "-- VCLA_ATTN_VIT=$f"; VCLA_ATTN_VIT=$f python tools/bench_kernels.py vitattn 2>&1 | grep -E "whole-seq"; done | tee gpurun_out/r04e_vitattn.txt
echo "== model tests (vision)"
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider -k "golden or vision_stack or bf16_path" 2>&1 | tail -3
echo "== bench B=64"
timeout 900 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --steps-c4 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['images_per_sec'], d['breakdown_ms'])" | tee gpurun_out/r04e_b64.txt
