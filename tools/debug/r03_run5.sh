#!/bin/bash
mkdir -p gpurun_out
exec > >(tee gpurun_out/r03_run5.log) 2>&1
echo "== macro graph debug (B=1, 3 generate calls)"
VCLA_MACRO_GRAPH_DEBUG=1 timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -30
import sys, os, time, torch
sys.path.insert(0, "."); sys.path.insert(0, "visual-chinese-llama-alpaca_amd")
import visualcla
from visualcla.synthetic import make_inputs, stub_tokenizer
m = visualcla.VisualCLAModel.from_random(visualcla.visualcla_7b_config(), device="cuda:0", torch_dtype=torch.bfloat16, seed=0)
m.tokenizer = stub_tokenizer(); m.image_at_head = False
px, ids, mask = make_inputs(m.config, 1, 128)
px, ids, mask = px.cuda().bfloat16(), ids.cuda(), mask.cuda()
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = m.generate(input_ids=ids, pixel_values=px, attention_mask=mask, max_new_tokens=1, do_sample=False, eos_token_id=None)
    torch.cuda.synchronize(); print(f"call {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
side = torch.cuda.Stream()
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(side):
        m.embed_images(px, _persistent=True)
    torch.cuda.synchronize(); print(f"vision on side stream {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
for i in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.embed_images(px)
    torch.cuda.synchronize(); print(f"vision eager {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
PY
echo "== vit microbench (tail selection by K)"; timeout 300 python tools/bench_kernels.py vit 2>&1 | grep -v amdgpu.ids | grep auto
echo "== kernel parity (gemm)"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm or attention" 2>&1 | tail -4
echo "== 7B vision parity"; timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider -x -k "7b_vision or 336 or prefill_and_decode" 2>&1 | tail -4
echo "== done"
