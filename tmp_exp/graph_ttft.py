import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd"))
import torch, visualcla
from visualcla.synthetic import make_inputs, stub_tokenizer
cfg = visualcla.visualcla_7b_config()
m = visualcla.VisualCLAModel.from_random(cfg, device="cuda:0", torch_dtype=torch.bfloat16, seed=0)
m.tokenizer = stub_tokenizer(); m.image_at_head = False
px, ids, mask = make_inputs(m.config, 1, 128)
px, ids = px.to(m.device, torch.bfloat16), ids.to(m.device)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("vision eager ms", timeit(lambda: m.embed_images(px)))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    m.embed_images(px)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = m.embed_images(px)
    print("vision graph ms", timeit(lambda: g.replay()))
    ref = m.embed_images(px); g.replay(); torch.cuda.synchronize()
    print("graph == eager:", torch.equal(ref, out))
    emb = torch.randn(1, 128, 4096, device="cuda:0").to(torch.bfloat16)
    cache = m._new_cache(1, 256)
    def pf():
        cache.length = 0
        return m._prefill(emb, cache, None, all_logits=False)
    print("prefill eager ms", timeit(pf))
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        lg = pf()
    print("prefill graph ms", timeit(lambda: g2.replay()))
