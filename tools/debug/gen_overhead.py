"""where does generate() spend host time at B=1?  (cProfile over 3 calls, top cumulative entries)"""
import cProfile, pstats, sys, time, torch
sys.path.insert(0, "."); sys.path.insert(0, "visual-chinese-llama-alpaca_amd")
import visualcla
from visualcla.synthetic import make_inputs, stub_tokenizer
cfg = visualcla.visualcla_7b_config()
m = visualcla.VisualCLAModel.from_random(cfg, device="cuda:0", torch_dtype=torch.bfloat16, seed=0)
m.tokenizer = stub_tokenizer(); m.image_at_head = False
px, ids, mask = make_inputs(cfg, 1, 128)
px, ids, mask = px.cuda().bfloat16(), ids.cuda(), mask.cuda()
kw = dict(input_ids=ids, pixel_values=px, attention_mask=mask, max_new_tokens=128, do_sample=False, eos_token_id=None)
for _ in range(2): m.generate(**kw)
torch.cuda.synchronize()
t0 = time.perf_counter(); 
for _ in range(3): m.generate(**kw)
torch.cuda.synchronize(); print("ms per generate", (time.perf_counter() - t0) / 3 * 1e3)
# host time until the last launch is enqueued (no sync) = pure CPU overhead visible if > GPU time
t0 = time.perf_counter(); out = m.generate(**kw); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0):.2f} ms, then wait {1e3*(t2-t1):.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(3): m.generate(**kw)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
