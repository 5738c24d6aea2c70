import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "visual-chinese-llama-alpaca_amd")
from oracle import sampling_oracle as S, preprocess_oracle as P
from tests.test_sampling_oracle import CFGS
from tests.test_gpu_sampling import _run
ci, h, V = 2, 200, 49958
cfg = CFGS[ci]
rng = np.random.default_rng(1000 * ci + h + V)
B = 5
logits = (rng.standard_normal((B, V)) * 3).astype(np.float32)
hist = rng.integers(0, 6, size=(h, B)).astype(np.int64)
u = rng.random(B).astype(np.float32)
out, kept_ids, kept_p, n_kept = _run(logits, hist, cfg, u)
for b in range(B):
    sc = S.process_scores(logits[b], hist[:, b], cfg)
    ids, probs = S.kept_distribution(sc)
    n = int(n_kept[b])
    k = kept_ids[b, :n]
    bad = np.nonzero(k != ids[:n])[0] if n == len(ids) else None
    print(b, n, len(ids), bad)
    if bad is not None and len(bad):
        for i in bad[:4]:
            print("  idx", i, "kernel", k[i], logits[b, k[i]].tobytes().hex(), float(logits[b, k[i]]), "oracle", ids[i], logits[b, ids[i]].tobytes().hex(), float(logits[b, ids[i]]), kept_p[b, i], probs[i])

# ---- preprocess
from visualcla.preprocess import GpuClipImageProcessor, plan_tables
hw, size = (300, 400), 224
img = (np.random.default_rng(hw[0] * 7 + hw[1]).random((*hw, 3)) * 255).astype(np.uint8)
proc = GpuClipImageProcessor(size=size)
got = proc(img).pixel_values[0].cpu().numpy()
want = P.clip_preprocess(img, size)
d = np.abs(got - want)
print("preprocess max diff", d.max(), "n mismatched", (got != want).sum(), "of", got.size)
# undo the normalisation to see whether the uint8 stage differs
mean = np.array(P.CLIP_MEAN, np.float32)[:, None, None]; std = np.array(P.CLIP_STD, np.float32)[:, None, None]
g8 = np.rint((got * std + mean) * 255); w8 = np.rint((want * std + mean) * 255)
print("uint8-stage mismatches", (g8 != w8).sum(), "max", np.abs(g8 - w8).max())
idx = np.argwhere(got != want)[:5]
for c, y, x in idx:
    print(c, y, x, got[c, y, x], want[c, y, x], g8[c, y, x], w8[c, y, x])
