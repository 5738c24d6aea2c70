"""End-to-end parity of the HIP path (through the drop-in Python API -> C ABI -> gfx950 kernels) against
 (a) the golden tensors produced by the reference's own VisualCLAModel (tests/golden/ref_*.npz), and
 (b) the CPU oracle on the same seeded inputs,
plus size-independent properties at the full VisualCLA-7B geometry.

Tolerances (north_star: logits within 1e-3 of the reference):
  * fp32 activation mode: every stage and the logits within 1e-3 ABS of the fp32 reference (observed ~1e-5);
    greedy token ids identical to the reference's.
  * bf16 product mode (bf16 storage, fp32 accumulate on MFMA): logits within 6e-2 ABS of the fp32 reference
    (bf16 carries 8 mantissa bits, logits are O(1)); mean abs error within 1e-2.
"""
import os

import numpy as np
import pytest
import torch

from oracle import visualcla_oracle as O
from tests.helpers import make_hip_model, to_vcla_config, stub_tokenizer

pytestmark = pytest.mark.gpu
CASES = {"tiny_b2": O.cfg_tiny, "small_b2": O.cfg_small}


def _report(line):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_report.txt", "a") as f:
        f.write(line + "\n")


def _setup(name, golden_dir):
    g = np.load(os.path.join(golden_dir, f"ref_{name}.npz"))
    cfg = CASES[name]()
    B, T, n_new = (int(x) for x in g["_meta"])
    W = O.make_weights(cfg, seed=0)
    px, ids, mask = O.make_inputs(cfg, B, T)
    return g, cfg, W, px, ids, mask, n_new


@pytest.mark.parametrize("name", sorted(CASES))
def test_fp32_mode_matches_reference_golden(name, golden_dir):
    g, cfg, W, px, ids, mask, n_new = _setup(name, golden_dir)
    m = make_hip_model(cfg, W, torch.float32)
    taps = {}
    out = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), taps=taps)
    torch.cuda.synchronize()
    worst = 0.0
    for k in g.files:
        if k.startswith("_") or k in ("generated", "vit_embed"):
            continue
        ref = torch.from_numpy(g[k])
        got = taps[k].float().cpu().reshape(ref.shape)
        err = (got - ref).abs().max().item()
        _report(f"fp32 {name} {k}: max_abs_err={err:.3e} (ref absmax {ref.abs().max().item():.2e})")
        worst = max(worst, err)
        assert err <= 1e-3, (k, err)
    assert (out.logits.cpu() - torch.from_numpy(g["logits"])).abs().max().item() <= 1e-3
    toks = m.generate(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), max_new_tokens=n_new,
                      do_sample=False, eos_token_id=None)
    assert np.array_equal(toks.cpu().numpy(), g["generated"]), (toks, g["generated"])
    _report(f"fp32 {name}: worst stage err {worst:.3e}; greedy ids == reference")


@pytest.mark.parametrize("name", sorted(CASES))
def test_bf16_mode_vs_reference_golden(name, golden_dir):
    g, cfg, W, px, ids, mask, n_new = _setup(name, golden_dir)
    m = make_hip_model(cfg, W, torch.bfloat16)
    taps = {}
    out = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), taps=taps)
    for k in g.files:
        if k.startswith("_") or k in ("generated", "vit_embed"):
            continue
        ref = torch.from_numpy(g[k])
        got = taps[k].float().cpu().reshape(ref.shape)
        err = (got - ref).abs()
        rel = err.max().item() / max(ref.abs().max().item(), 1e-6)
        _report(f"bf16 {name} {k}: max_abs_err={err.max().item():.3e} mean={err.mean().item():.3e} rel_to_absmax={rel:.3e}")
        assert rel <= 5e-2, (k, rel)          # every stage within 5% of its dynamic range
    ref = torch.from_numpy(g["logits"])
    err = (out.logits.cpu() - ref).abs()
    assert err.max().item() <= 6e-2 and err.mean().item() <= 1e-2, (err.max().item(), err.mean().item())
    # greedy agreement wherever the reference's top-1 margin exceeds the error bound
    top2 = ref.topk(2, dim=-1).values
    decided = (top2[..., 0] - top2[..., 1]) > 2 * 6e-2
    agree = out.logits.cpu().argmax(-1) == ref.argmax(-1)
    assert bool(agree[decided].all())


def test_generate_paths_agree(golden_dir):
    """device-resident greedy loop (eager and hipGraph replay) == host-driven step loop == reference ids (fp32 mode)."""
    g, cfg, W, px, ids, mask, n_new = _setup("small_b2", golden_dir)
    m = make_hip_model(cfg, W, torch.float32)
    kw = dict(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), max_new_tokens=n_new,
              do_sample=False, eos_token_id=None)
    eager = m.generate(use_graph=False, **kw)
    graph = m.generate(use_graph=True, **kw)
    graph2 = m.generate(use_graph=True, **kw)         # replays the cached graph
    from transformers import LogitsProcessorList
    stepwise = m.generate(logits_processor=LogitsProcessorList([lambda ids_, s: s]), **kw)
    ref = torch.from_numpy(g["generated"])
    for nm, t in (("eager", eager), ("graph", graph), ("graph2", graph2), ("stepwise", stepwise)):
        assert torch.equal(t.cpu(), ref), (nm, t, ref)


def test_eos_stops_and_pads(golden_dir):
    g, cfg, W, px, ids, mask, n_new = _setup("tiny_b2", golden_dir)
    m = make_hip_model(cfg, W, torch.float32)
    ref = torch.from_numpy(g["generated"])            # [[163, 55, 89, 245, 56, 56], [208, ...]]
    eos = int(ref[0, 2])
    toks = m.generate(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), max_new_tokens=n_new,
                      do_sample=False, eos_token_id=eos, pad_token_id=0).cpu()
    want = O.visualcla_generate(ids, px, mask, W, cfg, max_new_tokens=n_new, eos_token_id=eos)
    n = min(toks.shape[1], want.shape[1])
    assert torch.equal(toks[0, :3], ref[0, :3]) and bool((toks[0, 3:] == 0).all())
    assert torch.equal(toks[1, :n], ref[1, :n])


def test_left_padding_mask_matches_oracle():
    cfg = O.cfg_tiny()
    W = O.make_weights(cfg, seed=0)
    px, ids, mask = O.make_inputs(cfg, 2, 34, n_prefix=5)
    # left-pad sample 1 by 3 tokens (pad id 0), as a batched chat() would (the tail keeps the </img> marker)
    ids[1] = torch.cat([torch.zeros(3, dtype=ids.dtype), ids[1, :-3]])
    mask[1, :3] = 0
    ref = O.visualcla_forward(ids, px, mask, W, cfg)
    m = make_hip_model(cfg, W, torch.float32)
    out = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda()).logits.cpu()
    valid = mask.bool()
    assert (out[valid] - ref[valid]).abs().max().item() <= 1e-3


def test_error_conventions():
    cfg = O.cfg_tiny()
    W = O.make_weights(cfg, seed=0)
    px, ids, mask = O.make_inputs(cfg, 1, 24)
    m = make_hip_model(cfg, W, torch.float32)
    bad = ids.clone()
    p0 = int((bad[0] == cfg.img_start_token_id).nonzero()[0])
    bad[0, p0 + cfg.resampler.num_query_tokens + 1] = 5
    with pytest.raises(ValueError):
        m.forward(input_ids=bad.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda())
    with pytest.raises(ValueError):
        m.generate(input_ids=bad.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), max_new_tokens=2, do_sample=False)
    with pytest.raises(ValueError):
        m.forward(input_ids=ids.cuda(), pixel_values=px[:, :, :28, :28].cuda(), attention_mask=mask.cuda())
    # text-only prompt (no image slot) must run
    txt = torch.randint(3, 300, (1, 9))
    out = m.forward(input_ids=txt.cuda(), attention_mask=torch.ones_like(txt).cuda())
    ref = O.visualcla_forward(txt, None, torch.ones_like(txt), W, cfg)
    assert (out.logits.cpu() - ref).abs().max().item() <= 1e-3


def test_image_at_head_mode():
    cfg = O.cfg_tiny()
    W = O.make_weights(cfg, seed=0)
    px, _, _ = O.make_inputs(cfg, 2, 24)
    Q = cfg.resampler.num_query_tokens
    ids = torch.randint(3, 300, (2, 10))
    m = make_hip_model(cfg, W, torch.float32)
    m.image_at_head = True
    out = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=torch.ones_like(ids).cuda()).logits.cpu()
    # oracle: emb[:, :2] ++ image ++ emb[:, 2:]  (modeling_visualcla.py:291)
    img = O.image_embeds(px, W, cfg)
    emb = W["text_model.model.embed_tokens.weight"][ids]
    x = torch.cat([emb[:, :2], img, emb[:, 2:]], dim=1)
    h = O.llama_forward(x, W, cfg.text, torch.ones(2, 10 + Q, dtype=torch.int64))
    ref = O.lm_head(h, W)
    assert out.shape == ref.shape and (out - ref).abs().max().item() <= 1e-3


def test_state_dict_roundtrip_and_dtype_switch():
    cfg = O.cfg_tiny()
    W = O.make_weights(cfg, seed=0)
    px, ids, mask = O.make_inputs(cfg, 1, 24)
    m = make_hip_model(cfg, W, torch.bfloat16)
    sd = m.state_dict()
    for k, v in W.items():
        if "pooler" in k:
            continue            # dead weights (computed and discarded by the reference); not held by the HIP path
        assert torch.equal(sd[k].reshape(v.shape), v), k
    a = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda()).logits.cpu()
    m.float()
    b = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda()).logits.cpu()
    ref = O.visualcla_forward(ids, px, mask, W, cfg)
    assert (b - ref).abs().max().item() <= 1e-3 and (a - ref).abs().max().item() <= 6e-2


@pytest.mark.parametrize("nb", [1, 2])
def test_fp8_decode_weights_match_oracle_on_dequantised_weights(golden_dir, nb):
    """BASELINE configs[4] weight path: fp8 (e4m3fn) copies serve the decode steps (M = 1 GEMV, M > 1 panel kernel), prefill stays
    bf16.  The decode-step logits must equal the oracle evaluated the same way (prefill on W, decode steps on the DEQUANTISED
    matrices)."""
    from visualcla.weights import quantize_fp8_rows, dequantize_fp8_rows, pad_to
    from transformers import LogitsProcessorList
    g, cfg, W, px, ids, mask, n_new = _setup("small_b2", golden_dir)
    px, ids, mask = px[:nb], ids[:nb], mask[:nb]
    m = make_hip_model(cfg, W, torch.bfloat16)
    m.enable_fp8_decode()
    assert m.fp8_decode
    seen = []

    def grab(ids_, scores):
        seen.append(scores.detach().float().cpu().clone())
        return scores
    toks = m.generate(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), max_new_tokens=4, do_sample=False,
                      eos_token_id=None, logits_processor=LogitsProcessorList([grab])).cpu()
    Wq = dict(W)
    for k, v in W.items():
        if k.startswith("text_model.") and v.dim() == 2 and "embed_tokens" not in k:
            wp = torch.zeros(pad_to(v.shape[0], 128), v.shape[1])
            wp[: v.shape[0]] = v
            q, sc = quantize_fp8_rows(wp.to(torch.bfloat16))
            Wq[k] = dequantize_fp8_rows(q, sc)[: v.shape[0]]
    # oracle: prefill with the bf16-rounded weights, then teacher-forced decode steps on the dequantised fp8 weights
    img = O.image_embeds(px, W, cfg)
    x = O.embed_and_splice(ids, img, W, cfg)
    cache = [None] * cfg.text.num_hidden_layers
    h = O.llama_forward(x, W, cfg.text, mask, cache, 0)
    ref_steps = [O.lm_head(h[:, -1:], W)[:, 0]]
    past = ids.shape[1]
    for s_ in range(3):
        e = W["text_model.model.embed_tokens.weight"][toks[:, s_]][:, None, :]
        h = O.llama_forward(e, Wq, cfg.text, torch.ones(nb, past + 1, dtype=torch.int64), cache, past)
        ref_steps.append(O.lm_head(h, Wq)[:, 0])
        past += 1
    for s_ in range(4):
        err = (seen[s_] - ref_steps[s_]).abs()
        _report(f"fp8 decode step {s_}: logits max err {err.max().item():.3e} mean {err.mean().item():.3e}")
        assert err.max().item() <= 6e-2 and err.mean().item() <= 1e-2, (s_, err.max().item())
    m.enable_fp8_decode(False)
    assert not m.fp8_decode


# ------------------------------------------------------------------ full VisualCLA-7B geometry: properties
@pytest.fixture(scope="module")
def model_7b():
    import visualcla
    cfg7 = visualcla.visualcla_7b_config()
    m = visualcla.VisualCLAModel.from_random(cfg7, device="cuda:0", torch_dtype=torch.bfloat16, seed=0)
    ocfg = O.cfg_7b()
    m.tokenizer = stub_tokenizer(ocfg)
    m.image_at_head = False
    yield m, ocfg
    del m
    torch.cuda.empty_cache()


def test_7b_decode_equals_forward_and_batch_invariance(model_7b):
    """At full size the oracle is too slow to run in a test; the path is checked through properties:
    (1) KV-cache decode logits == full-sequence forward logits, (2) a sample's output does not depend on its batch
    neighbours or position, (3) greedy generate is deterministic and graph replay == eager."""
    m, ocfg = model_7b
    px, ids, mask = O.make_inputs(ocfg, 3, 128)
    px, ids, mask = px.cuda(), ids.cuda(), mask.cuda()
    kw = dict(max_new_tokens=6, do_sample=False, eos_token_id=None)
    toks = m.generate(input_ids=ids, pixel_values=px, attention_mask=mask, use_graph=False, **kw)
    toks_g = m.generate(input_ids=ids, pixel_values=px, attention_mask=mask, use_graph=True, **kw)
    assert torch.equal(toks, toks_g)
    # (2) permute the batch and run a single sample
    perm = torch.tensor([2, 0, 1], device="cuda")
    toks_p = m.generate(input_ids=ids[perm], pixel_values=px[perm], attention_mask=mask[perm], **kw)
    solo = m.generate(input_ids=ids[1:2], pixel_values=px[1:2], attention_mask=mask[1:2], **kw)
    full = m.forward(input_ids=torch.cat([ids, toks[:, :5]], 1), pixel_values=px,
                     attention_mask=torch.ones(3, 133, dtype=torch.int64, device="cuda")).logits
    # (1) the token chosen at step s must be the argmax of the full-forward logits at position 127+s, unless the
    # top-2 margin there is inside bf16 noise (different kernels are used for M=3 decode and M=384 prefill)
    for s in range(6):
        lg = full[:, 127 + s].float()
        top2 = lg.topk(2, dim=-1)
        decided = (top2.values[:, 0] - top2.values[:, 1]) > 0.1
        assert bool((top2.indices[:, 0] == toks[:, s])[decided].all()), s
        if not bool(decided.all()):
            break
    assert torch.isfinite(full).all()
    _report(f"7B: generate {toks.tolist()}; logits std {full.float().std().item():.3f}")
    # (2) same kernels, different batch placement: a wrong index anywhere shows up as a different token stream
    agree_perm = (toks_p == toks[perm]).float().mean().item()
    agree_solo = (solo == toks[1:2]).float().mean().item()
    _report(f"7B: permuted-batch agreement {agree_perm:.2f}, solo agreement {agree_solo:.2f}")
    assert agree_perm == 1.0            # same M, same kernels, same per-row arithmetic -> bit identical
    assert agree_solo >= 0.5            # M=1 uses the same GEMV kernel family; allow bf16 near-ties


def test_set_image_size_retargets_the_vision_tower(golden_dir):
    """BASELINE configs[4] patching (336 px at 7B; 84 px = 6x6 patches here): the position embedding grows bicubically
    (weights.extend_position_embedding), every kernel is generic in the token count, fp8 copies survive the re-pack"""
    import copy
    from visualcla.weights import extend_position_embedding
    g, cfg, W, px, ids, mask, n_new = _setup("tiny_b2", golden_dir)
    m = make_hip_model(cfg, W, torch.float32)
    big = cfg.vision.image_size + 2 * cfg.vision.patch_size
    m.set_image_size(big)
    cfg2 = copy.deepcopy(cfg)
    cfg2.vision.image_size = big
    W2 = {k: v.clone() for k, v in W.items()}
    extend_position_embedding(W2, cfg.vision.patch_size, big)
    px2, ids2, mask2 = O.make_inputs(cfg2, 2, ids.shape[1])
    got = m.forward(input_ids=ids2.cuda(), pixel_values=px2.cuda(), attention_mask=mask2.cuda()).logits.float().cpu()
    want = O.visualcla_forward(ids2, px2, mask2, W2, cfg2)
    want = want["logits"] if isinstance(want, dict) else want
    err = (got - want).abs().max().item()
    _report(f"set_image_size({big}) fp32 logits max err {err:.3e}")
    assert err <= 1e-3
    with pytest.raises(ValueError):                       # the old resolution is now refused, like the reference's CLIP
        m.embed_images(px.cuda())
    with pytest.raises(ValueError):
        m.set_image_size(big + 3)                         # not a multiple of the patch size
    mb = make_hip_model(cfg, W, torch.bfloat16)
    mb.enable_fp8_decode()
    mb.set_image_size(big)
    assert mb.fp8_decode
    toks = mb.generate(input_ids=ids2.cuda(), pixel_values=px2.cuda(), attention_mask=mask2.cuda(), max_new_tokens=3, do_sample=False,
                       eos_token_id=None)
    assert toks.shape == (2, 3)


@pytest.mark.parametrize("fp8", [False, True])
def test_7b_batch64_decode_matches_full_forward(model_7b, fp8):
    """configs[2] geometry (B = 64: 8-wave panel GEMMs, row-cooperative decode attention, fused reduce + RMSNorm): every token a
    decode step picks must be the argmax of the full-sequence forward logits at that position wherever the top-2 margin there is
    outside bf16 / fp8 noise"""
    m, ocfg = model_7b
    B, T, n_new = 64, 128, 4
    px, ids, mask = O.make_inputs(ocfg, B, T)
    px, ids, mask = px.cuda(), ids.cuda(), mask.cuda()
    if fp8:
        m.enable_fp8_decode()
    try:
        toks = m.generate(input_ids=ids, pixel_values=px, attention_mask=mask, max_new_tokens=n_new, do_sample=False, eos_token_id=None)
    finally:
        if fp8:
            m.enable_fp8_decode(False)
    full = m.forward(input_ids=torch.cat([ids, toks[:, :n_new - 1]], 1), pixel_values=px,
                     attention_mask=torch.ones(B, T + n_new - 1, dtype=torch.int64, device="cuda")).logits
    assert torch.isfinite(full).all()
    margin = 0.5 if fp8 else 0.1        # fp8 decode weights vs the bf16 forward: ~3 mantissa bits on the weights
    checked = agree = 0
    alive = torch.ones(B, dtype=torch.bool, device="cuda")     # a row is compared until its first undecided position
    for s in range(n_new):
        lg = full[:, T - 1 + s].float()
        top2 = lg.topk(2, dim=-1)
        decided = ((top2.values[:, 0] - top2.values[:, 1]) > margin) & alive
        checked += int(decided.sum())
        agree += int((top2.indices[:, 0] == toks[:, s])[decided].sum())
        alive &= decided
    _report(f"7B B=64 fp8={fp8}: {agree}/{checked} decided positions agree with the full forward")
    if fp8:     # quantisation noise is not bounded by a fixed margin on a random-weight model: ask for near-total agreement
        assert checked >= 16 and agree >= 0.9 * checked
    else:       # decode (panel kernels) and forward (256x256 tiles) round differently: allow the odd near-tie past the margin
        assert checked >= B and agree >= 0.98 * checked
