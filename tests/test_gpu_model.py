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
        if k.startswith("_") or k == "generated":
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
        if k.startswith("_") or k == "generated":
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


# ---------------------------------------------------------------------------------------------------------------------
# The reference's own outputs on the edge cases of forward / generate (tests/golden/ref_edge_cases.npz, oracle/make_golden.py:
# edge_cases): left-padded batches, text-only prompts, labels -> .loss in both image placements, past_key_values, a row without
# an image slot.  fp32 mode: logits within 1e-3 (north_star), loss within 1e-4, greedy ids identical; bf16 mode: the stated bf16 bounds.
# ---------------------------------------------------------------------------------------------------------------------
def _edge(golden_dir, case):
    g = np.load(os.path.join(golden_dir, "ref_edge_cases.npz"))
    pre = case + "__"
    return {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}


def _edge_model(dtype, cname="tiny", T=34, npre=5):
    cfg = {"tiny": O.cfg_tiny, "small": O.cfg_small}[cname]()
    W = O.make_weights(cfg, seed=0)
    px, ids, mask = O.make_inputs(cfg, 2, T, n_prefix=npre)
    return cfg, make_hip_model(cfg, W, dtype), px, ids, mask


def _logit_bounds(dtype):
    return (1e-3, 1e-3) if dtype == torch.float32 else (6e-2, 1e-2)       # (max, mean) abs error vs the fp32 reference


def _check_logits(tag, got, ref, dtype, valid=None):
    got, ref = got.float().cpu(), ref.float()
    assert got.shape == ref.shape, (tag, got.shape, ref.shape)
    err = (got - ref).abs()
    if valid is not None:
        err = err[valid]
    mx, mean = _logit_bounds(dtype)
    _report(f"edge {tag} [{dtype}]: logits max_abs_err={err.max().item():.3e} mean={err.mean().item():.3e}")
    assert err.max().item() <= mx and err.mean().item() <= mean, (tag, err.max().item(), err.mean().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cname", ["tiny", "small"])
def test_edge_left_padded_batch_matches_reference(cname, dtype, golden_dir):
    e = _edge(golden_dir, f"leftpad_{cname}")
    T, npre, n_new = (int(x) for x in e["meta"])
    cfg, m, px, _, _ = _edge_model(dtype, cname, T, npre)
    ids, mask = e["input_ids"], e["attention_mask"]
    out = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda())
    _check_logits(f"leftpad_{cname}", out.logits, e["logits"], dtype, valid=mask.bool())
    if dtype == torch.float32:          # bf16: ids may legitimately flip at small top-2 margins; the logits above are the check
        toks = m.generate(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), max_new_tokens=n_new, do_sample=False,
                          eos_token_id=None).cpu()
        assert torch.equal(toks, e["generated"]), (toks, e["generated"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_edge_text_only_forward_matches_reference(dtype, golden_dir):
    e = _edge(golden_dir, "textonly")
    cfg, m, _, _, _ = _edge_model(dtype)
    ids, mask = e["input_ids"], e["attention_mask"]
    out = m.forward(input_ids=ids.cuda(), attention_mask=torch.ones_like(ids).cuda())
    _check_logits("textonly(full mask)", out.logits, e["logits_full_mask"], dtype)
    out = m.forward(input_ids=ids.cuda(), attention_mask=mask.cuda())
    _check_logits("textonly(left pad)", out.logits, e["logits"], dtype, valid=mask.bool())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_edge_loss_with_the_image_in_its_slots_matches_reference(dtype, golden_dir):
    """forward(labels=...) -> .loss (modeling_visualcla.py:321-328 -> LlamaForCausalLM's shifted cross-entropy)"""
    e = _edge(golden_dir, "slot_labels")
    cfg, m, px, ids, mask = _edge_model(dtype)
    out = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), labels=e["labels"].cuda())
    _check_logits("slot_labels", out.logits, e["logits"], dtype)
    d = abs(float(out.loss) - float(e["loss"][0]))
    _report(f"edge slot_labels [{dtype}]: loss {float(out.loss):.6f} vs reference {float(e['loss'][0]):.6f}")
    assert d <= (1e-4 if dtype == torch.float32 else 1e-2), d
    tup = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), labels=e["labels"].cuda(), return_dict=False)
    assert len(tup) == 2 and abs(float(tup[0]) - float(out.loss)) == 0.0        # (loss, logits) as HF's tuple output


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_edge_image_at_head_logits_loss_and_ids_match_reference(dtype, golden_dir):
    """image_at_head=True with labels: the reference places the Q ignore-labels after position 0 while the image embeds go in after
    position 1 (modeling_visualcla.py:291 vs :315); the loss fixture has labels[:, 1] supervised, so a 'corrected' placement fails here"""
    e = _edge(golden_dir, "head_labels")
    cfg, m, px, _, _ = _edge_model(dtype)
    m.image_at_head = True
    ids, mask, labels = e["input_ids"], e["attention_mask"], e["labels"]
    out = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), labels=labels.cuda())
    _check_logits("head_labels", out.logits, e["logits"], dtype)
    d = abs(float(out.loss) - float(e["loss"][0]))
    _report(f"edge head_labels [{dtype}]: loss {float(out.loss):.6f} vs reference {float(e['loss'][0]):.6f}")
    assert d <= (1e-4 if dtype == torch.float32 else 1e-2), d
    if dtype == torch.float32:
        toks = m.generate(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), max_new_tokens=e["generated"].shape[1],
                          do_sample=False, eos_token_id=None).cpu()
        assert torch.equal(toks, e["generated"]), (toks, e["generated"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_edge_past_key_values_pass_through_matches_reference(dtype, golden_dir):
    e = _edge(golden_dir, "cache")
    cfg, m, px, ids, mask = _edge_model(dtype)
    first = m.forward(input_ids=ids[:, :-2].cuda(), pixel_values=px.cuda(), attention_mask=mask[:, :-2].cuda(), use_cache=True)
    _check_logits("cache(prompt)", first.logits, e["prompt_logits"], dtype)
    s1 = m.forward(input_ids=ids[:, -2:-1].cuda(), attention_mask=mask[:, :-1].cuda(), past_key_values=first.past_key_values, use_cache=True)
    _check_logits("cache(step 1)", s1.logits, e["step1_logits"], dtype)
    s2 = m.forward(input_ids=ids[:, -1:].cuda(), attention_mask=mask.cuda(), past_key_values=s1.past_key_values, use_cache=True)
    _check_logits("cache(step 2)", s2.logits, e["step2_logits"], dtype)
    assert s2.past_key_values.get_seq_length() == ids.shape[1]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_edge_row_without_an_image_slot_matches_reference(dtype, golden_dir):
    e = _edge(golden_dir, "mixed_rows")
    cfg, m, px, _, _ = _edge_model(dtype)
    ids, mask = e["input_ids"], e["attention_mask"]
    out = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda())
    _check_logits("mixed_rows", out.logits, e["logits"], dtype)
    if dtype == torch.float32:
        toks = m.generate(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), max_new_tokens=e["generated"].shape[1],
                          do_sample=False, eos_token_id=None).cpu()
        assert torch.equal(toks, e["generated"]), (toks, e["generated"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_edge_masks_with_interior_zeros_forward_matches_reference(dtype, golden_dir):
    """Zeros BETWEEN visible tokens in `forward`: the reference never forwards position_ids (modeling_visualcla.py:321-328), HF rotates by arange
    positions and the mask only removes keys -- computed exactly so here (round 4 refused it).  Fixtures made by the reference itself:
    image_at_head=True with a left-padded text mask = [1] * Q ++ [0, 0, 0, 1, ...] (logits and loss), and a text-only prompt with holes."""
    e = _edge(golden_dir, "head_leftpad")
    cfg, m, px, _, _ = _edge_model(dtype)
    m.image_at_head = True
    ids, mask, labels = e["input_ids"], e["attention_mask"], e["labels"]
    out = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), labels=labels.cuda())
    _check_logits("head_leftpad", out.logits, e["logits"], dtype)
    d = abs(float(out.loss) - float(e["loss"][0]))
    _report(f"edge head_leftpad [{dtype}]: loss {float(out.loss):.6f} vs reference {float(e['loss'][0]):.6f}")
    assert d <= (1e-4 if dtype == torch.float32 else 1e-2), d
    m.image_at_head = False
    e = _edge(golden_dir, "text_hole")
    ids, mask = e["input_ids"], e["attention_mask"]
    out = m.forward(input_ids=ids.cuda(), attention_mask=mask.cuda())
    _check_logits("text_hole", out.logits, e["logits"], dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_edge_beam_search_matches_reference(dtype, golden_dir):
    """generate(num_beams > 1) (round 4 raised; the reference forwards it to HF generate, modeling_visualcla.py:382-391): the reference's own outputs for
    (a) 3 beams without an eos id, (b) 3 beams with an eos id the beams do produce (hypotheses finish early, the length penalty ranks them, the returned
    row is filled with the eos id), (c) 4 beams, early_stopping=True, length_penalty 0.6, two returned hypotheses per prompt.  fp32 mode: the ids of the
    fixture; bf16 mode: the same calls run (every decode step goes through the B * num_beams-row kernels, the prompt is prefilled once and broadcast to its beams, the generated cache positions are gathered between steps)
    and return well-formed hypotheses."""
    e = _edge(golden_dir, "beams")
    cfg, m, px, ids, mask = _edge_model(dtype)
    assert torch.equal(ids, e["input_ids"])
    eos = int(e["eos"][0])
    kw = dict(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), do_sample=False, pad_token_id=0)
    a = m.generate(num_beams=3, max_new_tokens=6, eos_token_id=None, **kw).cpu()
    b = m.generate(num_beams=3, max_new_tokens=8, eos_token_id=eos, **kw).cpu()
    c = m.generate(num_beams=4, max_new_tokens=8, eos_token_id=eos, early_stopping=True, num_return_sequences=2, length_penalty=0.6, **kw).cpu()
    assert a.shape == e["a_generated"].shape and c.shape[0] == 4
    if dtype == torch.float32:
        assert torch.equal(a, e["a_generated"]), (a, e["a_generated"])
        assert torch.equal(b, e["b_generated"]), (b, e["b_generated"])
        assert torch.equal(c, e["c_generated"]), (c, e["c_generated"])
    greedy = m.generate(max_new_tokens=6, eos_token_id=None, **kw).cpu()
    assert greedy.shape == (2, 6)                              # num_beams = 1 is untouched by the beam path


def test_request_check_kernel_equals_the_tensor_formulation_on_random_requests():
    """vcla_check_request (ONE launch in front of every forward / generate) against the same five answers in tensor algebra
    (tests/repl_stub/oracle_backed.py:_request_flags, what the CPU tests of `_check_request` run on): 300 random requests over batch sizes, row lengths either
    side of the 64-lane chunks, well-formed / truncated / missing image slots, ids and labels at both edges of the vocabulary, masks with left padding, right
    padding, holes and nothing visible at all, with and without the image_at_head prefix.  Flags and img_pos must be EQUAL."""
    import sys as _sys
    from types import SimpleNamespace
    _sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "repl_stub"))
    from oracle_backed import OracleBackedModel
    import visualcla.modeling_visualcla as M
    V, S_ID, E_ID, T_ID = 1000, 997, 998, 999
    me = SimpleNamespace(config=SimpleNamespace(text_config={"vocab_size": V}))
    g = torch.Generator().manual_seed(11)
    r = lambda n: int(torch.randint(0, n, (1,), generator=g))
    seen = [0] * 5
    for case in range(300):
        B, T = (1, 3, 70)[r(3)], (1, 7, 24, 63, 64, 65, 130, 257)[r(8)]
        Q = (0, 4, 16)[r(3)]
        ids = torch.randint(0, 990, (B, T), generator=g)
        if Q and T > Q + 2:
            for b in range(B):
                kind = r(6)                                            # 0: no slot, 1 - 3: well-formed, 4: </img> missing, 5: <img> too close to the end
                if kind == 0:
                    continue
                p0 = r(T - Q - 1) if kind != 5 else T - 1 - r(min(Q, T - 1) + 1)
                ids[b, p0] = S_ID
                if p0 + 1 < T:
                    ids[b, p0 + 1:min(T, p0 + 1 + Q)] = T_ID
                if kind in (1, 2, 3) and p0 + Q + 1 < T:
                    ids[b, p0 + Q + 1] = E_ID
                if kind == 3 and p0 + Q + 2 < T:
                    ids[b, p0 + Q + 2 + r(T - p0 - Q - 2)] = S_ID      # a second <img> later in the row: the first one counts
                if r(8) == 0:
                    ids[b][ids[b] == T_ID] = 5                         # no <img_token>: forward lets the row pass as image-free
        if r(10) == 0:
            ids[r(B), r(T)] = (V, -1, V + 5)[r(3)]
        am = None
        if r(4):
            Tm = T if r(3) else T + Q
            am = torch.ones(B, Tm, dtype=torch.int64)
            for b in range(B):
                kind = r(7)                                            # 0 - 1: all ones, 2: left pad, 3: right pad, 4: hole, 5: nothing visible, 6: random
                if kind == 2:
                    am[b, :r(Tm + 1)] = 0
                elif kind == 3:
                    am[b, Tm - r(Tm + 1):] = 0
                elif kind == 4:
                    am[b, r(Tm)] = 0
                elif kind == 5:
                    am[b] = 0
                elif kind == 6:
                    am[b] = torch.randint(0, 2, (Tm,), generator=g)
        lab = None
        if r(3) == 0:
            lab = torch.randint(0, V, (B, T), generator=g)
            lab[:, :r(T + 1)] = -100
            if r(4) == 0:
                lab[r(B), r(T)] = (V, -1, -99)[r(3)]
        need_tok, prefix = bool(r(2)), bool(r(2))
        args = (Q, (S_ID, E_ID, T_ID), need_tok, prefix)
        want, want_pos = OracleBackedModel._request_flags(me, ids, am, lab, *args)
        got, got_pos = M.VisualCLAModel._request_flags(me, ids.cuda(), None if am is None else am.cuda(), None if lab is None else lab.cuda(), *args)
        assert got == want, (case, B, T, Q, got, want)
        if Q:
            assert got_pos.dtype == torch.int32 and torch.equal(got_pos.cpu(), want_pos), (case, got_pos.tolist(), want_pos.tolist())
        for i in range(5):
            seen[i] += int(want[i])
    assert all(n >= 5 for n in seen), seen                             # every flag was raised by some requests (and left down by most)


def test_masks_with_interior_zeros_are_refused_by_generate_only():
    """image_at_head=True + a left-padded text mask = [1]*Q ++ [0..0, 1..1] (modeling_visualcla.py:372-377): in `generate` the transformers
    versions the reference pins derive cumsum(mask) positions, which zeros between visible tokens would make differ from the absolute
    positions used here -> ValueError, not a silently different result.  `forward` computes it (test above); left- and right-padded masks
    are accepted everywhere."""
    cfg = O.cfg_tiny()
    W = O.make_weights(cfg, seed=0)
    px, _, _ = O.make_inputs(cfg, 2, 24)
    ids = torch.randint(3, 300, (2, 10))
    mask = torch.ones_like(ids)
    mask[1, :3] = 0
    m = make_hip_model(cfg, W, torch.float32)
    m.image_at_head = True
    m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda())
    with pytest.raises(ValueError, match="between visible tokens"):
        m.generate(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), max_new_tokens=2, do_sample=False)
    m.image_at_head = False
    txt_mask = torch.ones_like(ids)
    txt_mask[0, 4] = 0                                   # a hole in a text-only prompt
    m.forward(input_ids=ids.cuda(), attention_mask=txt_mask.cuda())
    with pytest.raises(ValueError, match="between visible tokens"):
        m.generate(input_ids=ids.cuda(), attention_mask=txt_mask.cuda(), max_new_tokens=2, do_sample=False)
    right = torch.ones_like(ids)
    right[1, 7:] = 0                                     # right padding: fine
    m.forward(input_ids=ids.cuda(), attention_mask=right.cuda())
    m.forward(input_ids=ids.cuda(), attention_mask=mask.cuda())      # left padding without the image prefix: fine
    lab = ids.clone()
    lab[0, 3] = cfg.text.vocab_size                      # labels are validated with the request, before any kernel runs
    with pytest.raises(ValueError, match="labels contain ids outside"):
        m.forward(input_ids=ids.cuda(), attention_mask=right.cuda(), labels=lab.cuda())
    with pytest.raises(ValueError, match="do not match"):
        m.forward(input_ids=ids.cuda(), attention_mask=right.cuda(), labels=lab[:, :-1].cuda())


@pytest.mark.parametrize("B", [2, 16, 64])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_single_token_forward_on_a_cache_matches_the_full_forward(B, dtype):
    """forward(input_ids[B, 1], past_key_values=cache) -- the reference's forward contract (modeling_visualcla.py:264-330 passes
    past_key_values through) -- against the full-sequence forward and the oracle.  2 <= B <= 64 single-token rows is exactly
    the shape the streaming DECODE kernels serve: the prefill entry point must not take that branch (its tail expects a
    row-major, normalised w.h)."""
    cfg = O.cfg_small()
    W = O.make_weights(cfg, seed=0)
    T = 41
    px, ids, mask = O.make_inputs(cfg, B, T + 1, n_prefix=4)
    m = make_hip_model(cfg, W, dtype)
    first = m.forward(input_ids=ids[:, :T].cuda(), pixel_values=px.cuda(), use_cache=True)
    cache = first.past_key_values
    assert cache is not None and cache.get_seq_length() == T
    taps = {}
    step = m.forward(input_ids=ids[:, T:].cuda(), past_key_values=cache, use_cache=True, taps=taps)
    assert step.logits.shape == (B, 1, cfg.text.vocab_size) and cache.get_seq_length() == T + 1
    full = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda()).logits[:, -1].float().cpu()
    ref = O.visualcla_forward(ids[:4], px[:4], mask[:4], W, cfg)[:, -1]
    got = step.logits[:, 0].float().cpu()
    tol = 1e-3 if dtype == torch.float32 else 6e-2
    e_full, e_ref = (got - full).abs().max().item(), (got[:4] - ref[: min(B, 4)]).abs().max().item()
    _report(f"single-token forward on a cache [B={B}, {dtype}]: vs full forward {e_full:.3e}, vs oracle {e_ref:.3e}")
    assert e_full <= tol and e_ref <= tol, (e_full, e_ref)
    assert torch.isfinite(taps["final_norm"].float()).all()


def test_prefix_allowed_tokens_fn_constrains_generation():
    """the reference forwards `prefix_allowed_tokens_fn` to HF generate (modeling_visualcla.py:382-391): every new token must come
    from the allowed set, and an unconstrained call is unchanged"""
    cfg = O.cfg_tiny()
    W = O.make_weights(cfg, seed=0)
    px, ids, mask = O.make_inputs(cfg, 2, 24)
    m = make_hip_model(cfg, W, torch.float32)
    allowed = [5, 7, 11, 13]
    kw = dict(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), max_new_tokens=5, do_sample=False, eos_token_id=None)
    got = m.generate(prefix_allowed_tokens_fn=lambda batch_id, sent: allowed, **kw).cpu()
    assert got.shape == (2, 5) and bool(torch.isin(got, torch.tensor(allowed)).all())
    # oracle: greedy over the allowed ids only
    want = O.visualcla_generate(ids, px, mask, W, cfg, max_new_tokens=5,
                                select_fn=lambda lg, gen: torch.tensor(allowed)[lg[:, allowed].argmax(-1)])
    assert torch.equal(got, want)
    free = m.generate(**kw).cpu()
    assert torch.equal(free, O.visualcla_generate(ids, px, mask, W, cfg, max_new_tokens=5))
    with pytest.raises(ValueError, match="do_sample=False"):          # beam SAMPLING is refused; beam search itself: test_edge_beam_search_matches_reference
        m.generate(num_beams=2, **dict(kw, do_sample=True))
    with pytest.raises(ValueError, match="num_return_sequences"):
        m.generate(num_return_sequences=2, **kw)


def test_generation_config_fields_reach_the_decode_loop_or_are_refused():
    """the reference forwards its whole generation config to HF generate (modeling_visualcla.py:382-391): the length rules of the `inputs_embeds`
    case, config-selected processors outside the device sampler's set (host-driven steps, transformers' classes in transformers' order:
    tests/test_host_cpu.py::test_logits_processors_equal_transformers_on_random_configs), `min_length` on the device-resident loop, and the
    refusal BY NAME of what is not implemented"""
    from transformers.generation.logits_process import NoBadWordsLogitsProcessor
    cfg = O.cfg_tiny()
    W = O.make_weights(cfg, seed=0)
    px, ids, mask = O.make_inputs(cfg, 2, 24)
    m = make_hip_model(cfg, W, torch.float32)
    base = dict(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), do_sample=False, eos_token_id=None)
    want = O.visualcla_generate(ids, px, mask, W, cfg, max_new_tokens=6)
    T = ids.shape[1]
    # an explicit max_length counts the prompt; with nothing left HF raises
    assert torch.equal(m.generate(max_length=T + 6, **base).cpu(), want)
    assert torch.equal(m.generate(max_length=T + 6, max_new_tokens=3, **base).cpu(), want[:, :3])
    with pytest.raises(ValueError, match="max_length"):
        m.generate(max_length=T, **base)
    # bad_words_ids: the greedy path's own first tokens are banned -> the config field and the same processor passed by hand agree, the ids are gone
    banned = [[int(t)] for t in want[:, 0].unique()]
    got = m.generate(max_new_tokens=6, bad_words_ids=banned, **base).cpu()
    by_hand = m.generate(max_new_tokens=6, logits_processor=[NoBadWordsLogitsProcessor(banned, None)], **base).cpu()
    flat = torch.tensor([b[0] for b in banned])
    assert torch.equal(got, by_hand) and not bool(torch.isin(got, flat).any()) and not torch.equal(got, want)
    assert torch.equal(got, O.visualcla_generate(ids, px, mask, W, cfg, max_new_tokens=6,
                                                 select_fn=lambda lg, gen: lg.index_fill(1, flat, float("-inf")).argmax(-1)))
    # suppress_tokens likewise; forced_eos_token_id puts the eos id at the last position of the budget
    sup = m.generate(max_new_tokens=4, suppress_tokens=[int(t) for t in flat], **base).cpu()
    assert torch.equal(sup, got[:, :4])
    eos = int(want[0, 5]) if int(want[0, 5]) not in want[0, :5].tolist() else 3
    forced = m.generate(max_new_tokens=4, forced_eos_token_id=eos, **dict(base, eos_token_id=eos, pad_token_id=0)).cpu()
    assert forced.shape[1] <= 4 and bool(((forced == eos).sum(dim=1) >= 1).all())
    # min_length (less the prompt) masks the eos id on the device-resident loop exactly as min_new_tokens does
    stop = int(want[0, 1])
    a = m.generate(max_new_tokens=6, min_length=T + 4, **dict(base, eos_token_id=stop, pad_token_id=0)).cpu()
    b = m.generate(max_new_tokens=6, min_new_tokens=4, **dict(base, eos_token_id=stop, pad_token_id=0)).cpu()
    assert torch.equal(a, b) and not bool((a[:, :4] == stop).any())
    # several sampled sequences per prompt (rows b * n .. + n - 1 = prompt b); top_k = 1 makes every draw the greedy token
    rep = m.generate(max_new_tokens=6, num_return_sequences=3, top_k=1, **dict(base, do_sample=True)).cpu()
    assert rep.shape == (6, 6) and torch.equal(rep, want.repeat_interleave(3, dim=0))
    free = m.generate(max_new_tokens=6, num_return_sequences=3, temperature=3.0, **dict(base, do_sample=True)).cpu()     # top_k: the global default 50
    assert free.shape == (6, 6) and (not torch.equal(free[0], free[1]) or not torch.equal(free[1], free[2]))
    # refused by name, never dropped
    for kw, word in ((dict(return_dict_in_generate=True), "return_dict_in_generate"), (dict(penalty_alpha=0.5, top_k=4), "penalty_alpha"),
                     (dict(guidance_scale=2.0), "guidance_scale"), (dict(streamer=object()), "streamer"), (dict(max_new_token=3), "max_new_token")):
        with pytest.raises(ValueError, match=word):
            m.generate(**dict(base, max_new_tokens=2, **kw))


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
    m._oracle_weights = None
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
    # every target size is interpolated from the checkpoint's own embedding: two steps == one step, and the native size comes back
    # bit for bit (bench.py runs its 336-px leg on the same model object and returns to 224 px)
    bigger = big + cfg.vision.patch_size
    W3 = {k: v.clone() for k, v in W.items()}
    extend_position_embedding(W3, cfg.vision.patch_size, bigger)
    pe = lambda sd: next(v for k, v in sd.items() if k.endswith("vision_model.embeddings.position_embedding.weight")).float().cpu()
    m.set_image_size(bigger)
    assert torch.equal(pe(m.state_dict()), pe(W3))
    m.set_image_size(cfg.vision.image_size)
    assert torch.equal(pe(m.state_dict()), pe(W))
    got0 = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda()).logits.float().cpu()
    assert (got0 - torch.from_numpy(g["logits"]).reshape(got0.shape)).abs().max().item() <= 1e-3
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
        m.enable_fp8_decode(True, prefill=False)    # fp8 WEIGHTS in the decode kernels; the prefill that fills the cache stays bf16
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


# ------------------------------------------------------------------ full VisualCLA-7B geometry: against the ORACLE (BASELINE shapes)
# Stated bounds of the bf16 product path against the fp32 oracle at depth 24 (ViT) / 32 (LLaMA), random-init weights,
# measured on MI355X (profiles/r02_parity_report.txt) and asserted with ~2x headroom:
# measured: ViT taps grow from 0.7 % (layer 0) to 2.0 % (layer 23) of the tap's dynamic range, mean error 3e-3 -> 2.6e-2;
# logits (std 1.29): max 0.45, mean 0.053 over 2 x 128 x 49958 values.  The same-dtype comparison below shows the reference
# itself, run in bf16, sits at the same distance from its fp32 self.
B7_VIT_REL = 4e-2        # every ViT / resampler tap: max |err| / max |ref|
B7_LOGIT_MAX = 0.9       # logits: max abs error
B7_LOGIT_MEAN = 0.1      # logits: mean abs error
B7_MARGIN = 0.5          # top-2 margin of the fp32 reference beyond which the bf16 argmax must agree (> the measured max error 0.45-0.49)


def _oracle_threads():
    torch.set_num_threads(min(32, os.cpu_count() or 1))   # one NUMA domain: torch's CPU kernels collapse on all SMT threads


def _w7(m):
    """the 7B weights as the oracle wants them (fp32 copies of the bf16-rounded values, ~27 GB of host RAM), unpacked once"""
    if getattr(m, "_oracle_weights", None) is None:
        m._oracle_weights = m.state_dict()
    return m._oracle_weights


def test_7b_vision_stack_matches_oracle_per_layer(model_7b):
    """a3-a7 at the BASELINE shapes: all 24 ViT layer taps, post-LN, 6 resampler taps and the projected image embeds of a
    B = 2 batch against the fp32 oracle on the same bf16-rounded weights"""
    m, ocfg = model_7b
    _oracle_threads()
    px, _, _ = O.make_inputs(ocfg, 2, 128)
    W = _w7(m)
    ref_t = {}
    with torch.no_grad():
        O.image_embeds(px, W, ocfg, ref_t)
    taps = {}
    m.embed_images(px.cuda(), taps)
    torch.cuda.synchronize()
    worst = 0.0
    for k, ref in ref_t.items():
        if k not in taps:
            continue
        got = taps[k].float().cpu().reshape(ref.shape)
        err = (got - ref).abs()
        rel = err.max().item() / max(ref.abs().max().item(), 1e-6)
        worst = max(worst, rel)
        _report(f"7B bf16 vs fp32 oracle {k}: max_abs_err={err.max().item():.3e} mean={err.mean().item():.3e} rel_to_absmax={rel:.3e}")
        assert rel <= B7_VIT_REL, (k, rel)
    assert len([k for k in ref_t if k in taps]) >= 24 + 1 + 6 + 1
    _report(f"7B vision stack: worst tap error {worst:.3e} of the tap's dynamic range (bound {B7_VIT_REL})")


def test_7b_prefill_and_decode_logits_match_oracle(model_7b):
    """a1 / a2 / a8-a12 at the BASELINE shapes (configs[1] geometry, B = 2, T = 128): full-sequence forward logits, the
    prefill's last-position logits and two KV-cache decode steps (teacher-forced with the HIP path's own tokens) against the
    fp32 oracle; B = 2 runs the streaming batch-decode kernels, B = 1 the M = 1 GEMV path."""
    from transformers import LogitsProcessorList
    m, ocfg = model_7b
    _oracle_threads()
    B, T, n_new = 2, 128, 3
    px, ids, mask = O.make_inputs(ocfg, B, T)
    W = _w7(m)
    with torch.no_grad():
        img = O.image_embeds(px, W, ocfg)
        x = O.embed_and_splice(ids, img, W, ocfg)
        cache = [None] * ocfg.text.num_hidden_layers
        h = O.llama_forward(x, W, ocfg.text, mask, cache, 0)
        ref_all = O.lm_head(h, W)                                        # [B, T, V]
    got_all = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda()).logits.float().cpu()
    err = (got_all - ref_all).abs()
    _report(f"7B forward logits [B={B},T={T}] vs fp32 oracle: max {err.max().item():.3e} mean {err.mean().item():.3e} (ref std {ref_all.std().item():.3f})")
    assert err.max().item() <= B7_LOGIT_MAX and err.mean().item() <= B7_LOGIT_MEAN
    top2 = ref_all.topk(2, dim=-1).values
    decided = (top2[..., 0] - top2[..., 1]) > B7_MARGIN
    agree = got_all.argmax(-1) == ref_all.argmax(-1)
    assert bool(agree[decided].all())
    _report(f"7B forward: argmax agrees at all {int(decided.sum())} positions whose fp32 top-2 margin exceeds {B7_MARGIN}, and at "
            f"{int(agree.sum())} of {B * T} positions overall (random-init logits: median margin {float((top2[..., 0] - top2[..., 1]).median()):.3f})")
    # SURVEY section 7 tolerance (ii) at the BASELINE shape: the reference arithmetic run in bf16 on the host (what the HF modules
    # compute under .to(bfloat16)) against its own fp32 self, sample 0 -- the HIP path must not be further from fp32 than that
    t0_ = __import__("time").time()
    with torch.no_grad():
        Wb = {k: v.to(torch.bfloat16) for k, v in W.items() if k.startswith("text_model.")}
        hb = O.llama_forward(x[:1].to(torch.bfloat16), Wb, ocfg.text, mask[:1], None, 0)
        hf_all = O.lm_head(hb, Wb).float()
        del Wb
    e_hf, e_hip = (hf_all - ref_all[:1]).abs(), (got_all[:1] - ref_all[:1]).abs()
    _report(f"7B same-dtype (decoder, sample 0, T={T}): HIP-bf16 vs fp32 max {e_hip.max().item():.3e} mean {e_hip.mean().item():.3e} | "
            f"oracle-in-bf16 vs fp32 max {e_hf.max().item():.3e} mean {e_hf.mean().item():.3e}  ({__import__('time').time() - t0_:.0f}s of host time)")
    assert e_hip.mean().item() <= 1.5 * e_hf.mean().item() + 5e-3 and e_hip.max().item() <= 1.5 * e_hf.max().item() + 5e-2
    for nb in (2, 1):
        seen = []

        def grab(ids_, scores):
            seen.append(scores.detach().float().cpu().clone())
            return scores
        toks = m.generate(input_ids=ids[:nb].cuda(), pixel_values=px[:nb].cuda(), attention_mask=mask[:nb].cuda(), max_new_tokens=n_new,
                          do_sample=False, eos_token_id=None, logits_processor=LogitsProcessorList([grab])).cpu()
        with torch.no_grad():
            cache = [None] * ocfg.text.num_hidden_layers
            h = O.llama_forward(x[:nb], W, ocfg.text, mask[:nb], cache, 0)
            refs = [O.lm_head(h[:, -1:], W)[:, 0]]
            past = T
            for s_ in range(n_new - 1):
                e = W["text_model.model.embed_tokens.weight"][toks[:, s_]][:, None, :]
                h = O.llama_forward(e, W, ocfg.text, torch.ones(nb, past + 1, dtype=torch.int64), cache, past)
                refs.append(O.lm_head(h, W)[:, 0])
                past += 1
        for s_ in range(n_new):
            e_ = (seen[s_] - refs[s_]).abs()
            _report(f"7B B={nb} {'prefill' if s_ == 0 else f'decode step {s_}'} logits vs fp32 oracle: max {e_.max().item():.3e} mean {e_.mean().item():.3e}")
            assert e_.max().item() <= B7_LOGIT_MAX and e_.mean().item() <= B7_LOGIT_MEAN, (nb, s_, e_.max().item())


def test_7b_fp32_mode_logits_are_within_1e3_of_the_oracle(model_7b):
    """north_star's tolerance AT THE BASELINE SHAPE (configs[1] geometry: VisualCLA-7B, B = 1, T = 128 with the 64 image slots):
    the fp32 activation mode of the HIP path -- fp32 storage between the kernels, the same bf16-rounded weights as the oracle --
    must give the full-sequence logits within 1e-3 ABS of the fp32 oracle (measured 1.7e-4), and so must two single-token forwards on
    the returned cache; every stage tap within 1e-3 of its dynamic range (all 24 ViT layers, the resampler, the projection and the
    splice are also within 1e-3 absolute; the report names the first hidden-state tap past 1e-3 absolute, llama_layer22 at |x| ~ 50)."""
    import visualcla
    m16, ocfg = model_7b
    _oracle_threads()
    W = _w7(m16)
    B, T = 1, 128
    px, ids, mask = O.make_inputs(ocfg, B, T)
    m = visualcla.VisualCLAModel.from_state_dict(m16.config, W, device="cuda:0", torch_dtype=torch.float32)
    m.tokenizer, m.image_at_head = stub_tokenizer(ocfg), False
    try:
        ref_t = {}
        cache = [None] * ocfg.text.num_hidden_layers
        with torch.no_grad():
            ref = O.visualcla_forward(ids, px, mask, W, ocfg, taps=ref_t, cache=cache)
        taps = {}
        out = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), use_cache=True, taps=taps)
        torch.cuda.synchronize()
        first_bad, first_abs, worst = None, None, 0.0
        order = (["vit_embed"] + [f"vit_layer{i}" for i in range(ocfg.vision.num_hidden_layers)] + ["vit_post_ln"] +
                 [f"resampler_layer{i}" for i in range(ocfg.resampler.num_hidden_layers)] + ["image_embeds", "spliced_embeds"] +
                 [f"llama_layer{i}" for i in range(ocfg.text.num_hidden_layers)] + ["final_norm", "logits"])
        for k in order:
            r = ref_t[k].float()
            e = (taps[k].float().cpu().reshape(r.shape) - r).abs().max().item()
            worst = max(worst, e)
            if k in ("vit_embed", "vit_layer23", "vit_post_ln", "resampler_layer5", "image_embeds", "llama_layer0", "llama_layer15", "llama_layer31", "final_norm", "logits"):
                _report(f"7B fp32 mode vs fp32 oracle {k}: max_abs_err={e:.3e} (ref absmax {r.abs().max().item():.2e})")
            # north_star's 1e-3 is a bound on the LOGITS.  Hidden-state taps are held to it relative to their dynamic range: the LLaMA
            # residual stream of a random-init 32-layer network grows to |x| ~ 66, where 1e-3 absolute would be 1.5e-5 relative (~130
            # fp32 ulps accumulated over 32 layers and two reduction orders); the first stage past 1e-3 ABSOLUTE is reported below.
            bound = 1e-3 * max(1.0, r.abs().max().item())
            if e > 1e-3 and first_abs is None:
                first_abs = (k, e)
            if e > bound and first_bad is None:
                first_bad = (k, e)
        e_log = (out.logits.float().cpu() - ref).abs()
        _report(f"7B fp32 mode [B={B}, T={T}]: logits max_abs_err={e_log.max().item():.3e} mean={e_log.mean().item():.3e} (ref std {ref.std().item():.3f}); "
                f"worst stage abs err {worst:.3e}; first stage over 1e-3 x its dynamic range: {first_bad}; first hidden-state tap over 1e-3 absolute: {first_abs}")
        assert first_bad is None and e_log.max().item() <= 1e-3, (first_bad, e_log.max().item())
        assert torch.equal(out.logits.argmax(-1).cpu(), ref.argmax(-1))
        # two decode steps on the cache (single-token forward: the fp32 GEMV path), teacher-forced with the oracle's greedy ids
        kv, past = out.past_key_values, T
        nxt = ref[:, -1].argmax(-1)
        for step in range(2):
            full_mask = torch.ones(B, past + 1, dtype=torch.int64)
            with torch.no_grad():
                r = O.visualcla_forward(nxt[:, None], None, full_mask, W, ocfg, cache=cache, past_len=past)
            o = m.forward(input_ids=nxt[:, None].cuda(), attention_mask=full_mask.cuda(), past_key_values=kv, use_cache=True)
            e = (o.logits.float().cpu() - r).abs().max().item()
            _report(f"7B fp32 mode decode step {step + 1} (context {past + 1}): logits max_abs_err={e:.3e}")
            assert e <= 1e-3, (step, e)
            kv, past, nxt = o.past_key_values, past + 1, r[:, -1].argmax(-1)
    finally:
        del m
        torch.cuda.empty_cache()


@pytest.mark.parametrize("name", sorted(CASES))
def test_bf16_path_is_no_worse_than_the_reference_run_in_bf16(name, golden_dir):
    """SURVEY section 7, tolerance (ii): the product dtype against the SAME-dtype reference.  The oracle evaluated in bf16
    (= what the HF modules do under .to(bfloat16)) has its own distance to the fp32 reference; the HIP path (bf16 storage,
    fp32 accumulation) must not be further away than that (x1.5 + a small absolute floor), stage by stage and on the logits."""
    g, cfg, W, px, ids, mask, n_new = _setup(name, golden_dir)
    hf_t = {}
    with torch.no_grad():
        hf_logits = O.visualcla_forward(ids, px, mask, W, cfg, dtype=torch.bfloat16, taps=hf_t).float()
    m = make_hip_model(cfg, W, torch.bfloat16)
    taps = {}
    out = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), taps=taps)
    ref = torch.from_numpy(g["logits"])
    e_hf, e_hip = (hf_logits - ref).abs(), (out.logits.float().cpu() - ref).abs()
    _report(f"same-dtype {name} logits: HIP-bf16 max {e_hip.max().item():.3e} mean {e_hip.mean().item():.3e} | oracle-in-bf16 max {e_hf.max().item():.3e} mean {e_hf.mean().item():.3e}")
    assert e_hip.max().item() <= 1.5 * e_hf.max().item() + 5e-3 and e_hip.mean().item() <= 1.5 * e_hf.mean().item() + 1e-3
    for k in g.files:
        if k.startswith("_") or k in ("generated", "vit_embed", "logits") or k not in hf_t or k not in taps:
            continue
        r = torch.from_numpy(g[k])
        a_ = (hf_t[k].float().reshape(r.shape) - r).abs().mean().item()
        b_ = (taps[k].float().cpu().reshape(r.shape) - r).abs().mean().item()
        _report(f"same-dtype {name} {k}: HIP-bf16 mean err {b_:.3e} | oracle-in-bf16 {a_:.3e}")
        assert b_ <= 1.5 * a_ + 1e-3 * max(r.abs().max().item(), 1.0), (k, b_, a_)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_generate_on_a_left_padded_batch_matches_oracle(dtype):
    """a batched chat(): sample 1 is left-padded; the reference (under the installed transformers) keeps arange positions and
    masks the pad keys -- greedy ids must equal the oracle's in fp32 mode, and wherever the margin allows in bf16 mode"""
    cfg = O.cfg_small()
    W = O.make_weights(cfg, seed=0)
    px, ids, mask = O.make_inputs(cfg, 3, 48, n_prefix=6)
    for b, npad in ((1, 4), (2, 1)):
        ids[b] = torch.cat([torch.zeros(npad, dtype=ids.dtype), ids[b, :-npad]])
        mask[b, :npad] = 0
    n_new = 6
    want, ref_logits = O.visualcla_generate(ids, px, mask, W, cfg, max_new_tokens=n_new, return_logits=True)
    m = make_hip_model(cfg, W, dtype)
    for use_graph in (False, True):
        got = m.generate(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), max_new_tokens=n_new, do_sample=False,
                         eos_token_id=None, use_graph=use_graph).cpu()
        if dtype == torch.float32:
            assert torch.equal(got, want), (got, want)
        else:
            alive = torch.ones(3, dtype=torch.bool)
            for s_ in range(n_new):
                t2 = ref_logits[s_].topk(2, dim=-1).values
                alive &= (t2[:, 0] - t2[:, 1]) > 0.12
                assert bool((got[:, s_] == want[:, s_])[alive].all()), (s_, got, want)
    _report(f"left-padded generate [{dtype}]: ids == oracle")


def test_fp8_mfma_prefill_against_oracle_on_dequantised_weights(golden_dir):
    """BASELINE configs[4]: with fp8 copies loaded, a prefill of more than 128 rows runs fp8 x fp8 on the fp8 MFMA pipe.  Against
    the oracle on the DEQUANTISED weights the remaining error is the per-row activation quantisation (e4m3: 3 mantissa bits);
    against the bf16-weight oracle it is weights + activations.  Stated bounds: max 0.25 / mean 0.03 on logits of std ~0.4."""
    from visualcla.weights import quantize_fp8_rows, dequantize_fp8_rows, pad_to
    g, cfg, W, px, ids, mask, n_new = _setup("small_b2", golden_dir)
    B, T = 4, 48                                                        # 192 rows > 128 -> the MFMA tile kernels
    px, ids, mask = O.make_inputs(cfg, B, T)
    m = make_hip_model(cfg, W, torch.bfloat16)
    base = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda()).logits.float().cpu()
    m.enable_fp8_decode()                                               # prefill=True: fp8 MFMA for M > 128
    got = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda()).logits.float().cpu()
    assert not torch.equal(got, base)                                   # the fp8 path really ran
    Wq = dict(W)
    for k, v in W.items():
        if k.startswith("text_model.model.layers.") and v.dim() == 2:
            wp = torch.zeros(pad_to(v.shape[0], 128), v.shape[1])
            wp[: v.shape[0]] = v
            q, sc = quantize_fp8_rows(wp.to(torch.bfloat16))
            Wq[k] = dequantize_fp8_rows(q, sc)[: v.shape[0]]
    ref_q = O.visualcla_forward(ids, px, mask, Wq, cfg)
    ref = O.visualcla_forward(ids, px, mask, W, cfg)
    e_q, e_w = (got - ref_q).abs(), (got - ref).abs()
    _report(f"fp8 MFMA prefill [B={B},T={T}] logits: vs oracle on dequantised weights max {e_q.max().item():.3e} mean {e_q.mean().item():.3e}; "
            f"vs bf16-weight oracle max {e_w.max().item():.3e} mean {e_w.mean().item():.3e}; bf16 path vs oracle max {(base - ref).abs().max().item():.3e} (logit std {ref.std().item():.3f})")
    assert e_q.max().item() <= 0.25 and e_q.mean().item() <= 0.03
    toks = m.generate(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), max_new_tokens=3, do_sample=False, eos_token_id=None)
    assert toks.shape == (B, 3)
    m.enable_fp8_decode(True, prefill=False)                            # decode-only fp8: prefill back on bf16 MFMA
    again = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda()).logits.float().cpu()
    assert torch.equal(again, base)


def test_7b_fp8_mfma_prefill_error_is_bounded(model_7b):
    """BASELINE configs[4] at the 7B shape: the prefill on the fp8 MFMA pipe (e4m3 weights AND per-row-quantised e4m3 activations,
    32 layers deep) against the bf16 prefill of the same model.  W8A8 with 3 mantissa bits is a lossy mode by construction: every
    GEMM output carries ~5 % relative noise (3.6 % rms per operand), and a RANDOM-INIT 32-layer network amplifies any per-op
    perturbation about 10x end to end (the bf16 path itself: 0.4 % per op -> 4 % of the logit spread, see the fp32-oracle test
    above), so ~40-50 % is the expected figure here; measured: mean 0.38 x logit std, cosine 0.88.  Bounds with headroom; the
    per-GEMM exactness (function of the dequantised operands) is pinned in tests/test_gpu_kernels.py::test_gemm_fp8_mfma."""
    m, ocfg = model_7b
    B, T = 2, 128
    px, ids, mask = O.make_inputs(ocfg, B, T)
    px, ids, mask = px.cuda(), ids.cuda(), mask.cuda()
    base = m.forward(input_ids=ids, pixel_values=px, attention_mask=mask).logits.float()
    m.enable_fp8_decode()                 # prefill=True
    try:
        got = m.forward(input_ids=ids, pixel_values=px, attention_mask=mask).logits.float()
    finally:
        m.enable_fp8_decode(False)
    assert torch.isfinite(got).all() and not torch.equal(got, base)
    err = (got - base).abs()
    std = base.std().item()
    cos = torch.nn.functional.cosine_similarity(got.flatten(), base.flatten(), dim=0).item()
    _report(f"7B fp8 MFMA prefill vs bf16 prefill [B={B},T={T}]: logits max err {err.max().item():.3e} mean {err.mean().item():.3e} "
            f"(logit std {std:.3f}, cosine {cos:.4f})")
    assert err.mean().item() <= 0.55 * std and cos >= 0.8


FP8_MODE_BOUNDS = {
    # mode: (prefill on the fp8 MFMA pipe, e4m3 K/V cache) -> bounds (prefill cos >=, prefill mean/sigma <=, decode cos >=, decode mean/sigma <=)
    # measured on MI355X, random-init 7B (profiles/r04_parity_report.txt): w8a16 decode cos 0.963 / 0.218 sigma; +kv8 0.958 / 0.232;
    # w8a8+kv8 prefill 0.894 / 0.368, decode 0.906 / 0.346.  (A random-init 32-layer network amplifies per-op noise ~10x: the bf16 path
    # itself sits 0.04 sigma from fp32.)
    "w8a16": ((False, False), (0.99999, 1e-6, 0.94, 0.27)),
    "w8a16+kv8": ((False, True), (0.99999, 1e-6, 0.935, 0.28)),
    "w8a8+kv8": ((True, True), (0.85, 0.45, 0.87, 0.42)),
}


@pytest.mark.parametrize("mode", sorted(FP8_MODE_BOUNDS))
def test_7b_fp8_modes_error_vs_bf16(model_7b, mode):
    """BASELINE configs[4]'s numeric footing: every fp8 mode the benchmark can time, against the bf16 path of the same 7B model on the same
    inputs (bench.fp8_mode_accuracy: the function whose output the config4 line carries).  w8a16 = the `load_in_8bit` analogue (fp8 WEIGHTS
    in the decode kernels, dequantised in registers; the prefill runs the bf16 tiles -> its logits are bit-identical to bf16); +kv8 adds
    the e4m3 K/V cache; w8a8 = prefill on the fp8 MFMA pipe with per-row e4m3 activations as well (the lossy speed mode)."""
    import bench
    m, ocfg = model_7b
    (prefill, kv), bounds = FP8_MODE_BOUNDS[mode]
    acc = bench.fp8_mode_accuracy(m, prefill=prefill, kv_cache=kv)
    assert not m.fp8_decode
    _report(f"7B fp8 mode {mode} vs bf16: prefill cos {acc['prefill']['cosine']:.4f} mean {acc['prefill']['mean_over_sigma']:.4f} sigma | "
            f"decode steps cos {acc['decode_steps']['cosine']:.4f} mean {acc['decode_steps']['mean_over_sigma']:.4f} sigma")
    if not prefill:
        assert acc["prefill"]["cosine"] >= 0.99999 and acc["prefill"]["mean_over_sigma"] <= 1e-6      # the prefill IS the bf16 prefill
    if bounds is not None:
        pc, pm, dc, dm = bounds
        assert acc["prefill"]["cosine"] >= pc and acc["prefill"]["mean_over_sigma"] <= pm, acc
        assert acc["decode_steps"]["cosine"] >= dc and acc["decode_steps"]["mean_over_sigma"] <= dm, acc


class _DequantisedLlama(dict):
    """the oracle's weight dictionary with every LLaMA projection + lm_head replaced by dequant(quant_fp8_rows(W)) -- e4m3 bytes and per-row scales
    are taken from the model's OWN packed copies (`<name>.q8` / `.s8`, what the kernels read), kept as 1-byte tensors on the host (6.7 GB) and
    widened per access, so the 27 GB fp32 dictionary is not duplicated"""

    def __init__(self, base, model):
        super().__init__(base)
        t = model.config.text_config
        Dt, It, V = t["hidden_size"], t["intermediate_size"], t["vocab_size"]
        P = model._packed
        self._q, self._s = {}, {}

        def put(name, q, sc):
            self._q[name], self._s[name] = q.view(torch.float8_e4m3fn).cpu().contiguous(), sc.float().cpu().contiguous()
        for i in range(t["num_hidden_layers"]):
            s_, d = f"text_model.model.layers.{i}.", f"llama.l{i}."
            q, sc = P[d + "wqkv.q8"][:3 * Dt], P[d + "wqkv.s8"][:3 * Dt]
            for j, n in enumerate("qkv"):
                put(s_ + f"self_attn.{n}_proj.weight", q[j * Dt:(j + 1) * Dt], sc[j * Dt:(j + 1) * Dt])
            put(s_ + "self_attn.o_proj.weight", P[d + "wo.q8"][:Dt], P[d + "wo.s8"][:Dt])
            gq = P[d + "wgu.q8"][:2 * It].reshape(It // 16, 2, 16, Dt)          # gate / up rows interleaved in blocks of 16 (weights.interleave_gate_up)
            gs = P[d + "wgu.s8"][:2 * It].reshape(It // 16, 2, 16)
            put(s_ + "mlp.gate_proj.weight", gq[:, 0].reshape(It, Dt), gs[:, 0].reshape(It))
            put(s_ + "mlp.up_proj.weight", gq[:, 1].reshape(It, Dt), gs[:, 1].reshape(It))
            put(s_ + "mlp.down_proj.weight", P[d + "wd.q8"][:Dt], P[d + "wd.s8"][:Dt])
        put("text_model.lm_head.weight", P["llama.lm_head.q8"][:V], P["llama.lm_head.s8"][:V])

    def __getitem__(self, k):
        if k in self._q:
            return self._q[k].float() * self._s[k][:, None]
        return super().__getitem__(k)


def _quantised_rows(x):
    """per-row dynamic e4m3 quantisation of an activation matrix, dequantised again: what vcla_quant_fp8_rows + the fp8 MFMA compute on
    (scale = max(absmax / 448, 1e-20), values clamped to +-448, round-to-nearest-even into e4m3fn)"""
    xf = x.to(torch.bfloat16).float()                       # the kernel quantises the bf16 activation rows
    sc = (xf.abs().amax(dim=-1, keepdim=True) / 448.0).clamp_min(1e-20)
    return (xf * (1.0 / sc)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * sc      # x * (1 / scale), as the kernel (tests/test_gpu_kernels.py::test_quant_fp8_rows_matches_torch)


@pytest.mark.parametrize("B,rows", [(1, [0]), (2, [0, 1]), (256, [0, 100, 255])])
def test_7b_fp8_kernels_match_oracle_on_dequantised_weights(model_7b, B, rows):
    """BASELINE configs[4] at the 7B shape, FORMAT noise and KERNEL error separated: the W8A16 decode steps of a B-row generate()
    against the fp32 ORACLE EVALUATED ON THE DEQUANTISED WEIGHTS the kernels read -- the e4m3 format's own error is then in the reference too, and
    what is left is what the bf16 path also carries (bf16 activations), so the bf16 bounds apply unchanged.  B = 1: the M = 1 fp8 GEMV; B = 2: the
    fp8 streaming GEMMs (fragment-pair-major copies); B = 256: the ring kernel on the row-major e4m3 rows (129 - 256 decode rows: the N = 1 leg of
    configs[4], which read the bf16 matrices in round 4), rows {0, 100, 255}.  lm_head included.  The prefill of this mode is the bf16 prefill (its
    logits are compared with the oracle on the bf16 weights); bf16 K/V cache, so the cache format does not enter."""
    from transformers import LogitsProcessorList
    m, ocfg = model_7b
    _oracle_threads()
    T, n_new = 128, 3
    px, ids, mask = O.make_inputs(ocfg, B, T)
    W = _w7(m)
    m.enable_fp8_decode(True, prefill=False, kv_cache=False)
    try:
        Wq = _DequantisedLlama(W, m)
        seen = []

        def grab(ids_, scores):
            seen.append(scores[rows].detach().float().cpu().clone())
            return scores
        toks = m.generate(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), max_new_tokens=n_new, do_sample=False,
                          eos_token_id=None, logits_processor=LogitsProcessorList([grab])).cpu()
        loop = m.generate(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), max_new_tokens=n_new, do_sample=False,
                          eos_token_id=None).cpu()          # the device-resident hipGraph loop the benchmark times: same kernels
        assert torch.equal(loop, toks)
    finally:
        m.enable_fp8_decode(False)
    with torch.no_grad():
        img = O.image_embeds(px[rows], W, ocfg)
        x = O.embed_and_splice(ids[rows], img, W, ocfg)
        cache = [None] * ocfg.text.num_hidden_layers
        h = O.llama_forward(x, W, ocfg.text, mask[rows], cache, 0)                     # prefill: bf16 weights in this mode
        refs = [O.lm_head(h[:, -1:], W)[:, 0]]
        past = T
        for s_ in range(n_new - 1):                                                     # decode steps: dequantised e4m3 weights, lm_head included
            e = W["text_model.model.embed_tokens.weight"][toks[rows, s_]][:, None, :]
            h = O.llama_forward(e, Wq, ocfg.text, torch.ones(len(rows), past + 1, dtype=torch.int64), cache, past)
            refs.append(O.lm_head(h, Wq)[:, 0])
            past += 1
    for s_ in range(n_new):
        e_ = (seen[s_] - refs[s_]).abs()
        _report(f"7B W8A16 B={B} rows {rows} {'prefill (bf16 weights)' if s_ == 0 else f'decode step {s_} vs oracle on DEQUANTISED weights'}: "
                f"max {e_.max().item():.3e} mean {e_.mean().item():.3e} (logit std {refs[s_].std().item():.3f})")
        assert e_.max().item() <= B7_LOGIT_MAX and e_.mean().item() <= B7_LOGIT_MEAN, (B, s_, e_.max().item(), e_.mean().item())
        t2 = refs[s_].topk(2, dim=-1)
        decided = (t2.values[:, 0] - t2.values[:, 1]) > B7_MARGIN
        assert bool((seen[s_].argmax(-1) == t2.indices[:, 0])[decided].all())


def test_7b_fp8_mfma_prefill_matches_oracle_fed_the_same_quantised_operands(model_7b):
    """W8A8 at the 7B shape (the prefill on the fp8 MFMA pipe: e4m3 weights x per-row-quantised e4m3 activations) against the fp32 oracle fed the SAME
    kind of operands -- dequantised weights, and the input rows of every LLaMA projection quantised per row as vcla_quant_fp8_rows does.
    What this CAN and CANNOT separate (measured, profiles/r05_parity_report.txt): the kernel itself is exact on its operands -- pinned GEMM by GEMM at the
    LLaMA-7B shapes with the kernel's own codes in tests/test_gpu_kernels.py::test_gemm_fp8_mfma (max 3e-3).  One level up the two streams cannot stay
    identical: a quantiser turns a bf16-ulp difference in its INPUT (q / k / v and the attention output are bf16 in the HIP path, fp32 in the oracle) into
    whole-code flips on ~10 % of the elements, and flips of one e4m3 step on 10 % of a row inject about as much noise as the format's own rounding of all
    of it (power ~ perturbation x step vs step^2 / 12).  So, LAYER BY LAYER with the HIP path's own layer input (no amplification across layers), the
    distance to the same-operand oracle layer is required to be NO LARGER than the format's own activation noise (same oracle layer, quantised vs
    unquantised activations) -- a kernel-side slip (a wrong scale, a dropped K tail: tens of percent of the layer's delta) cannot hide under that, the
    format's noise itself is 5 - 7 % of the delta -- and END TO END the logits must sit closer to the same-operand oracle than to the plain one.
    Layers 0, 1, 15, 31: activation magnitudes from |x| ~ 1 to ~ 60.  lm_head runs on the bf16 weights in the prefill."""
    m, ocfg = model_7b
    _oracle_threads()
    B, T = 2, 128
    px, ids, mask = O.make_inputs(ocfg, B, T)
    W = _w7(m)
    base = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda()).logits.float().cpu()
    m.enable_fp8_decode(True, prefill=True, kv_cache=False)
    try:
        Wq = _DequantisedLlama(W, m)
        taps = {}
        got = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), taps=taps).logits.float().cpu()
        taps = {k: v.float().cpu() for k, v in taps.items() if k.startswith("llama_layer") or k == "spliced_embeds"}
    finally:
        m.enable_fp8_decode(False)
    orig_linear = O.linear

    def q_linear(x, w, b=None):                      # every LLaMA projection of the prefill: quantised rows x dequantised weights
        return orig_linear(_quantised_rows(x), w, b)
    positions = torch.arange(T)
    add_mask = O._causal_add_mask(T, 0, mask, torch.float32)
    with torch.no_grad():
        for l in (0, 1, 15, 31):
            x_in = taps["spliced_embeds"] if l == 0 else taps[f"llama_layer{l - 1}"]
            O.linear = q_linear
            try:
                ref_q = O.llama_layer(x_in, Wq, ocfg.text, l, positions, add_mask, None)
            finally:
                O.linear = orig_linear
            ref_u = O.llama_layer(x_in, Wq, ocfg.text, l, positions, add_mask, None)         # same weights, UNquantised activations
            out = taps[f"llama_layer{l}"]
            rng = ref_q.abs().max().item()
            delta = (ref_u - x_in).abs().mean().item()                      # what the layer adds to the stream
            noise = (ref_q - ref_u).abs().mean().item()                     # the format's own activation-quantisation noise on this layer's output
            e_q, e_u = (out - ref_q).abs(), (out - ref_u).abs()
            _report(f"7B W8A8 layer {l} (fed the HIP path's own input; layer delta mean {delta:.3f}, |x| max {rng:.1f}): vs oracle layer on the same kind of quantised operands "
                    f"max {e_q.max().item():.3e} mean {e_q.mean().item():.3e} ({e_q.mean().item() / delta:.3f} of the delta); vs the same layer with unquantised activations mean "
                    f"{e_u.mean().item():.3e}; the format's own noise (oracle quantised vs unquantised) mean {noise:.3e} ({noise / delta:.3f} of the delta)")
            assert e_q.mean().item() <= 1.15 * noise and e_q.mean().item() <= 0.1 * delta, (l, e_q.mean().item(), noise, delta)
        img = O.image_embeds(px, W, ocfg)
        x = O.embed_and_splice(ids, img, W, ocfg)
        O.linear = q_linear
        try:
            h = O.llama_forward(x, Wq, ocfg.text, mask, None, 0)
        finally:
            O.linear = orig_linear
        ref_q = O.lm_head(h, W)                          # lm_head: bf16 weights, unquantised rows
        ref = O.lm_head(O.llama_forward(x, W, ocfg.text, mask, None, 0), W)
    std = ref.std().item()
    e_q, e_f, e_b = (got - ref_q).abs(), (got - ref).abs(), (base - ref).abs()
    cos_q = torch.nn.functional.cosine_similarity(got.flatten(), ref_q.flatten(), dim=0).item()
    cos_f = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item()
    _report(f"7B W8A8 prefill [B={B},T={T}] logits end to end: vs oracle fed the same quantised operands max {e_q.max().item():.3e} mean {e_q.mean().item():.3e} "
            f"({e_q.mean().item() / std:.3f} sigma, cosine {cos_q:.4f}); vs plain fp32 oracle mean {e_f.mean().item():.3e} ({e_f.mean().item() / std:.3f} sigma, "
            f"cosine {cos_f:.4f}); bf16 path vs fp32 oracle mean {e_b.mean().item():.3e} (logit std {std:.3f})")
    assert e_q.mean().item() < e_f.mean().item() and cos_q > cos_f and cos_q >= 0.9, (e_q.mean().item(), e_f.mean().item(), cos_q, cos_f)


@pytest.mark.parametrize("B", [2, 64])
def test_7b_fp8_kv_cache_error_is_bounded(model_7b, B):
    """enable_fp8_decode(kv_cache=True): the K / V cache holds e4m3 bytes (unit scale), read by the decode steps (B = 2: the 4-wave
    form of the single-pass kernel, B = 64: the 2-wave batch form + fragment-major output); the prompt's own attention stays on exact
    bf16 rows.  Compared with the SAME weights (fp8 decode weights, bf16 prefill) on a bf16 cache: logits of the first decode steps of
    one greedy run, teacher-forced through the bf16-cache run's tokens.  A lossy mode by construction (3 mantissa bits on every cached
    key / value); stated bound with headroom over the measured figures (profiles/r03_parity_report.txt)."""
    from transformers import LogitsProcessorList
    m, ocfg = model_7b
    T, n_new = 128, 4
    px, ids, mask = O.make_inputs(ocfg, B, T)
    px, ids, mask = px.cuda(), ids.cuda(), mask.cuda()
    runs = {}
    try:
        for kv in (False, True):
            m.enable_fp8_decode(True, prefill=False, kv_cache=kv)
            seen = []

            def grab(ids_, scores):
                seen.append(scores.detach().float().clone())
                return scores
            force = None if not kv else runs[False][1]

            def teacher(ids_, scores):            # the fp8-cache run follows the bf16-cache run's tokens: same inputs at every step
                if force is None:
                    return scores
                step = len(seen) - 1
                out = torch.full_like(scores, -1e30)
                out.scatter_(1, force[:, step:step + 1], 0.0)
                return out
            toks = m.generate(input_ids=ids, pixel_values=px, attention_mask=mask, max_new_tokens=n_new, do_sample=False, eos_token_id=None,
                              logits_processor=LogitsProcessorList([grab, teacher]))
            runs[kv] = (seen, toks)
            if kv:      # the device-resident loop (what the benchmark times) runs the same kernels
                loop = m.generate(input_ids=ids, pixel_values=px, attention_mask=mask, max_new_tokens=2, do_sample=False, eos_token_id=None)
                assert loop.shape == (B, 2)
    finally:
        m.enable_fp8_decode(False)
    assert torch.equal(runs[True][1], runs[False][1])
    assert torch.equal(runs[True][0][0], runs[False][0][0])            # the prefill does not touch the cache: identical first logits
    worst_cos, worst_mean = 1.0, 0.0
    for s_ in range(1, n_new):
        a, b = runs[True][0][s_], runs[False][0][s_]
        std = b.std().item()
        err = (a - b).abs()
        cos = torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
        worst_cos, worst_mean = min(worst_cos, cos), max(worst_mean, err.mean().item() / std)
        _report(f"7B fp8 K/V cache vs bf16 cache [B={B}] decode step {s_}: logits max err {err.max().item():.3e} mean {err.mean().item():.3e} "
                f"(logit std {std:.3f}, cosine {cos:.4f})")
    assert worst_cos >= 0.97 and worst_mean <= 0.2


@pytest.mark.parametrize("B,rows", [(64, [0, 37, 63]), (256, [0, 100, 255])])
def test_7b_batch_rows_match_oracle(model_7b, B, rows):
    """BASELINE configs[2] (B = 64, T = 128) and the N = 1 leg of north_star's batch-256 scaling claim (B = 256 on ONE GPU) end to
    end against the ORACLE: the prefill's last-position logits and two teacher-forced decode steps of three rows of a B-row
    generate().  Rows are independent (no cross-sample reduction on the path), so the oracle runs on those rows only; the HIP side
    runs the full-batch instances the benchmark times -- B = 64: 256x256 prefill tiles, streaming decode GEMMs with MT = 4, the 2-wave
    batch decode attention; B = 256: the decode steps on the ring kernel (gemm_ring.hip: one launch per GEMM, whole K per tile, the weight pieces from
    the fragment-major twins) and the batch decode attention over 8192 (sequence, head) pairs."""
    from transformers import LogitsProcessorList
    m, ocfg = model_7b
    _oracle_threads()
    T, n_new = 128, 3
    px, ids, mask = O.make_inputs(ocfg, B, T)
    W = _w7(m)
    seen = []

    def grab(ids_, scores):
        seen.append(scores[rows].detach().float().cpu().clone())
        return scores
    toks = m.generate(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), max_new_tokens=n_new, do_sample=False,
                      eos_token_id=None, logits_processor=LogitsProcessorList([grab])).cpu()
    loop = m.generate(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), max_new_tokens=n_new, do_sample=False,
                      eos_token_id=None).cpu()          # the device-resident hipGraph loop the benchmark times: same kernels
    assert torch.equal(loop, toks)
    with torch.no_grad():
        img = O.image_embeds(px[rows], W, ocfg)
        x = O.embed_and_splice(ids[rows], img, W, ocfg)
        cache = [None] * ocfg.text.num_hidden_layers
        h = O.llama_forward(x, W, ocfg.text, mask[rows], cache, 0)
        refs = [O.lm_head(h[:, -1:], W)[:, 0]]
        past = T
        for s_ in range(n_new - 1):
            e = W["text_model.model.embed_tokens.weight"][toks[rows, s_]][:, None, :]
            h = O.llama_forward(e, W, ocfg.text, torch.ones(len(rows), past + 1, dtype=torch.int64), cache, past)
            refs.append(O.lm_head(h, W)[:, 0])
            past += 1
    for s_ in range(n_new):
        e_ = (seen[s_] - refs[s_]).abs()
        _report(f"7B B={B} rows {rows} {'prefill' if s_ == 0 else f'decode step {s_}'} logits vs fp32 oracle: max {e_.max().item():.3e} mean {e_.mean().item():.3e}")
        assert e_.max().item() <= B7_LOGIT_MAX and e_.mean().item() <= B7_LOGIT_MEAN, (s_, e_.max().item(), e_.mean().item())
        t2 = refs[s_].topk(2, dim=-1)
        decided = (t2.values[:, 0] - t2.values[:, 1]) > B7_MARGIN
        assert bool((seen[s_].argmax(-1) == t2.indices[:, 0])[decided].all())


def test_7b_336px_vision_stack_matches_oracle(model_7b):
    """BASELINE configs[4] patching at the 7B shape: 336 px -> N = 577 ViT tokens, 641 resampler keys (the ViT attention takes
    the 4-wave / multi-block form here, the GEMMs other tile counts).  All 24 ViT taps, post-LN, the 6 resampler taps and the
    projected image embeds of a B = 2 batch against the fp32 oracle with the bicubically grown position embedding
    (models/visualcla/modeling_visualcla.py:13-43)."""
    import copy
    from visualcla.weights import extend_position_embedding
    m, ocfg = model_7b
    _oracle_threads()
    W = _w7(m)
    pos224 = m._packed["vit.pos"]
    ocfg2 = copy.deepcopy(ocfg)
    ocfg2.vision.image_size = 336
    W2 = {k: v for k, v in W.items() if not k.startswith("text_model.")}
    pe_key = next(k for k in W2 if k.endswith("vision_model.embeddings.position_embedding.weight"))
    W2[pe_key] = W2[pe_key].clone()
    extend_position_embedding(W2, ocfg.vision.patch_size, 336)
    px, _, _ = O.make_inputs(ocfg2, 2, 128)
    m.set_image_size(336)
    try:
        taps = {}
        m.embed_images(px.cuda(), taps)
        torch.cuda.synchronize()
    finally:            # back to the 224-px model the other tests of this module share (same tensor, not a second interpolation)
        m.config.vision_config["image_size"] = 224
        m.vision_model.config.image_size = 224
        m._packed["vit.pos"] = pos224
        m._ws.clear()
        m._build_ctx()
    ref_t = {}
    with torch.no_grad():
        O.image_embeds(px, W2, ocfg2, ref_t)
    worst, n = 0.0, 0
    for k, ref in ref_t.items():
        if k not in taps:
            continue
        got = taps[k].float().cpu().reshape(ref.shape)
        err = (got - ref).abs()
        rel = err.max().item() / max(ref.abs().max().item(), 1e-6)
        worst, n = max(worst, rel), n + 1
        _report(f"7B@336px bf16 vs fp32 oracle {k}: max_abs_err={err.max().item():.3e} mean={err.mean().item():.3e} rel_to_absmax={rel:.3e}")
        assert rel <= B7_VIT_REL, (k, rel)
    assert n >= 24 + 1 + 6 + 1 and ref_t["vit_post_ln"].shape[1] == 577
    _report(f"7B@336px vision stack (N=577, KV=641): worst tap error {worst:.3e} of the tap's dynamic range (bound {B7_VIT_REL})")
