"""Next row N2: the on-device sampler (vcla_sample, and the decode loop that embeds it) against the HF-pinned oracle.
Kept sets must be identical, kept probabilities within 1e-5, and the draw must be the oracle's inverse-CDF pick (a pick one
rank off is accepted only when the uniform sits within 1e-5 of that CDF boundary: fp32 summation order)."""
import numpy as np
import pytest
import torch

from oracle import sampling_oracle as S
from oracle import visualcla_oracle as O
from tests.test_sampling_oracle import CFGS

pytestmark = pytest.mark.gpu


def _run(logits, hist, cfg, u):
    from visualcla import _lib
    B, V = logits.shape
    dev = "cuda:0"
    lg = torch.from_numpy(logits).to(dev)
    h = torch.from_numpy(hist).to(dev) if hist.shape[0] else torch.zeros(1, B, dtype=torch.int64, device=dev)
    ut = torch.zeros(hist.shape[0] + 1, B, device=dev)
    ut[hist.shape[0]] = torch.from_numpy(u).to(dev)
    kept_ids = torch.full((B, _lib.SAMPLE_KEPT_LD), -1, dtype=torch.int64, device=dev)
    kept_p = torch.zeros(B, _lib.SAMPLE_KEPT_LD, device=dev)
    n_kept = torch.zeros(B, dtype=torch.int32, device=dev)
    args = _lib.sample_args(cfg.repetition_penalty, cfg.no_repeat_ngram_size, cfg.min_new_tokens, cfg.eos_ids, cfg.temperature,
                            cfg.top_k, cfg.top_p, cfg.min_tokens_to_keep, uniforms=ut, history=h, kept_ids=kept_ids,
                            kept_probs=kept_p, n_kept=n_kept)
    out = _lib.sample(lg, args, n_hist=hist.shape[0])
    torch.cuda.synchronize()
    return out.cpu().numpy(), kept_ids.cpu().numpy(), kept_p.cpu().numpy(), n_kept.cpu().numpy()


def _check(logits, hist, cfg, u):
    out, kept_ids, kept_p, n_kept = _run(logits, hist, cfg, u)
    for b in range(logits.shape[0]):
        sc = S.process_scores(logits[b], hist[:, b], cfg)
        ids, probs = S.kept_distribution(sc)
        n = int(n_kept[b])
        assert n == len(ids), (b, n, len(ids))
        assert kept_ids[b, :n].tolist() == ids.tolist()
        np.testing.assert_allclose(kept_p[b, :n], probs, atol=1e-5)
        want, cdf, r = S.draw(sc, float(u[b]))
        if out[b] != want:
            got_r = ids.tolist().index(int(out[b]))
            edge = cdf[min(r, got_r)]
            assert abs(got_r - r) == 1 and abs(float(u[b]) - edge) < 1e-5, (b, out[b], want, u[b], edge)


@pytest.mark.parametrize("ci", range(len(CFGS)))
@pytest.mark.parametrize("h", [0, 1, 2, 9, 40, 200])
@pytest.mark.parametrize("V", [997, 49958])
def test_sampler_matches_oracle(ci, h, V):
    cfg = CFGS[ci]
    rng = np.random.default_rng(1000 * ci + h + V)
    B = 5
    logits = (rng.standard_normal((B, V)) * 3).astype(np.float32)
    hist = rng.integers(0, 6, size=(h, B)).astype(np.int64)
    u = rng.random(B).astype(np.float32)
    _check(logits, hist, cfg, u)


def test_reference_default_config_on_wide_history():
    """DEFAULT_GENERATION_CONFIG (modeling_utils.py:36-47) with a long history over the real vocabulary and a repeated
    15-gram planted so the n-gram ban fires"""
    cfg = CFGS[0]
    rng = np.random.default_rng(7)
    B, V, h = 3, 49958, 300
    logits = (rng.standard_normal((B, V)) * 4).astype(np.float32)
    hist = rng.integers(0, V, size=(h, B)).astype(np.int64)
    hist[h - 14:, :] = hist[20:34, :]                 # last 14 tokens = an earlier 14-gram -> hist[34] must be banned
    for b in range(B):
        logits[b, hist[34, b]] = 50.0                 # and it would otherwise win by a mile
    u = rng.random(B).astype(np.float32)
    _check(logits, hist, cfg, u)
    out, *_ = _run(logits, hist, cfg, u)
    assert all(out[b] != hist[34, b] for b in range(B))


def test_ties_extremes_and_greedy():
    from visualcla import _lib
    V = 4096
    logits = np.zeros((2, V), np.float32)
    logits[0, [3, 7, 11, 30]] = [2.0, 1.0, 1.0, 1.0]          # ties at the k-th value all survive
    logits[1, :] = -np.inf
    logits[1, 77] = -3.0                                      # a single finite logit
    cfg = S.SampleCfg(top_k=2)
    out, kept_ids, kept_p, n_kept = _run(logits, np.zeros((0, 2), np.int64), cfg, np.array([0.99, 0.5], np.float32))
    assert n_kept.tolist() == [4, 1] and kept_ids[0, :4].tolist() == [3, 7, 11, 30] and out[1] == 77
    # top_k = 1, no uniforms: argmax of the processed scores, first index on ties
    rng = np.random.default_rng(3)
    lg = rng.standard_normal((4, V)).astype(np.float32)
    lg[2, 100] = lg[2, 50] = 9.0
    t = torch.from_numpy(lg).cuda()
    got = _lib.sample(t.clone(), _lib.sample_args(top_k=1), n_hist=0).cpu()
    assert torch.equal(got, torch.from_numpy(lg).argmax(-1)) and got[2] == 50


def test_bad_arguments_raise():
    from visualcla import _lib
    t = torch.zeros(1, 100, device="cuda")
    for kw in (dict(top_k=0), dict(top_k=257), dict(temperature=0.0), dict(top_p=0.0), dict(top_p=1.5), dict(repetition_penalty=0.0)):
        with pytest.raises(ValueError):
            _lib.sample(t, _lib.sample_args(**kw))


# ---------------------------------------------------------------- through generate()
def _model_and_inputs(dtype=torch.float32):
    from tests.helpers import make_hip_model
    cfg = O.cfg_tiny()
    W = O.make_weights(cfg, seed=0)
    model = make_hip_model(cfg, W, dtype)
    px, ids, mask = O.make_inputs(cfg, 2, 24)
    return cfg, W, model, px, ids, mask


@pytest.mark.parametrize("use_graph", [False, True])
def test_generate_sampling_matches_oracle_with_the_same_uniforms(use_graph):
    from transformers import GenerationConfig
    cfg, W, model, px, ids, mask = _model_and_inputs()
    n_new = 12
    gc = GenerationConfig(max_new_tokens=n_new, do_sample=True, top_p=0.9, top_k=40, temperature=0.5, repetition_penalty=1.1,
                          no_repeat_ngram_size=3, eos_token_id=None)
    scfg = S.SampleCfg(repetition_penalty=1.1, no_repeat_ngram_size=3, temperature=0.5, top_k=40, top_p=0.9)
    torch.manual_seed(1234)
    got = model.generate(input_ids=ids, pixel_values=px, attention_mask=mask, generation_config=gc, use_graph=use_graph,
                         device_sampling=True).cpu()
    torch.manual_seed(1234)
    u = torch.rand(n_new, 2, device="cuda:0").cpu().numpy()

    def select(logits, generated):
        return torch.from_numpy(S.sample_step(logits.numpy(), generated.numpy().T.copy(), scfg, u[generated.shape[1]]))

    want = O.visualcla_generate(ids, px, mask, W, cfg, max_new_tokens=n_new, select_fn=select)
    assert torch.equal(got, want)


def test_generate_greedy_with_penalties_runs_on_device_and_matches_host_path():
    """do_sample=False + repetition penalty / n-gram ban: the device loop (top_k = 1) and the host-driven HF-processor path
    must pick the same tokens"""
    from transformers import GenerationConfig
    cfg, W, model, px, ids, mask = _model_and_inputs()
    gc = GenerationConfig(max_new_tokens=16, do_sample=False, repetition_penalty=1.3, no_repeat_ngram_size=2, eos_token_id=None)
    dev = model.generate(input_ids=ids, pixel_values=px, attention_mask=mask, generation_config=gc, device_sampling=True)
    host = model.generate(input_ids=ids, pixel_values=px, attention_mask=mask, generation_config=gc, device_sampling=False)
    assert torch.equal(dev, host)
    toks = dev.cpu().tolist()
    for row in toks:                                   # no bigram repeats, by construction of the ban
        bigrams = list(zip(row, row[1:]))
        assert len(bigrams) == len(set(bigrams))


def test_generate_sampling_eos_min_new_tokens_and_fallbacks():
    from transformers import GenerationConfig
    cfg, W, model, px, ids, mask = _model_and_inputs()
    # find what greedy would emit second, make it the eos: min_new_tokens must keep it away until step 5
    g = model.generate(input_ids=ids[:1], pixel_values=px[:1], attention_mask=mask[:1],
                       generation_config=GenerationConfig(max_new_tokens=3, do_sample=False, eos_token_id=None))
    eos = int(g[0, 1])
    gc = GenerationConfig(max_new_tokens=10, min_new_tokens=5, do_sample=True, top_k=1, eos_token_id=eos, pad_token_id=0)
    out = model.generate(input_ids=ids[:1], pixel_values=px[:1], attention_mask=mask[:1], generation_config=gc, device_sampling=True)
    row = out[0].tolist()
    assert eos not in row[:5]
    # top_k = 0 cannot run on the device: explicit request fails loudly, default silently takes the host path
    gc0 = GenerationConfig(max_new_tokens=4, do_sample=True, top_k=0, top_p=0.9, eos_token_id=None)
    with pytest.raises(ValueError):
        model.generate(input_ids=ids, pixel_values=px, attention_mask=mask, generation_config=gc0, device_sampling=True)
    assert model.generate(input_ids=ids, pixel_values=px, attention_mask=mask, generation_config=gc0).shape == (2, 4)


def test_sampling_is_reproducible_and_seed_dependent_in_bf16():
    from transformers import GenerationConfig
    cfg, W, model, px, ids, mask = _model_and_inputs(torch.bfloat16)
    gc = GenerationConfig(max_new_tokens=24, do_sample=True, top_k=50, top_p=0.95, temperature=1.5, eos_token_id=None)
    outs = []
    for seed in (1, 1, 2):
        torch.manual_seed(seed)
        outs.append(model.generate(input_ids=ids, pixel_values=px, attention_mask=mask, generation_config=gc).cpu())
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])


def test_top_p_cut_inside_a_tie_group_matches_the_oracle_convention():
    logits = np.full((1, 64), -30.0, np.float32)
    logits[0, [5, 9, 20, 33, 40, 41]] = [3.0, 1.0, 1.0, 1.0, 1.0, 2.0]
    for top_p in (0.75, 0.8, 0.85, 0.9, 0.95):
        _check(logits, np.zeros((0, 1), np.int64), S.SampleCfg(top_k=50, top_p=top_p), np.array([0.97], np.float32))


def test_streaming_path_uses_the_device_sampler_and_matches_the_graph_loop():
    """stopping criteria (what chat_in_stream installs) force the host-driven loop; token selection there is the same
    kernel, so for the same uniforms the two loops must emit the same tokens -- and HF's own processors must agree on greedy"""
    from transformers import GenerationConfig, StoppingCriteria, StoppingCriteriaList
    cfg, W, model, px, ids, mask = _model_and_inputs()
    seen = []

    class Spy(StoppingCriteria):
        def __call__(self, input_ids, scores, **kw):
            seen.append(input_ids.shape)
            return False

    gc = GenerationConfig(max_new_tokens=9, do_sample=True, top_p=0.9, top_k=40, temperature=0.5, repetition_penalty=1.1,
                          no_repeat_ngram_size=3, eos_token_id=None)
    torch.manual_seed(77)
    loop = model.generate(input_ids=ids, pixel_values=px, attention_mask=mask, generation_config=gc)
    torch.manual_seed(77)
    stepped = model.generate(input_ids=ids, pixel_values=px, attention_mask=mask, generation_config=gc,
                             stopping_criteria=StoppingCriteriaList([Spy()]))
    assert torch.equal(loop, stepped) and seen == [(2, i + 1) for i in range(9)]
    g2 = GenerationConfig(max_new_tokens=9, do_sample=False, repetition_penalty=1.2, no_repeat_ngram_size=2, eos_token_id=None)
    dev = model.generate(input_ids=ids, pixel_values=px, attention_mask=mask, generation_config=g2, stopping_criteria=StoppingCriteriaList([Spy()]))
    host = model.generate(input_ids=ids, pixel_values=px, attention_mask=mask, generation_config=g2, stopping_criteria=StoppingCriteriaList([Spy()]),
                          device_sampling=False)
    assert torch.equal(dev, host)


def test_tie_run_beyond_the_kept_capacity_is_deterministic():
    """More equal scores at the k-th value than the kept-set capacity (HF would keep them all): the kernel keeps every token
    strictly above the tie and fills the rest of its 512 slots with the LOWEST tied token ids -- the same set on every run,
    not whichever lanes won an atomic race"""
    import dataclasses
    cfg = dataclasses.replace(CFGS[0], repetition_penalty=1.0, no_repeat_ngram_size=0, top_k=40, top_p=1.0, temperature=1.0)
    rng = np.random.default_rng(11)
    B, V = 2, 49958
    logits = (rng.standard_normal((B, V)) * 0.5 - 10.0).astype(np.float32)
    above = rng.choice(V, size=7, replace=False)
    tied = np.setdiff1d(rng.choice(V, size=900, replace=False), above)
    logits[:, tied] = 3.0
    logits[:, above] = 5.0 + np.arange(7, dtype=np.float32)
    u = np.array([0.3, 0.9], dtype=np.float32)
    hist = np.zeros((0, B), dtype=np.int64)
    runs = [_run(logits.copy(), hist, cfg, u) for _ in range(3)]
    out0, ids0, p0, n0 = runs[0]
    cap = ids0.shape[1]
    want = set(above.tolist()) | set(np.sort(tied)[: cap - len(above)].tolist())
    for b in range(B):
        assert int(n0[b]) == cap and set(ids0[b].tolist()) == want
        assert abs(float(p0[b].sum()) - 1.0) < 1e-4 and int(out0[b]) in want
    for out, ids, p, n in runs[1:]:
        assert np.array_equal(out, out0) and np.array_equal(ids, ids0) and np.array_equal(n, n0)
