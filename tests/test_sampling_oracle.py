"""Pin the N2 oracle (oracle/sampling_oracle.py) against the real HF logits processors it restates -- the classes
`model.generate` instantiates for the reference's DEFAULT_GENERATION_CONFIG (models/visualcla/modeling_utils.py:36-47)."""
import numpy as np
import pytest
import torch

from oracle import sampling_oracle as S


def hf_scores(logits, history, cfg):
    from transformers.generation import logits_process as LP
    procs = []
    if cfg.repetition_penalty != 1.0:
        procs.append(LP.RepetitionPenaltyLogitsProcessor(penalty=cfg.repetition_penalty))
    if cfg.no_repeat_ngram_size > 0:
        procs.append(LP.NoRepeatNGramLogitsProcessor(cfg.no_repeat_ngram_size))
    if cfg.min_new_tokens and cfg.eos_ids:
        procs.append(LP.MinNewTokensLengthLogitsProcessor(0, cfg.min_new_tokens, cfg.eos_ids, device="cpu"))
    if cfg.temperature != 1.0:
        procs.append(LP.TemperatureLogitsWarper(cfg.temperature))
    procs.append(LP.TopKLogitsWarper(top_k=cfg.top_k, min_tokens_to_keep=cfg.min_tokens_to_keep))
    if cfg.top_p < 1.0:
        procs.append(LP.TopPLogitsWarper(top_p=cfg.top_p, min_tokens_to_keep=cfg.min_tokens_to_keep))
    ids = torch.tensor(history, dtype=torch.long).reshape(1, -1)
    s = torch.from_numpy(logits.copy())[None]
    for p in procs:
        s = p(ids, s)
    return s[0].numpy()


CFGS = [
    S.SampleCfg(repetition_penalty=1.1, no_repeat_ngram_size=15, temperature=0.5, top_k=40, top_p=0.9),      # the reference default
    S.SampleCfg(repetition_penalty=1.3, no_repeat_ngram_size=2, temperature=1.7, top_k=7, top_p=0.5),
    S.SampleCfg(no_repeat_ngram_size=1, top_k=256, top_p=0.99),
    S.SampleCfg(repetition_penalty=2.0, top_k=1),                                                             # greedy + penalty
    S.SampleCfg(temperature=0.1, top_k=50, top_p=0.3, min_tokens_to_keep=3),
    S.SampleCfg(min_new_tokens=12, eos_ids=[2, 5], top_k=5, temperature=0.8),
    S.SampleCfg(no_repeat_ngram_size=3, top_k=20),
]


@pytest.mark.parametrize("ci", range(len(CFGS)))
@pytest.mark.parametrize("h", [0, 1, 2, 9, 40])
def test_processed_scores_match_hf(ci, h):
    cfg = CFGS[ci]
    rng = np.random.default_rng(100 * ci + h)
    V = 997
    logits = (rng.standard_normal(V) * 3).astype(np.float32)
    history = rng.integers(0, 6, size=h).tolist()          # tiny alphabet: repeats, repeated n-grams
    got = S.process_scores(logits, history, cfg)
    want = hf_scores(logits, history, cfg)
    assert np.array_equal(np.isinf(got), np.isinf(want))
    fin = ~np.isinf(want)
    np.testing.assert_allclose(got[fin], want[fin], rtol=0, atol=0)


def test_ties_at_the_kth_value_survive_like_hf():
    logits = np.zeros(50, np.float32)
    logits[[3, 7, 11, 30]] = [2.0, 1.0, 1.0, 1.0]
    cfg = S.SampleCfg(top_k=2)
    got = S.process_scores(logits, [], cfg)
    assert np.array_equal(np.isinf(got), np.isinf(hf_scores(logits, [], cfg)))
    ids, probs = S.kept_distribution(got)
    assert ids.tolist() == [3, 7, 11, 30] and abs(probs.sum() - 1) < 1e-6


def test_draw_is_the_inverse_cdf_and_has_the_right_distribution():
    rng = np.random.default_rng(0)
    logits = (rng.standard_normal(300) * 2).astype(np.float32)
    cfg = S.SampleCfg(temperature=0.7, top_k=8, top_p=0.95)
    sc = S.process_scores(logits, [], cfg)
    ids, probs = S.kept_distribution(sc)
    assert S.draw(sc, 0.0)[0] == ids[0] and S.draw(sc, 0.999999)[0] == ids[-1]
    us = rng.random(20000)
    counts = np.zeros(len(ids))
    for u in us:
        counts[S.draw(sc, u)[2]] += 1
    assert np.abs(counts / len(us) - probs).max() < 0.015
    # same distribution as the softmax HF hands to torch.multinomial
    want = torch.softmax(torch.from_numpy(hf_scores(logits, [], cfg)), -1).numpy()
    np.testing.assert_allclose(probs, want[ids], atol=1e-6)


def test_sample_step_batch_layout():
    rng = np.random.default_rng(1)
    B, V, h = 3, 64, 5
    logits = rng.standard_normal((B, V)).astype(np.float32)
    hist = rng.integers(0, V, size=(h, B))
    cfg = S.SampleCfg(repetition_penalty=1.2, top_k=4, top_p=0.9, temperature=0.9)
    u = rng.random(B).astype(np.float32)
    got = S.sample_step(logits, hist, cfg, u)
    for b in range(B):
        assert got[b] == S.draw(S.process_scores(logits[b], hist[:, b], cfg), float(u[b]))[0]


def test_top_p_cut_inside_a_tie_group():
    """equal scores straddling the top-p boundary: HF removes a prefix of torch.sort's ascending order, whose order inside a
    tie is implementation-defined (torch.sort is not stable).  What is defined -- how MANY survive and their scores -- must
    match HF; the oracle (and the kernel) fix the free choice as a stable sort would: lower token ids are dropped first."""
    logits = np.full(64, -30.0, np.float32)
    logits[[5, 9, 20, 33, 40, 41]] = [3.0, 1.0, 1.0, 1.0, 1.0, 2.0]
    kept = {}
    for top_p in (0.75, 0.8, 0.85, 0.9, 0.95):
        cfg = S.SampleCfg(top_k=50, top_p=top_p)
        got = S.process_scores(logits, [], cfg)
        want = hf_scores(logits, [], cfg)
        assert np.array_equal(np.sort(got), np.sort(want)), top_p
        kept[top_p] = np.nonzero(~np.isinf(got))[0].tolist()
    assert kept[0.75] == [5, 40, 41] and kept[0.8] == [5, 33, 40, 41] and kept[0.9] == [5, 20, 33, 40, 41]
