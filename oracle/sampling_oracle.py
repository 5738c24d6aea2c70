"""CPU oracle for next-row N2 (the step after the logits): a numpy restatement of the logits processors / warpers the
reference's DEFAULT_GENERATION_CONFIG (models/visualcla/modeling_utils.py:36-47) switches on inside HF `generate`, in HF's
order (third-party transformers, generation/logits_process.py + generation/utils.py `_get_logits_processor`):

    RepetitionPenaltyLogitsProcessor  score <0 ? score*p : score/p, gathered/scattered -> once per distinct token
    NoRepeatNGramLogitsProcessor      ban tokens that completed an earlier copy of the last n-1 generated tokens
    MinNewTokensLengthLogitsProcessor eos = -inf while fewer than min_new_tokens were generated
    TemperatureLogitsWarper           scores / temperature
    TopKLogitsWarper                  scores < k-th largest -> -inf   (ties at the k-th value survive)
    TopPLogitsWarper                  ascending cumulative softmax <= 1 - top_p -> -inf, keep >= min_tokens_to_keep
    softmax -> one draw

The history the processors see is the GENERATED tokens only: the reference drives generate with inputs_embeds
(modeling_visualcla.py:382-391).  HF draws with torch.multinomial (not reproducible across implementations); this oracle
and the HIP kernel draw by inverse CDF over the kept set in descending probability (ties: lower token id first) at a given
uniform, which has the same distribution and is a pure function of its inputs.

Ties: HF's top-p cut removes a prefix of `torch.sort(scores)`, and torch.sort is not stable, so WHICH of several equal
scores straddling the cut survive is implementation-defined in the reference stack; this oracle (and the kernel) use the
stable order -- lower token ids are dropped first.  The draw order breaks ties by ascending id (top_k = 1 == torch.argmax).

TEST INFRASTRUCTURE ONLY.  Pinned in tests/test_sampling_oracle.py against the HF classes themselves (transformers is
installed here and on the GPU box).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np


@dataclass
class SampleCfg:
    repetition_penalty: float = 1.0
    no_repeat_ngram_size: int = 0
    min_new_tokens: int = 0
    eos_ids: List[int] = field(default_factory=list)
    temperature: float = 1.0
    top_k: int = 1
    top_p: float = 1.0
    min_tokens_to_keep: int = 1


def process_scores(logits: np.ndarray, history: Sequence[int], cfg: SampleCfg) -> np.ndarray:
    """fp32 [V] logits of ONE sequence -> processed / warped scores (-inf = removed), fp32."""
    x = np.array(logits, dtype=np.float32, copy=True)
    V = x.shape[0]
    hist = [int(t) for t in history]
    h = len(hist)
    p = np.float32(cfg.repetition_penalty)
    if cfg.repetition_penalty != 1.0:
        for tok in sorted(set(hist)):
            x[tok] = x[tok] * p if x[tok] < 0 else x[tok] / p
    n = cfg.no_repeat_ngram_size
    if n > 0 and h + 1 >= n:
        prefix = hist[h + 1 - n:h]
        for i in range(0, h - n + 1):
            if hist[i:i + n - 1] == prefix:
                x[hist[i + n - 1]] = -np.inf
    if h < cfg.min_new_tokens:
        for e in cfg.eos_ids:
            x[e] = -np.inf
    x = x / np.float32(cfg.temperature)
    k = min(max(cfg.top_k, cfg.min_tokens_to_keep), V)
    kth = np.sort(x)[V - k]
    x = np.where(x < kth, np.float32(-np.inf), x)
    if cfg.top_p < 1.0:
        order = np.argsort(x, kind="stable")                    # ascending
        s = x[order]
        e = np.exp(s - s[-1], dtype=np.float32)
        probs = e / e.sum(dtype=np.float32)
        cum = np.cumsum(probs, dtype=np.float32)
        remove = cum <= np.float32(1 - cfg.top_p)
        remove[-cfg.min_tokens_to_keep:] = False
        x[order[remove]] = -np.inf
    return x


def kept_distribution(scores: np.ndarray):
    """-> (token ids, probabilities) of the surviving tokens, descending probability, ties by ascending id."""
    ids = np.nonzero(scores > -np.inf)[0]
    v = scores[ids]
    order = np.lexsort((ids, -v.astype(np.float64)))
    ids, v = ids[order], v[order]
    e = np.exp((v - v[0]).astype(np.float32))
    return ids, e / e.sum(dtype=np.float32)


def draw(scores: np.ndarray, u: float):
    """inverse CDF at u over the kept set -> (token id, the CDF, rank picked)"""
    ids, probs = kept_distribution(scores)
    cdf = np.cumsum(probs.astype(np.float64))
    r = int(np.searchsorted(cdf, u * cdf[-1], side="right"))
    r = min(r, len(ids) - 1)
    return int(ids[r]), cdf / cdf[-1], r


def sample_step(logits: np.ndarray, history: np.ndarray, cfg: SampleCfg, u: np.ndarray) -> np.ndarray:
    """logits [B, V], history [h, B] (step-major, as the device loop stores it), u [B] -> next tokens [B]"""
    B = logits.shape[0]
    return np.array([draw(process_scores(logits[b], history[:, b], cfg), float(u[b]))[0] for b in range(B)], dtype=np.int64)
