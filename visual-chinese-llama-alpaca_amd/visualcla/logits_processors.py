"""GenerationConfig -> the logits processors of one generate() call, and the generated-length rules.

The reference forwards its generation config and every keyword to HuggingFace's `generate`, driven by `inputs_embeds`
(models/visualcla/modeling_visualcla.py:382-391), so the config fields select transformers' own processor classes in transformers' own order
(`GenerationMixin._get_logits_processor`), every one of them seeing the NEW tokens only (the "decoder prompt" is empty: `input_ids_seq_length`
is 0).  This module restates that selection over the public classes of `transformers.generation.logits_process` -- the arithmetic of each
processor stays transformers' -- and the length rules of `_prepare_generated_length` for the `inputs_embeds` case.  Fields that would change the
result and have no implementation here are REFUSED by name (`refuse_unsupported`): nothing a caller sets is dropped silently.

The device-resident decode loop (csrc/sample.hip) implements repetition penalty, no-repeat-ngram, min-new-tokens, temperature, top-k and top-p;
`needs_host_processors` says when a config asks for more than that, which sends the request to the host-driven step path.
"""
from __future__ import annotations

import warnings
from typing import Callable, List, Optional, Sequence

import torch

# fields the device sampler does not implement: any of them set -> host-driven path with the processor classes below
_HOST_ONLY_FIELDS = ("sequence_bias", "bad_words_ids", "forced_bos_token_id", "forced_eos_token_id", "remove_invalid_values",
                     "exponential_decay_length_penalty", "suppress_tokens", "begin_suppress_tokens", "renormalize_logits",
                     "top_h", "min_p", "typical_p", "epsilon_cutoff", "eta_cutoff")

# (field, "is it switched on", what it is): generation features that change the output and are not implemented on this path
_REFUSED = (
    ("guidance_scale", lambda v: v is not None and v != 1, "classifier-free guidance"),
    ("watermarking_config", lambda v: v is not None, "watermarking"),
    ("penalty_alpha", lambda v: v is not None and v > 0, "contrastive search"),
    ("num_beam_groups", lambda v: v is not None and v > 1, "group beam search"),
    ("diversity_penalty", lambda v: v is not None and v != 0.0, "group beam search"),
    ("dola_layers", lambda v: v is not None, "DoLa decoding"),
    ("prompt_lookup_num_tokens", lambda v: v is not None, "prompt-lookup decoding"),
    ("constraints", lambda v: v is not None, "constrained beam search"),
    ("force_words_ids", lambda v: v is not None, "constrained beam search"),
    ("forced_decoder_ids", lambda v: v is not None, "forced decoder ids"),
    ("stop_strings", lambda v: v is not None, "stop strings (pass a StoppingCriteria built with the tokenizer instead)"),
    ("return_dict_in_generate", lambda v: v is True, "dictionary outputs (generate returns the new token ids)"),
    ("output_scores", lambda v: v is True, "score outputs"),
    ("output_logits", lambda v: v is True, "logit outputs"),
    ("output_attentions", lambda v: v is True, "attention outputs"),
    ("output_hidden_states", lambda v: v is True, "hidden-state outputs"),
)


def refuse_unsupported(gc, leftover_kwargs: Optional[dict] = None) -> None:
    """ValueError naming every generation feature the caller switched on that this path does not implement, and every keyword that is neither a
    GenerationConfig field nor an argument of generate() (HF's `_validate_model_kwargs` raises for those too)"""
    on = [f"{name} ({what})" for name, test, what in _REFUSED if test(getattr(gc, name, None))]
    if on:
        raise ValueError("generate(): not supported by the HIP path: " + ", ".join(on))
    if leftover_kwargs:
        raise ValueError(f"The following `model_kwargs` are not used by the model: {sorted(leftover_kwargs)} (note: typos in the generate arguments "
                         "will also show up in this list)")


def new_token_budget(gc, prompt_len: int) -> int:
    """how many tokens generate() may produce (hf generation/utils.py `_prepare_generated_length`, `inputs_embeds` case: the ids start empty):
    `max_new_tokens` if set; else an explicit `max_length` counts the prompt, so max_length - prompt_len (ValueError when nothing is left, HF's
    `_validate_generated_length`); else 20"""
    if gc.max_new_tokens is not None:
        return int(gc.max_new_tokens)
    from transformers import GenerationConfig
    library_default = GenerationConfig().max_length               # None since transformers 5; 20 before (there a config that still says 20 is "not set")
    if gc.max_length is not None and gc.max_length != library_default:
        n = int(gc.max_length) - int(prompt_len)
        if n <= 0:
            raise ValueError(f"Input length of input_ids is 0, but `max_length` is set to {n}. This can lead to unexpected behavior. You should consider "
                             "increasing `max_length` or, better yet, setting `max_new_tokens`.")
        return n
    return 20


def min_token_floor(gc, prompt_len: int) -> int:
    """new tokens before an eos id may be chosen: `min_new_tokens`, else `min_length` less the prompt (same function of HF); 0 = no floor"""
    if getattr(gc, "min_new_tokens", None) is not None:
        return max(int(gc.min_new_tokens), 0)
    if getattr(gc, "min_length", None) is not None:
        return max(int(gc.min_length) - int(prompt_len), 0)
    return 0


def needs_host_processors(gc) -> bool:
    """does the config switch on a processor the device sampler (csrc/sample.hip) does not implement?"""
    for f in _HOST_ONLY_FIELDS:
        v = getattr(gc, f, None)
        if v is None or v is False:
            continue
        if f == "typical_p" and v >= 1.0:
            continue
        if f in ("epsilon_cutoff", "eta_cutoff") and not 0.0 < v < 1.0:
            continue
        if f in ("top_h", "min_p", "typical_p", "epsilon_cutoff", "eta_cutoff") and not getattr(gc, "do_sample", None):
            continue                                                # warpers act only when sampling
        return True
    return False


def build_logits_processors(gc, eos: Sequence[int], device, prompt_len: int = 0, n_new: Optional[int] = None, extra: Optional[Sequence[Callable]] = None,
                            prefix_allowed_tokens_fn: Optional[Callable] = None) -> List[Callable]:
    """the processor list of one request, in transformers' order; `eos` = the stop ids ([] = none: the processors that act on them are left out,
    as upstream), `extra` = the caller's own processors (appended after the configured ones, before the sampling warpers), `n_new` = the token
    budget (what HF's adjusted `max_length` is when the ids start empty)"""
    from transformers.generation import logits_process as LP

    def on(name):
        return getattr(gc, name, None)

    def cls(name, field):
        c = getattr(LP, name, None)
        if c is None:
            raise ValueError(f"generate(): `{field}` needs transformers' {name}, which this transformers version does not have")
        return c
    eos_t = torch.tensor(list(eos), dtype=torch.int64, device=device) if len(eos) else None
    dev = str(device)
    nb = int(on("num_beams") or 1)
    procs: List[Callable] = []
    if on("sequence_bias") is not None:
        procs.append(LP.SequenceBiasLogitsProcessor(sequence_bias=gc.sequence_bias))
    for f in ("encoder_repetition_penalty", "encoder_no_repeat_ngram_size"):
        v = on(f)
        if v is not None and v not in (0, 1.0):
            # upstream: "requires some form of `input_ids` to be passed to `generate`, ignoring the argument" -- the reference passes inputs_embeds only
            warnings.warn(f"Passing `{f}` requires some form of `input_ids` to be passed to `generate`, ignoring the argument.", UserWarning)
    if on("repetition_penalty") is not None and gc.repetition_penalty != 1.0:
        procs.append(LP.RepetitionPenaltyLogitsProcessor(penalty=gc.repetition_penalty))
    if on("no_repeat_ngram_size") is not None and gc.no_repeat_ngram_size > 0:
        procs.append(LP.NoRepeatNGramLogitsProcessor(gc.no_repeat_ngram_size))
    if on("bad_words_ids") is not None:
        procs.append(LP.NoBadWordsLogitsProcessor(gc.bad_words_ids, eos_t))
    floor = min_token_floor(gc, prompt_len)
    if eos_t is not None and floor > 0:
        # upstream builds MinLength (from min_length, which min_new_tokens overwrites) and, with min_new_tokens, MinNewTokensLength too: the same mask twice
        procs.append(LP.MinLengthLogitsProcessor(floor, eos_t, device=dev))
        if on("min_new_tokens"):
            procs.append(LP.MinNewTokensLengthLogitsProcessor(0, int(gc.min_new_tokens), eos_t, device=dev))
    if prefix_allowed_tokens_fn is not None:
        procs.append(LP.PrefixConstrainedLogitsProcessor(prefix_allowed_tokens_fn, nb))
    if on("forced_bos_token_id") is not None:
        procs.append(LP.ForcedBOSTokenLogitsProcessor(gc.forced_bos_token_id))
    if on("forced_eos_token_id") is not None:
        budget = int(n_new) if n_new is not None else new_token_budget(gc, prompt_len)
        procs.append(LP.ForcedEOSTokenLogitsProcessor(budget, gc.forced_eos_token_id, device=dev))
    if on("remove_invalid_values") is True:
        procs.append(LP.InfNanRemoveLogitsProcessor())
    if on("exponential_decay_length_penalty") is not None:
        if eos_t is None:
            raise ValueError("generate(): `exponential_decay_length_penalty` raises the score of the eos ids: set `eos_token_id`")
        procs.append(LP.ExponentialDecayLengthPenalty(gc.exponential_decay_length_penalty, eos_t, 0))
    if on("suppress_tokens") is not None:
        procs.append(LP.SuppressTokensLogitsProcessor(gc.suppress_tokens, device=dev))
    if on("begin_suppress_tokens") is not None:
        # upstream: begin_index = input_ids_seq_length (0 here), one later when a forced bos token occupies the first position
        procs.append(LP.SuppressTokensAtBeginLogitsProcessor(gc.begin_suppress_tokens, 0 if on("forced_bos_token_id") is None else 1, device=dev))
    if extra:
        procs.extend(list(extra))
    if on("do_sample"):
        keep = (len(eos) + 1 if len(eos) else 2) if nb > 1 else 1
        if on("temperature") is not None and gc.temperature != 1.0:
            procs.append(LP.TemperatureLogitsWarper(gc.temperature))
        if on("top_h") is not None:
            procs.append(cls("TopHLogitsWarper", "top_h")(top_h=gc.top_h))
        if on("top_k") is not None and gc.top_k != 0:
            procs.append(LP.TopKLogitsWarper(top_k=gc.top_k, min_tokens_to_keep=keep))
        if on("top_p") is not None and gc.top_p < 1.0:
            procs.append(LP.TopPLogitsWarper(top_p=gc.top_p, min_tokens_to_keep=keep))
        if on("min_p") is not None:
            procs.append(cls("MinPLogitsWarper", "min_p")(min_p=gc.min_p, min_tokens_to_keep=keep))
        if on("typical_p") is not None and gc.typical_p < 1.0:
            procs.append(LP.TypicalLogitsWarper(mass=gc.typical_p, min_tokens_to_keep=keep))
        if on("epsilon_cutoff") is not None and 0.0 < gc.epsilon_cutoff < 1.0:
            procs.append(LP.EpsilonLogitsWarper(epsilon=gc.epsilon_cutoff, min_tokens_to_keep=keep))
        if on("eta_cutoff") is not None and 0.0 < gc.eta_cutoff < 1.0:
            procs.append(LP.EtaLogitsWarper(epsilon=gc.eta_cutoff, min_tokens_to_keep=keep, device=dev))
    if on("renormalize_logits") is True:
        procs.append(LP.LogitNormalization())
    return procs
