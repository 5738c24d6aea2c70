"""Beam search over the decode steps of VisualCLAModel.generate (num_beams > 1).

The reference forwards every generate() keyword to HuggingFace (models/visualcla/modeling_visualcla.py:382-391), so `num_beams`,
`length_penalty`, `early_stopping` and `num_return_sequences` select transformers' beam search over the LLaMA decoder, driven by
`inputs_embeds` (the prompt is not part of the returned ids: the "decoder prompt" is empty).  This module restates that procedure
(transformers 5.x `GenerationMixin._beam_search`: the vectorised form -- K = max(2, 1 + #eos) * num_beams candidates per step, live beams
and finished hypotheses kept in fixed-shape tensors, length penalty applied when a hypothesis finishes, the "can a live beam still beat the
worst finished one" heuristic) as HOST bookkeeping on small tensors; the arithmetic of the path stays where it is -- the caller supplies

    step(tokens [B * nb] int64, beam_rows [B * nb] int64) -> logits [B * nb, V] fp32

which first re-orders the K / V cache rows to `beam_rows` (row r of the new cache = row beam_rows[r] of the old one) and then runs one decode
step on `tokens` (libvisualcla_hip.so in the product; the CPU oracle in tests/).  Same tie-breaking as HF: every selection is a torch.topk
over the same float32 scores in the same layout.  Greedy beam search only (do_sample with beams draws without replacement from an
implementation-defined stream upstream; it is refused by the caller).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch

NEG = -1.0e9          # "cannot be chosen": the constant HF adds to scores it wants out of a top-k


def _take(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """x [B, n, ...], idx [B, k] -> [B, k, ...]: per batch row, rows idx of x"""
    ix = idx
    while ix.dim() < x.dim():
        ix = ix.unsqueeze(-1)
    return torch.take_along_dim(x, ix.expand(*idx.shape, *x.shape[2:]), dim=1)


@torch.no_grad()
def beam_search(first_logits: torch.Tensor, step: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], batch: int, num_beams: int,
                max_new_tokens: int, eos_ids: Sequence[int] = (), pad_token_id: Optional[int] = None, length_penalty: float = 1.0,
                early_stopping=False, num_return_sequences: int = 1, processors: Sequence[Callable] = (),
                stopping_criteria: Sequence[Callable] = ()) -> torch.Tensor:
    """first_logits: [batch * num_beams, V] -- the prefill's last-position logits of the EXPANDED batch (every prompt repeated num_beams times,
    rows b * num_beams .. + num_beams - 1 = prompt b).  Returns int64 [batch * num_return_sequences, n] new tokens, best hypothesis first."""
    dev = first_logits.device
    nb, V, L = int(num_beams), first_logits.shape[-1], int(max_new_tokens)
    if not 1 <= num_return_sequences <= nb:
        raise ValueError(f"num_return_sequences ({num_return_sequences}) has to be in [1, num_beams = {nb}]")
    eos = torch.tensor(list(eos_ids), dtype=torch.int64, device=dev) if len(eos_ids) else None
    keep = max(2, 1 + len(eos_ids)) * nb                       # candidates per step: enough that nb of them are NOT finished
    # HF: `pad_token_id or eos_token_id[0] if eos_token_id is not None else -1` (a pad id of 0 is falsy there: the eos id fills)
    fill = ((pad_token_id or int(eos_ids[0])) if len(eos_ids) else -1)
    top_mask = torch.zeros(keep, dtype=torch.bool, device=dev)
    top_mask[:nb] = True                                       # only the best nb candidates of a step may become finished hypotheses

    live_seq = torch.full((batch, nb, L), fill, dtype=torch.int64, device=dev)
    live_score = torch.zeros(batch, nb, dtype=torch.float32, device=dev)
    live_score[:, 1:] = NEG                                    # the nb copies of a prompt are identical: only beam 0 may branch at step 0
    live_rows = torch.full((batch, nb, L), -1, dtype=torch.int64, device=dev)      # cache row each live beam came from, per step
    done_seq = live_seq.clone()
    done_score = torch.full((batch, nb), NEG, dtype=torch.float32, device=dev)
    done_rows = live_rows.clone()
    done_flag = torch.zeros(batch, nb, dtype=torch.bool, device=dev)
    may_improve = torch.ones(batch, 1, dtype=torch.bool, device=dev)
    offs = (torch.arange(batch, device=dev) * nb)[:, None]

    logits = first_logits
    t = 0                                                      # tokens generated so far
    while True:
        flat_prefix = live_seq[:, :, :t].reshape(batch * nb, t)
        logp = torch.log_softmax(logits.to(torch.float32), dim=-1)
        for p in processors:
            logp = p(flat_prefix, logp)
        total = (logp.view(batch, nb, V) + live_score[:, :, None]).view(batch, nb * V)
        cand_score, cand = torch.topk(total, k=keep, dim=1)
        cand_beam, cand_tok = cand // V, cand % V
        cand_seq = _take(live_seq, cand_beam)
        cand_seq[:, :, t] = cand_tok
        cand_rows = _take(live_rows, cand_beam)
        cand_rows[:, :, t] = cand_beam + offs
        # which candidates end here: an eos token, the length limit, or a caller's stopping criterion
        hit = torch.zeros(batch, keep, dtype=torch.bool, device=dev)
        if eos is not None:
            hit |= torch.isin(cand_tok, eos)
        if t + 1 >= L:
            hit[:] = True
        for crit in stopping_criteria:
            r = crit(cand_seq[:, :, :t + 1].reshape(batch * keep, t + 1), None)
            r = r if isinstance(r, torch.Tensor) else torch.full((batch * keep,), bool(r), device=dev)
            hit |= r.to(dev).bool().view(batch, keep)
        # live beams of the next step: the best nb candidates that did NOT end
        masked = cand_score + hit.to(torch.float32) * NEG
        nxt = torch.topk(masked, k=nb, dim=1)[1]
        live_seq, live_score, live_rows = _take(cand_seq, nxt), _take(masked, nxt), _take(cand_rows, nxt)
        # finished hypotheses: candidates ranked inside the best nb that ended, scored with the length penalty, merged with the ones held so far
        just_done = hit & top_mask[None, :]
        fin = cand_score / float((t + 1) ** length_penalty)
        full = done_flag.all(dim=-1, keepdim=True) & (early_stopping is True)
        fin = fin + full.to(torch.float32) * NEG
        fin = fin + (~may_improve).to(torch.float32) * NEG
        fin = fin + (~just_done).to(torch.float32) * NEG
        m_seq, m_score = torch.cat([done_seq, cand_seq], dim=1), torch.cat([done_score, fin], dim=1)
        m_rows, m_flag = torch.cat([done_rows, cand_rows], dim=1), torch.cat([done_flag, just_done], dim=1)
        best = torch.topk(m_score, k=nb, dim=1)[1]
        done_seq, done_score, done_rows, done_flag = _take(m_seq, best), _take(m_score, best), _take(m_rows, best), _take(m_flag, best)
        beam_rows = live_rows[:, :, t].reshape(batch * nb)
        t += 1
        # can a live beam still beat the worst finished hypothesis?  (HF's heuristics: "never" with a positive penalty prices a live beam at the
        # full length, everything else at the current one)
        ref_len = L if (early_stopping == "never" and length_penalty > 0.0) else t
        best_live = live_score[:, :1] / float(ref_len ** length_penalty)
        worst_done = torch.where(done_flag, done_score.min(dim=1, keepdim=True)[0], torch.full_like(done_score, NEG))
        may_improve = may_improve & (best_live > worst_done).any(dim=-1, keepdim=True)
        go_on = bool(may_improve.any()) and not (bool(done_flag.all()) and early_stopping is True) and not bool(hit.all())
        if not go_on:
            break
        logits = step(live_seq[:, :, t - 1].reshape(batch * nb), beam_rows)
    out_seq = done_seq[:, :num_return_sequences].reshape(batch * num_return_sequences, L)
    out_rows = done_rows[:, :num_return_sequences].reshape(batch * num_return_sequences, L)
    n = int((out_rows >= 0).sum(dim=1).max())                # generated length of the longest returned hypothesis
    return out_seq[:, :n].contiguous()
