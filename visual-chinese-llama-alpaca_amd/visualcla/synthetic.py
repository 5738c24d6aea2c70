"""Synthetic request batches for benchmarking: N(0,1) pixels and
`BOS + prefix + <img> + Q x <img_token> + </img> + tail` prompts with random ordinary ids, identical length across the batch.
Product-side twin of the test oracle's generator (the benchmark must not lean on test infrastructure for its inputs);
tests/test_host_cpu.py checks the two produce the same tensors."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional, Tuple

import torch

# token ids of the 7B release: 49954 <img>, 49955 </img>, 49956 <pad>, 49957 <img_token> (added-token order of the
# reference's tokenizer, models/visualcla/modeling_utils.py:95; 49957 is pinned by the tgwebui pipeline)
IMG_START_7B, IMG_END_7B, IMG_TOKEN_7B = 49954, 49955, 49957


def stub_tokenizer(img_start: int = IMG_START_7B, img_end: int = IMG_END_7B, img_token: int = IMG_TOKEN_7B):
    """The three ids `VisualCLAModel` reads off `model.tokenizer` (no sentencepiece model ships with this repo)."""
    return SimpleNamespace(img_start_token_id=img_start, img_end_token_id=img_end, img_token_id=img_token, bos_token_id=1,
                           eos_token_id=2, pad_token_id=0)


def make_inputs(config, batch: int, seq_len: int, n_prefix: Optional[int] = None, image_size: Optional[int] = None,
                img_ids=(IMG_START_7B, IMG_END_7B, IMG_TOKEN_7B), seed_pixels: int = 1,
                seed_ids: int = 2, first_request: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """config: VisualCLAConfig.  -> (pixel_values [B,3,S,S] fp32 with bf16-representable values, input_ids [B,T], mask).
    `first_request`: return requests [first_request, first_request + batch) of the (unbounded) synthetic request stream -- a
    data-parallel rank builds only its own shard; the earlier requests are drawn one at a time and dropped, never held."""
    v = config.vision_config
    Q = config.visual_resampler_config["num_query_tokens"]
    S = image_size or v["image_size"]
    if n_prefix is None:
        n_prefix = max(0, min(23, seq_len - (Q + 3) - 1))
    n_tail = seq_len - (1 + n_prefix + 1 + Q + 1)
    if n_tail < 0:
        raise ValueError(f"seq_len {seq_len} too short for {Q} image tokens")
    g1 = torch.Generator().manual_seed(seed_pixels)
    g2 = torch.Generator().manual_seed(seed_ids)
    C = v.get("num_channels", 3)
    start, end, tok = img_ids
    hi = min(img_ids)   # ordinary ids stay below the special ones
    for _ in range(first_request):   # advance both streams past the other ranks' requests (C*S*S is a multiple of 16, the
        torch.randn(1, C, S, S, generator=g1)   # block size of torch's CPU normal sampler: per-request draws == one big draw)
        torch.randint(3, hi, (n_prefix,), generator=g2)
        torch.randint(3, hi, (n_tail,), generator=g2)
    px = torch.randn(batch, C, S, S, generator=g1).to(torch.bfloat16).to(torch.float32)
    rows = []
    for _ in range(batch):
        pre = torch.randint(3, hi, (n_prefix,), generator=g2)
        tail = torch.randint(3, hi, (n_tail,), generator=g2)
        rows.append(torch.cat([torch.tensor([1]), pre, torch.tensor([start]), torch.full((Q,), tok), torch.tensor([end]), tail]))
    ids = torch.stack(rows).to(torch.int64)
    return px, ids, torch.ones_like(ids)
