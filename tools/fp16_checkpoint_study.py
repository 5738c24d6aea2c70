#!/usr/bin/env python
"""What a real fp16 checkpoint loses when this package re-rounds it to bf16 (CPU, the oracle only; VERDICT r4 "missing" item 5).

The reference's GPU default is fp16 weights AND fp16 activations (models/visualcla/modeling_utils.py:88, :159); this package maps
`torch_dtype=float16` onto its bf16 product mode (visualcla/modeling_visualcla.py:_act_dtype): the checkpoint's fp16 values (11 significant
bits) are re-rounded to bf16 (8) at load and the activations are bf16.  Four forward passes of the oracle on the same inputs, all against
(0) = fp32 arithmetic on the fp16 checkpoint values (what the checkpoint "means"):
  (1) the reference's own GPU mode: fp16 weights, fp16 activations (oracle dtype=float16: what the HF modules compute under .half());
  (2) re-rounded WEIGHTS only: fp32 arithmetic on bf16(fp16 values)      -> the load-time loss in isolation;
  (3) this package's product mode: bf16(fp16 values), bf16 activations   (oracle dtype=bfloat16; the HIP path sits within a few 1e-3 of it,
      tests/test_gpu_model.py::test_bf16_path_is_no_worse_than_the_reference_run_in_bf16);
  (4) the same arithmetic on weights that were bf16 from the start (no fp16 detour): is the re-rounding visible next to bf16 activations at all?
usage: python tools/fp16_checkpoint_study.py [tiny|small]   (prints; profiles/r05_fp16_checkpoint_study.txt is its output)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import visualcla_oracle as O  # noqa: E402


def fp32_weights(cfg, seed=0):
    """the oracle's generator WITHOUT its bf16 rounding: fp32 values, as a training run leaves them"""
    orig = O._round_bf16
    O._round_bf16 = lambda t: t.float()
    try:
        return O.make_weights(cfg, seed=seed)
    finally:
        O._round_bf16 = orig


def study(name, out=sys.stdout):
    cfg = {"tiny": O.cfg_tiny, "small": O.cfg_small}[name]()
    W32 = fp32_weights(cfg)
    W16 = {k: v.to(torch.float16).float() for k, v in W32.items()}                 # the fp16 checkpoint
    Wb = {k: v.to(torch.bfloat16).float() for k, v in W16.items()}                 # what this package stores
    Wb_direct = {k: v.to(torch.bfloat16).float() for k, v in W32.items()}
    B, T = 2, 48 if name == "small" else 24
    px, ids, mask = O.make_inputs(cfg, B, T)
    with torch.no_grad():
        ref = O.visualcla_forward(ids, px, mask, W16, cfg)
        runs = [("(1) reference GPU mode: fp16 weights, fp16 activations", O.visualcla_forward(ids, px, mask, W16, cfg, dtype=torch.float16).float()),
                ("(2) weights re-rounded to bf16, fp32 arithmetic        ", O.visualcla_forward(ids, px, mask, Wb, cfg)),
                ("(3) this package: bf16(fp16 weights), bf16 activations ", O.visualcla_forward(ids, px, mask, Wb, cfg, dtype=torch.bfloat16).float()),
                ("(4) bf16 weights without the fp16 detour, bf16 activations", O.visualcla_forward(ids, px, mask, Wb_direct, cfg, dtype=torch.bfloat16).float())]
    std = ref.std().item()
    print(f"## {name}: logits [B={B}, T={T}, V={ref.shape[-1]}], std {std:.3f}; distance to fp32 arithmetic on the fp16 checkpoint values", file=out)
    res = {}
    for tag, got in runs:
        e = (got - ref).abs()
        agree = (got.argmax(-1) == ref.argmax(-1)).float().mean().item()
        res[tag[:3]] = (e.max().item(), e.mean().item())
        print(f"{tag}: max {e.max().item():.3e} mean {e.mean().item():.3e} ({e.mean().item() / std:.4f} sigma), argmax agreement {agree * 100:.1f} %", file=out)
    return res


if __name__ == "__main__":
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    for nm in (sys.argv[1:] or ["tiny", "small"]):
        study(nm)
