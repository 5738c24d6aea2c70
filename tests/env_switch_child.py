"""Child process of tests/test_gpu_env_switches.py: with the VCLA_* switches of the parent's choice in the environment, run the small
model's forward + generate (B = 3: the 2 <= M <= 64 batch-decode kernels; B = 1: the GEMV path) and print the logits error against
the oracle and the generated ids as one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402
from oracle import visualcla_oracle as O  # noqa: E402
from tests.helpers import make_hip_model  # noqa: E402

cfg = O.cfg_small()
W = O.make_weights(cfg, seed=0)
out = {}
m = make_hip_model(cfg, W, torch.bfloat16)
for B in (3, 1):
    px, ids, mask = O.make_inputs(cfg, B, 48)
    ref = O.visualcla_forward(ids, px, mask, W, cfg)
    got = m.forward(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda()).logits.float().cpu()
    want, ref_steps = O.visualcla_generate(ids, px, mask, W, cfg, max_new_tokens=5, return_logits=True)
    toks = None
    for _ in range(3):      # three calls: eager, captured, replayed (macro graphs) must agree
        t = m.generate(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda(), max_new_tokens=5, do_sample=False,
                       eos_token_id=None).cpu()
        assert toks is None or torch.equal(t, toks)
        toks = t
    alive = torch.ones(B, dtype=torch.bool)
    agree = True
    for s_ in range(5):
        t2 = ref_steps[s_].topk(2, dim=-1).values
        alive &= (t2[:, 0] - t2[:, 1]) > 0.12
        agree &= bool((toks[:, s_] == want[:, s_])[alive].all())
    out[f"B{B}"] = {"max": (got - ref).abs().max().item(), "mean": (got - ref).abs().mean().item(), "ids_agree": agree}
print("RESULT " + json.dumps(out))
