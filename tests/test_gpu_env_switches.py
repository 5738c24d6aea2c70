"""Every VCLA_* A/B switch the product library reads selects a kernel form that was the default at some point:
each must still produce the reference's numbers.  One child process per setting (the switches are read once per process): small
model, bf16, forward logits within the bf16 bound of the fp32 oracle and greedy ids equal wherever the oracle's margin allows."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SETTINGS = [
    {},                                     # defaults
    {"VCLA_DSTREAM": "0"},                  # split-K panel kernels instead of the streaming decode GEMMs
    {"VCLA_DS_DEFER": "0"},                 # a norm launch per RMSNorm instead of the deferred form
    {"VCLA_DS_SPLITK": "2"},
    {"VCLA_DS_GRID": "128"},
    {"VCLA_DS_QKV_SPLIT": "0"},             # qkv unsplit + plain decode attention instead of two raw K slices summed by the attention kernel
    {"VCLA_GEMV1X": "0"},                   # runtime-K decode GEMV
    {"VCLA_GEMV_OCC": "1"},
    {"VCLA_ATTN_FLASH": "0"},               # the phased decode attention of round 2
    {"VCLA_ATTN_FLASH": "0", "VCLA_ATTN_NW": "4", "VCLA_ATTN_COOP": "1"},
    {"VCLA_ATTN_MFMA_WHOLE": "0"},
    {"VCLA_ATTN_MFMA_NW": "4"},
    {"VCLA_GEMM_PERSIST": "1"},
    {"VCLA_GEMM_PF": "0"},                  # the 256 x 256 GEMM without the L2 prefetch (and without 257-row tiles)
    {"VCLA_GEMM_XR": "0"},                  # 256-row tiles + a tail launch for the ViT's M = B * 257
    {"VCLA_MFMA128_SPLITK": "0"},
    {"VCLA_TAIL_KERNEL": "7"},
    {"VCLA_TAIL_KERNEL": "8"},
    {"VCLA_MACRO_GRAPH": "0"},
    {"VCLA_DECODE_GRAPH": "0"},
    {"VCLA_GRAPH_STEPS": "2"},
]


@pytest.mark.parametrize("env", SETTINGS, ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()) or "defaults")
def test_env_switch_keeps_parity(env):
    child_env = {k: v for k, v in os.environ.items() if not k.startswith("VCLA_") or k == "VCLA_LIB"}
    child_env.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "env_switch_child.py")], env=child_env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    for tag, v in res.items():
        assert v["max"] <= 6e-2 and v["mean"] <= 1e-2 and v["ids_agree"], (env, tag, v)
