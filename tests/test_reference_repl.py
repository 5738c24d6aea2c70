"""The reference's own REPL as the drop-in test: `/root/reference/scripts/inference/inference.py` runs
UNMODIFIED as a child process with this repo's `visualcla` package on its PYTHONPATH, a synthetic checkpoint in the merged on-disk layout
and scripted stdin (two chat turns, `clear`, `change image:`, a missing file, `exit`).

There is no GPU in the container that holds /root/reference and no /root/reference on the GPU box, so the child gets the model's ARITHMETIC
from the CPU oracle (tests/repl_stub/oracle_backed.py) -- everything else it executes is the package's own host code: the loader called with
the REPL's exact keyword arguments (`--only_cpu`: torch.device('cpu'), device_map={'': cpu}, torch_dtype=float16), `.float()`, `.eval()`,
tokenizer / image-processor attachment, prompt assembly, history mutation and printing in `visualcla.chat`.  `peft` (not installed here,
imported by the REPL at module level) is a five-line stub.  Skipped where the reference is absent."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("VCLA_REFERENCE", "/root/reference")
SCRIPT = os.path.join(REF, "scripts", "inference", "inference.py")
STUB = os.path.join(ROOT, "tests", "repl_stub")
PKG = os.path.join(ROOT, "visual-chinese-llama-alpaca_amd")

pytestmark = pytest.mark.skipif(not os.path.isfile(SCRIPT), reason="the reference checkout is not on this machine")


def _expected(ckpt, turns, seed):
    """the same requests through the package's Python API in THIS process (same stand-in arithmetic, same seed)"""
    import transformers
    for p in (STUB, PKG, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_backed
    import visualcla
    saved = (visualcla.VisualCLAModel.from_state_dict, visualcla._lib.require_device)
    oracle_backed.install()
    try:
        transformers.set_seed(seed)
        model, tok, proc = visualcla.get_model_and_tokenizer_and_processor(visualcla_model=ckpt, torch_dtype=torch.float16,
                                                                           default_device=torch.device("cpu"), device_map={"": torch.device("cpu")})
        out = []
        for image, texts in turns:
            history = []
            for text in texts:
                response, history = visualcla.chat(model, image=image, text=text, history=history)
                out.append(response)
        return out
    finally:
        visualcla.VisualCLAModel.from_state_dict, visualcla._lib.require_device = saved


def test_reference_repl_runs_unmodified_against_this_package(tmp_path, capsys):
    from PIL import Image
    from oracle import visualcla_oracle as O
    from tests.test_gpu_dropin import _tiny_cfg, make_merged_dir
    cfg = _tiny_cfg()
    W = O.make_weights(cfg, seed=0)
    ckpt = make_merged_dir(str(tmp_path / "merged"), cfg, W)
    # keep the sampled responses short: the model's own generation config caps what DEFAULT_GENERATION_CONFIG (max_new_tokens=512) asks for?  No --
    # explicit fields win; a 2-layer 256-wide oracle does 512 tokens in seconds, so the REPL's defaults are left alone.
    rng = np.random.default_rng(0)
    img_a, img_b = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    Image.fromarray((rng.random((90, 120, 3)) * 255).astype(np.uint8)).save(img_a)
    Image.fromarray((rng.random((64, 64, 3)) * 255).astype(np.uint8)).save(img_b)
    stdin = "\n".join(["what is in the image?", "and what else?", "clear", "hello again", f"change image:{img_b}", "describe it",
                       f"change image:{tmp_path / 'missing.png'}", "anything there?", "exit"]) + "\n"
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([STUB, PKG]), VCLA_REPL_PATHS=os.pathsep.join([STUB, PKG, ROOT]), PYTHONUNBUFFERED="1")
    r = subprocess.run([sys.executable, SCRIPT, "--visualcla_model", ckpt, "--image_file", img_a, "--only_cpu", "--seed", "7"],
                       input=stdin, capture_output=True, text=True, env=env, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    out = r.stdout
    assert "Start Inference with instruction mode." in out and f"Image: {img_a}" in out
    assert out.count("Conversation history cleared.") == 1
    assert f"Cannot find file {tmp_path / 'missing.png'}. Clear history" in out
    assert "*** Exit Inference ***" in r.stderr + out
    got = re.findall(r"^(?:>)*Response: (.*)$", out, flags=re.M)
    hist = re.findall(r"^History: (.*)$", out, flags=re.M)
    assert len(got) == 4 and len(hist) == 4, out[-3000:]
    # turn 2 continues turn 1's history (4 entries, the image slot only in the first instruction); `clear` and `change image:` restart it
    n_entries = [h.count("'type'") for h in hist]
    assert n_entries == [2, 4, 2, 2], n_entries
    assert "first_instruction" in hist[0] and hist[1].count("first_instruction") == 1
    want = _expected(ckpt, [(img_a, ["what is in the image?", "and what else?"]), (img_a, ["hello again"]), (img_b, ["describe it"])], seed=7)
    capsys.readouterr()
    assert got == want, (got, want)
