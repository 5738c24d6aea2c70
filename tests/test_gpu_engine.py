"""The persistent B = 1 decode step (csrc/decode_engine.hip) against the launch path it replaces (engine.hip's five launches per layer), on a reduced-depth
decoder at the LLaMA-7B widths (the engine's geometry is fixed; depth and vocabulary are free).  The full-depth engine is what every B = 1 bf16 test of
tests/test_gpu_model.py runs (oracle comparisons at 7B: test_7b_prefill_and_decode_logits_match_oracle, test_7b_decode_equals_forward_...); here the two
forms are compared with EACH OTHER step by step, stage by stage where a stage is observable (K/V cache rows, logits, tokens)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

LAYERS, VOCAB = 3, 5003            # 5003: a ragged last lm_head slot on most CUs, rows past the vocabulary on the last ones


@pytest.fixture(scope="module")
def small7b():
    import visualcla
    cfg = visualcla.visualcla_7b_config()
    cfg.text_config.update(num_hidden_layers=LAYERS, vocab_size=VOCAB)
    cfg.vision_config.update(num_hidden_layers=1, hidden_size=256, intermediate_size=512, num_attention_heads=4)
    cfg.visual_resampler_config.update(num_hidden_layers=1, hidden_size=256, intermediate_size=512, num_attention_heads=4)
    m = visualcla.VisualCLAModel.from_random(cfg, device="cuda:0", torch_dtype=torch.bfloat16, seed=3)
    assert "llama.engine.w" in m._packed and "llama.engine.g" in m._packed
    yield m
    os.environ.pop("VCLA_ENGINE", None)
    del m
    torch.cuda.empty_cache()


def _steps(m, mode, T, n_steps, masked, forced=None):
    """prefill a T-token prompt, then n_steps host-driven decode steps (vcla_llama_decode_step) with VCLA_ENGINE = mode; returns tokens, logits, cache"""
    from visualcla import _lib
    lib = _lib.load()
    os.environ["VCLA_ENGINE"] = mode
    dev = m.device
    V = m.config.text_config["vocab_size"]
    ids = torch.randint(3, V - 8, (1, T), generator=torch.Generator().manual_seed(5)).to(dev)
    ctx_max = (T + n_steps + 2 + 63) // 64 * 64
    embeds, _ = m._embed(ids, None, None)
    cache = m._new_cache(1, ctx_max)
    cache.kv.zero_()
    am = None
    if masked:                     # holes in the prompt's mask: the MASK instantiation must skip exactly those keys
        am = torch.ones(1, T, dtype=torch.int64, device=dev)
        am[0, 3:9] = 0
        am[0, T // 2] = 0
    key_mask = m._key_mask(am, 1, T, ctx_max)
    logits = m._prefill(embeds, cache, key_mask, all_logits=False)
    ws = m._buf("llama", lib.vcla_llama_workspace_bytes(m._ctx, 1, 1))
    step_logits = torch.empty(1, V, dtype=torch.float32, device=dev)
    tok = logits.argmax(-1)
    toks, lgs = [int(tok)], []
    for s in range(n_steps):
        if forced is not None:
            tok = torch.tensor([forced[s]], device=dev)
        _lib.check(lib.vcla_llama_decode_step(m._ctx, tok.contiguous().data_ptr(), 1, T + s, None, 0, cache.kv.data_ptr(), ctx_max, _lib.ptr(key_mask),
                                              step_logits.data_ptr(), None, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        _lib.check(lib.vcla_llama_decode_status(m._ctx, 1, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        lgs.append(step_logits.clone())
        tok = step_logits.argmax(-1)
        toks.append(int(tok))
    torch.cuda.synchronize()
    return toks, lgs, cache.kv[..., :T + n_steps, :].float().clone()


# 510 / 509: the four steps cross args.split_min = 512 inside one loop (one CU per head -> the head's 8 CUs share the cache walk); 1011 / 700 / 1900: split throughout
@pytest.mark.parametrize("T,masked", [(160, False), (37, False), (200, True), (1, False), (1011, False), (700, True), (510, False), (509, True), (1900, False)])
def test_engine_steps_match_the_launch_path(small7b, T, masked):
    """logits of every decode step (teacher-forced on the launch path's tokens) and the K / V rows both forms append: within bf16 rounding of each other.
    Bounds: the two forms round the RMSNorm output differently (the engine to bf16 once, the launches keep fp32) and sum in different orders; measured at
    3 layers: logits max 0.08 / mean 0.017 at std 1.29 (a 32-layer model: 0.27 / 0.05, the same as either form's distance to the fp32 oracle)."""
    n = 4
    t0, l0, kv0 = _steps(small7b, "0", T, n, masked)
    t1, l1, kv1 = _steps(small7b, "1", T, n, masked, forced=t0)
    for s in range(n):
        d = (l0[s] - l1[s]).abs()
        assert torch.isfinite(l1[s]).all()
        assert d.max().item() <= 0.2 and d.mean().item() <= 0.04, (T, s, d.max().item(), d.mean().item())
        top2 = l0[s].topk(2, dim=-1).values[0]
        if float(top2[0] - top2[1]) > 0.3:
            assert int(l0[s].argmax()) == int(l1[s].argmax()), (T, s)
    # the cache rows of the decode steps (and only those) were written by the step kernels
    assert (kv0[..., T:, :] - kv1[..., T:, :]).abs().max().item() <= 0.15
    assert (kv1[..., T:, :].abs().amax(dim=(-1,)) > 0).all()               # every (layer, k/v, head) row of every step was appended


def test_engine_graph_loop_equals_host_driven_steps(small7b):
    """the device-resident loop (hipGraph replay of ONE captured engine step, launch sequence numbers advancing in device memory) must produce the tokens of
    host-driven single steps (workspace zeroed per call) -- the engine is deterministic, so exactly"""
    m = small7b
    os.environ["VCLA_ENGINE"] = "1"
    dev = m.device
    V = m.config.text_config["vocab_size"]
    ids = torch.randint(3, V - 8, (1, 90), generator=torch.Generator().manual_seed(9)).to(dev)
    kw = dict(input_ids=ids, max_new_tokens=24, do_sample=False, eos_token_id=None)
    a = m.generate(use_graph=True, **kw)
    b = m.generate(use_graph=False, **kw)
    from transformers import LogitsProcessorList
    c = m.generate(logits_processor=LogitsProcessorList([lambda i, s: s]), **kw)       # host-driven path (one vcla_llama_decode_step per token)
    assert torch.equal(a, b) and torch.equal(a, c), (a.tolist(), b.tolist(), c.tolist())
    a2 = m.generate(use_graph=True, **kw)                                              # the cached graph, a fresh loop: sequence numbers restart
    assert torch.equal(a, a2)


def test_engine_is_not_used_where_it_does_not_apply(small7b):
    """fp8 decode copies, batch sizes other than 1 and VCLA_ENGINE=0 keep the launch path; the status call is a no-op there"""
    from visualcla import _lib
    m = small7b
    os.environ["VCLA_ENGINE"] = "1"
    V = m.config.text_config["vocab_size"]
    ids = torch.randint(3, V - 8, (2, 40), generator=torch.Generator().manual_seed(1)).to(m.device)
    kw = dict(max_new_tokens=5, do_sample=False, eos_token_id=None)
    two = m.generate(input_ids=ids, **kw)
    assert two.shape == (2, 5)
    m.enable_fp8_decode(True, prefill=False)
    try:
        one8 = m.generate(input_ids=ids[:1], **kw)
    finally:
        m.enable_fp8_decode(False)
    assert one8.shape == (1, 5)
    lib = _lib.load()
    ws = m._buf("llama", lib.vcla_llama_workspace_bytes(m._ctx, 2, 1))
    _lib.check(lib.vcla_llama_decode_status(m._ctx, 2, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))


def test_engine_steps_match_the_oracle():
    """the engine against the fp32 CPU oracle directly (two layers at the LLaMA-7B widths: the oracle runs in seconds): every decode step's logits within
    0.12 max / 0.022 mean of the oracle (measured: 0.056 - 0.081 / 0.0129 - 0.0163; the launch path on the same model and steps: 0.055 - 0.080 / 0.0122 -
    0.0155), argmax equal wherever the oracle's top-2 margin exceeds twice the max bound, and the engine's mean error within 1.25 x the launch path's --
    the two forms are interchangeable"""
    from oracle import visualcla_oracle as O
    from tests.helpers import cfg_engine_small, engine_steps_vs_oracle, make_hip_model
    cfg = cfg_engine_small()
    W = O.make_weights(cfg, seed=1)
    m = make_hip_model(cfg, W, torch.bfloat16)
    assert "llama.engine.w" in m._packed
    got = {}
    try:
        for mode in ("1", "0"):
            os.environ["VCLA_ENGINE"] = mode
            steps = engine_steps_vs_oracle(m, cfg, W, T=33, n_steps=4)
            for s, (mx, mean, a_hip, a_ref, margin) in enumerate(steps):
                assert mx < 0.12 and mean < 0.022, (mode, s, mx, mean)
                assert margin < 0.24 or a_hip == a_ref, (mode, s, a_hip, a_ref, margin)
            got[mode] = steps
            print(f"VCLA_ENGINE={mode}: decode-step logits vs fp32 oracle max {max(x[0] for x in steps[1:]):.3e} mean {max(x[1] for x in steps[1:]):.3e}")
    finally:
        os.environ.pop("VCLA_ENGINE", None)
    for s in range(1, 5):
        assert got["1"][s][1] <= 1.25 * got["0"][s][1] + 1e-3, (s, got["1"][s], got["0"][s])


def test_a_workgroup_that_never_publishes_is_a_status_code_not_a_hang(small7b):
    """every wait inside the persistent launch is bounded: with one workgroup's consumers gone (test hook VCLA_ENGINE_FAULT=1) the launch must drain within
    about a second, `vcla_llama_decode_status` must name a wait site, `generate()` must raise instead of returning tokens -- and the next call must work"""
    import time
    from visualcla import _lib
    m = small7b
    os.environ["VCLA_ENGINE"] = "1"
    V = m.config.text_config["vocab_size"]
    ids = torch.randint(3, V - 8, (1, 33), generator=torch.Generator().manual_seed(2)).to(m.device)
    kw = dict(input_ids=ids, max_new_tokens=3, do_sample=False, eos_token_id=None, use_graph=False)
    good = m.generate(**kw)
    os.environ["VCLA_ENGINE_FAULT"] = "1"
    try:
        t0 = time.perf_counter()
        with pytest.raises(_lib.VclaError, match="timed out"):
            m.generate(**kw)
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 20.0
    finally:
        os.environ.pop("VCLA_ENGINE_FAULT", None)
    assert torch.equal(m.generate(**kw), good)


def test_engine_soak_many_calls_of_changing_shape(small7b):
    """60 generate() calls in one process with prompt lengths, new-token counts, masks and graph / eager loops changing from call to call: every call must end
    with a clean decode status (no wait ran out: `generate()` checks it), return finite in-vocabulary tokens of the right shape, and repeat exactly when the same
    call is made again later (launch sequence numbers, mailbox parities and cached step graphs carry nothing from one call into the next)."""
    m = small7b
    os.environ["VCLA_ENGINE"] = "1"
    V = m.config.text_config["vocab_size"]
    g = torch.Generator().manual_seed(23)
    r = lambda n: int(torch.randint(0, n, (1,), generator=g))
    first = {}
    calls = []
    for i in range(40):
        T, n_new, graph, masked = 1 + r(300), 1 + r(40), bool(r(2)), r(4) == 0
        calls.append((T, n_new, graph, masked, 100 + i))
    calls += [calls[j] for j in (3, 17, 0, 29, 8, 21, 35, 12, 5, 38, 26, 1, 33, 14, 9, 30, 19, 7, 24, 11)]      # repeats, in another order
    for k, (T, n_new, graph, masked, seed) in enumerate(calls):
        ids = torch.randint(3, V - 8, (1, T), generator=torch.Generator().manual_seed(seed)).to(m.device)
        am = None
        if masked and T > 4:
            am = torch.ones(1, T, dtype=torch.int64, device=m.device)
            am[0, :1 + (seed % (T // 2))] = 0                                   # left padding
        toks = m.generate(input_ids=ids, attention_mask=am, max_new_tokens=n_new, do_sample=False, eos_token_id=None, use_graph=graph)
        assert toks.shape == (1, n_new) and int(toks.min()) >= 0 and int(toks.max()) < V, (k, T, n_new, toks.shape)
        key = (T, n_new, masked, seed)                                          # graph or eager: the same tokens
        if key in first:
            assert torch.equal(first[key], toks.cpu()), (k, key)
        first[key] = toks.cpu()


def test_split_attention_equals_the_one_cu_walk(small7b):
    """args.split_min (VCLA_ENGINE_SPLIT): above it the 8 CUs of a head's group share the walk over its cached keys and the owner merges their (o, m, l);
    below it one CU walks alone.  Same keys, same arithmetic per key, another summation order: the logits of the two forms must agree to fp32 rounding of the
    softmax sums (far inside the bf16 bound of the other tests) at contexts either side of the default threshold."""
    m = small7b
    for T in (300, 700, 1500):
        got = {}
        for mode in ("0", "128"):                       # never split | split from 128 cached keys on
            os.environ["VCLA_ENGINE_SPLIT"] = mode
            try:
                got[mode] = _steps(m, "1", T, 3, masked=(T == 700))
            finally:
                os.environ.pop("VCLA_ENGINE_SPLIT", None)
        for s in range(3):
            d = (got["0"][1][s] - got["128"][1][s]).abs()
            assert d.max().item() < 0.06 and d.mean().item() < 0.01, (T, s, d.max().item(), d.mean().item())
        assert (got["0"][2] - got["128"][2]).abs().max().item() < 0.05          # the appended K / V rows
