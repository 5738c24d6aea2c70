#!/usr/bin/env python
"""bench.py -- VisualCLA-7B hot path on MI355X: output tokens/sec + images/sec, with both rooflines and the CPU baseline.

One "step" = one pass of the hot path over one batch of synthetic requests:
    ViT-L/14 (224 px) -> Resampler -> projection -> splice -> LLaMA-7B prefill (T=128) -> 128-token greedy decode.
Four workloads are timed per default run (SURVEY.md section 8d "two regimes, report both"; every leg carries BOTH images/sec definitions:
`images_per_sec` = whole requests, `images_per_sec_prefill` = images through ViT + Resampler + projection + prefill):
  * configs[1] -- `value`: B = 1 request per GPU (BASELINE.json configs[1]; the latency / HBM-bound regime), EXACTLY --steps steps;
  * configs[2] -- `images_per_sec`, `config2`: B = 64 requests per GPU (BASELINE.json configs[2]; ViT + Resampler throughput,
    the MFMA-bound vision / prefill regime next to batch decode), a few steps of its own (--steps-b64);
  * configs[4]'s per-GPU share -- `config4`: fp8 weights, 336 px, B = 32 = 256 / 8, in both numeric modes (--steps-c4);
  * `strong256` -- north_star's scaling workload: a GLOBAL batch of 256 (bf16, 224 px) split over the ranks, B = 256 / N per GPU
    (--steps-strong): a driver series --gpus 1 / 2 / 4 / 8 reads ">= 6x images/sec at batch 256" off this leg.
Weights are random-init of the 7B architecture, inputs synthetic and resident in HBM before the timed
region.

Multi-GPU: `python bench.py --gpus N` spawns N ranks itself (torch.distributed.run, one process per GPU, 127.0.0.1 rendezvous)
unless it already runs under a launcher (WORLD_SIZE set), in which case it is one of the ranks.  Full replica per rank, the
batch is sharded by rank (weak scaling: per-GPU batch fixed), the only collective is ONE RCCL all-gather of the generated ids
per step, inside the timed region.  `--plumbing-check` runs the same launch / shard / gather / max-over-ranks path on CPU
(gloo) with a token-pattern stand-in for the model: tests/test_host_cpu.py uses it; it measures nothing.

Prints ONE JSON line on rank 0: throughput, `roofline` (the dominant kernel of the timed configs[1] workload, HBM regime),
`rooflines` (all regimes: whole-step decode bytes vs 8 TB/s for B = 1 and B = 64, the batch-decode weight-streaming GEMM, the
ViT MFMA GEMM vs 2.5 PFLOP/s, the vision stack as a whole) and `cpu_baseline` (the CPU oracle timed on this host).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16 MFMA peak
KV_BYTES_PER_TOKEN = 512 * 1024  # bf16 K + V of one context token over the 32 layers


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="requests per GPU of the main timed workload (1 = configs[1], 64 = configs[2])")
    ap.add_argument("--steps-b64", type=int, default=5, help="timed steps of the second workload (configs[2], B = 64 per GPU); 0 = skip it")
    ap.add_argument("--steps-c4", type=int, default=2, help="timed steps of the third workload (BASELINE configs[4] per GPU: fp8 weights, 336 px, "
                    "B = 32 = 256 / 8); 0 = skip it")
    ap.add_argument("--steps-strong", type=int, default=2, help="timed steps of the fourth workload `strong256`: north_star's scaling claim -- a GLOBAL batch of 256 "
                    "requests (bf16, 224 px) split evenly over the ranks (B = 256 at N = 1 ... B = 32 at N = 8); 0 = skip it")
    ap.add_argument("--steps-strong-c4", type=int, default=1, help="timed steps of `strong256_fp8_336`: the same split of a GLOBAL batch of 256 in BASELINE configs[4]'s mode "
                    "(fp8 weights W8A16, 336 px) -- the N = 1 denominator of configs[4]'s 8-GPU claim; 0 = skip it")
    ap.add_argument("--global-batch", type=int, default=0, help="STRONG scaling: this many requests in total, split evenly over the ranks "
                    "(e.g. 256 = north_star's '>= 6x images/sec 1 -> 8 GPUs at batch 256'); replaces --batch, reports scaling = strong")
    ap.add_argument("--prompt-len", type=int, default=128)
    ap.add_argument("--new-tokens", type=int, default=128)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay in decode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp8", action="store_true", help="BASELINE configs[4] weight path: fp8 (e4m3) decode weights, bf16 activations")
    ap.add_argument("--fp8-kv", type=int, default=1, help="with the fp8 weight path (--fp8 and the config4 leg): 1 = the K/V cache holds e4m3 bytes too, 0 = bf16 cache")
    ap.add_argument("--image-size", type=int, default=224, help="336 = BASELINE configs[4] patching (577 ViT tokens, position embedding grown bicubically)")
    ap.add_argument("--sample", action="store_true", help="decode under the reference's DEFAULT_GENERATION_CONFIG (sampling + "
                    "penalties, on-device sampler) instead of greedy; a side measurement, not BASELINE's metric")
    ap.add_argument("--cpu-tokens", type=int, default=32, help="decode tokens of the CPU baseline run (32 = BASELINE configs[0], timed in full)")
    ap.add_argument("--plumbing-check", action="store_true", help="CPU / gloo check of the multi-rank launch, sharding, gather and timing path (no model)")
    return ap.parse_args(argv)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def respawn_ranks(args) -> int:
    """`bench.py --gpus N` outside a launcher: become the launcher (one rank per GPU, rendezvous on 127.0.0.1)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on these hosts (RCCL needs it)
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------------- rooflines
def _event_time(fn, reps: int):
    """seconds per call of fn(), HIP events on the stream the kernels are launched on (torch's current stream)"""
    import torch
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


PROFILE_ROUNDS = ("r06", "r05", "r04", "r03", "r02")     # committed rocprofv3 summaries, newest first


def _pmc_traffic(stem: str = "pmc_gemv1p"):
    """HBM bytes per launch of a kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in their own
    runs, FETCH_SIZE doubled per the gfx950 correction; tools/gpu_check.sh pmc).  Not collectable inside this run."""
    import re
    for rnd in PROFILE_ROUNDS:
        try:
            tot = 0.0
            for nm in ("fetch_size", "write_size"):
                m = re.search(r"-> ([0-9.]+) MB per launch", open(os.path.join(ROOT, "profiles", f"{rnd}_{stem}_{nm}.txt")).read())
                tot += float(m.group(1)) * 1e6
            return int(tot), f"profiles/{rnd}_{stem}_{{fetch,write}}_size.txt (separate rocprofv3 --pmc passes)"
        except Exception:
            continue
    return None, None


def _in_model_us(table: str, kernel_substr: str, grid: int):
    """average duration of one kernel INSIDE the timed workload, from the committed rocprofv3 --kernel-trace summary of this same
    command grouped by (kernel, grid) (tools/prof_by_grid.py): the figure the back-to-back HIP-event loop below is ~3 % optimistic
    against (no neighbouring kernels, no cold first lines)"""
    import re
    for rnd in PROFILE_ROUNDS:
        try:
            for line in open(os.path.join(ROOT, "profiles", f"{rnd}_{table}")):
                m = re.match(r"\s*([0-9.]+) us x\s*(\d+)\s+[0-9.]+%\s+grid=\s*(\d+)", line)
                if m and kernel_substr in line and int(m.group(3)) == grid:
                    return float(m.group(1)), f"profiles/{rnd}_{table} ({m.group(2)} launches)"
        except Exception:
            continue
    return None, None


def _label_static(d: dict) -> dict:
    """`achieved`, `frac`, `avg_launch_us` are measured live in this run (HIP events); the fields named here are READ from committed
    rocprofv3 summaries under profiles/ (PMC counters and per-kernel durations cannot be collected inside the timed process)."""
    static = [k for k in ("traffic", "avg_launch_us_in_model", "frac_in_model") if d.get(k) is not None]
    if static:
        d["static_fields"] = {"fields": static, "source": "committed profile (see traffic_source / in_model_source), not measured in this run"}
    return d


def gemv_roofline(model, n_rep: int = 20):
    """Dominant kernel of the B = 1 workload: the gate/up SwiGLU GEMV with fused RMSNorm (gemv1p_kernel<R=2,K=4096,SWIGLU>,
    gemv_decode.hip; ~36 % of the decode time, 43 % of the weight bytes).  One launch streams W_gu [2*11008, 4096] bf16 exactly once: algorithmic
    bytes = 180.4 MB.  The 32 layers' matrices are launched back to back (5.8 GB footprint, so nothing is served from the 256 MB
    Infinity Cache) between two HIP events; achieved = bytes / mean launch duration (inter-launch gaps included)."""
    import torch
    from visualcla import _lib
    t = model.config.text_config
    D, I, L = t["hidden_size"], t["intermediate_size"], t["num_hidden_layers"]
    P = model._packed
    x = torch.randn(1, D, device=model.device).to(torch.bfloat16)
    out = torch.empty(1, I, dtype=torch.bfloat16, device=model.device)
    alg_bytes = 2 * I * D * 2

    def run():
        for l in range(L):
            _lib.gemm(x, P[f"llama.l{l}.wgu"], 2 * I, out=out, epilogue=_lib.EPI_SWIGLU, force_kernel=2, norm_gamma=P[f"llama.l{l}.ln2.g"], norm_eps=1e-6)
    sec = _event_time(run, n_rep) / L
    achieved = alg_bytes / sec / 1e9
    traffic, src = _pmc_traffic()
    us_model, us_src = _in_model_us("bench_b1_by_grid.txt", "gemv1p_kernel<2, 4096, true", 262144)
    out = {"bound": "hbm", "kernel": "gemv1p_kernel<R=2,K=4096,SWIGLU> (B=1 gate/up GEMV + fused RMSNorm, bf16, persistent)",
           "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
           "traffic": traffic, "traffic_source": src, "alg_bytes_per_launch": alg_bytes, "avg_launch_us": round(sec * 1e6, 2),
           "launches_timed": n_rep * L}
    if us_model:     # the same kernel inside the decode graph (rocprofv3 of this command): what the step really pays per launch
        out.update(avg_launch_us_in_model=us_model, frac_in_model=round(alg_bytes / (us_model * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), in_model_source=us_src)
    return _label_static(out)


def engine_roofline(model, prompt_len: int = 128, n_steps: int = 96):
    """Dominant kernel of the B = 1 workload since round 6: decode_engine_kernel (csrc/decode_engine.hip), ONE persistent launch per token that streams
    every LLaMA linear weight and the lm_head exactly once (algorithmic bytes = those matrices, 13.36 GB at 7B, + the K/V rows of the context, 512 KiB per
    cached token) -- ~99 % of the decode time.  Measured live: n_steps hipGraph-replayed decode steps between two HIP events on the stream they run on; a
    step is the engine launch plus the argmax and the token hand-off launch (two sub-10-us kernels), so the figure is the engine's own duration rounded UP.
    None when the model has no engine stream or VCLA_ENGINE=0 (the launch path then; gemv_roofline describes it)."""
    import torch
    from visualcla import _lib
    if "llama.engine.w" not in model._packed or os.environ.get("VCLA_ENGINE", "1") == "0" or model.fp8_decode:
        return None
    lib = _lib.load()
    t = model.config.text_config
    D, I, L, V, H = t["hidden_size"], t["intermediate_size"], t["num_hidden_layers"], t["vocab_size"], t["num_attention_heads"]
    dev = model.device
    ids = torch.randint(3, V - 8, (1, prompt_len), generator=torch.Generator().manual_seed(11)).to(dev)
    ctx_max = (prompt_len + n_steps + 2 + 63) // 64 * 64
    embeds, _ = model._embed(ids, None, None)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        cache = model._new_cache(1, ctx_max, _persistent=True)
        model._prefill(embeds, cache, None, all_logits=False, _persistent=True)
        ws = model._buf("llama", lib.vcla_llama_workspace_bytes(model._ctx, 1, 1))
        out = model._typed_buf("gen_out", (n_steps + 1, 1), torch.int64)
        out[0] = 17

        def loop():
            model._pos_dev.zero_()
            _lib.check(lib.vcla_llama_decode_loop(model._ctx, out[0].data_ptr(), 1, prompt_len, model._pos_dev.data_ptr(), n_steps, cache.kv.data_ptr(), ctx_max,
                                                  None, out[1:].data_ptr(), ws.data_ptr(), ws.numel(), 1, _lib.stream_ptr()))
        loop()                                   # captures the step graph
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 2
        e0.record(stream)
        for _ in range(reps):
            loop()
        e1.record(stream)
        torch.cuda.synchronize()
        _lib.check(lib.vcla_llama_decode_status(model._ctx, 1, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    sec = e0.elapsed_time(e1) * 1e-3 / (reps * n_steps)
    w_bytes = 2 * (L * (3 * D * D + D * D + 2 * I * D + D * I) + V * D)
    kv_bytes = (prompt_len + (n_steps + 1) / 2.0) * KV_BYTES_PER_TOKEN
    alg = w_bytes + kv_bytes
    ach = alg / sec / 1e9
    out_ = {"bound": "hbm", "kernel": "decode_engine_kernel (B=1: the whole decode step -- 32 layers + lm_head -- as ONE persistent launch: per CU an LDS-DMA loader wave "
            "+ 3 consumer waves, bf16; timed with the step's argmax and token hand-off launches)",
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
            "alg_bytes_per_launch": int(alg), "alg_bytes_note": "LLaMA linear weights + lm_head once (%.3f GB) + K/V rows of the mean context (%.1f MB)" % (w_bytes / 1e9, kv_bytes / 1e6),
            "avg_launch_us": round(sec * 1e6, 1), "launches_timed": reps * n_steps}
    traffic, src = _pmc_traffic("pmc_engine")
    if traffic:
        out_.update(traffic=traffic, traffic_source=src)
    us_model, us_src = _in_model_us("bench_b1_by_grid.txt", "decode_engine_kernel", 65536)
    if us_model:
        out_.update(avg_launch_us_in_model=us_model, frac_in_model=round(alg / (us_model * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), in_model_source=us_src)
    return _label_static(out_)


def batch_decode_gemm_roofline(model, M: int = 64, n_rep: int = 10):
    """Dominant kernel of the B = 64 decode step: the gate/up SwiGLU streaming GEMM (gemm_dstream_kernel, fragment-major W, M = 64
    rows): 22016 x 4096 bf16 = 180.4 MB streamed once per launch, over the 32 layers' matrices back to back."""
    import torch
    from visualcla import _lib
    t = model.config.text_config
    D, I, L = t["hidden_size"], t["intermediate_size"], t["num_hidden_layers"]
    P = model._packed
    alg_bytes = 2 * I * D * 2
    if M > 64:
        # 65 - 128 decode rows: the split-K panel kernel; 129 - 256: the ring kernel (gemm_ring.hip).  The weights are streamed once per launch, so the
        # bound quoted is HBM; at M = 256 neither HBM nor MFMA binds (46 GF and 180 MB in ~57 us = 0.8 PF/s and 3.2 TB/s) but what a CU can keep in
        # flight against the loaded round trip, DESIGN.md section 5 "Round 5"
        x = torch.randn(M, D, device=model.device).to(torch.bfloat16)
        out = torch.empty(M, I, dtype=torch.bfloat16, device=model.device)
        ws = torch.zeros(64 << 20, dtype=torch.uint8, device=model.device)

        # the fragment-major twin, as the engine passes it: M <= 128 -> the split-K panel kernel streams it; 129 - 256 -> the ring kernel takes its weight pieces from it
        use_frag = f"llama.l0.wgu.f" in P

        def run_tiles():
            for l in range(L):
                _lib.gemm(x, P[f"llama.l{l}.wgu"], 2 * I, out=out, epilogue=_lib.EPI_SWIGLU, splitk_ws=ws,
                          w_frag=P[f"llama.l{l}.wgu.f"] if use_frag else None)
        sec = _event_time(run_tiles, n_rep) / L
        achieved = alg_bytes / sec / 1e9
        return {"bound": "hbm", "kernel": (f"gemm_ring_kernel<SWIGLU, 256x96> (B={M} gate/up decode GEMM: full-K tiles fed by an LDS-DMA ring, weight pieces from the "
                                                 f"{'fragment-major twin' if use_frag else 'row-major matrix'}, bf16; the default dispatch for 129 - 256 rows)" if M > 128 else
                                                 (f"gemm_panel_kernel<SWIGLU> (B={M} gate/up split-K panel GEMM on the fragment-major copy, bf16)" if use_frag else
                                                  f"vcla_gemm default dispatch on the row-major weights (B={M} gate/up decode GEMM, bf16)")),
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "alg_bytes_per_launch": alg_bytes,
                "avg_launch_us": round(sec * 1e6, 2), "launches_timed": n_rep * L, "mfma_tflops": round(2.0 * M * 2 * I * D / sec / 1e12, 1)}
    if f"llama.l0.wgu.f" not in P:
        return None
    af = _lib.to_frag(torch.randn(M, D, device=model.device).to(torch.bfloat16))
    cf = torch.zeros(I // 32, (M + 15) // 16, 64, 8, dtype=torch.bfloat16, device=model.device)
    out = torch.empty(M, I, dtype=torch.bfloat16, device=model.device)

    def run():
        for l in range(L):
            _lib.gemm(None, P[f"llama.l{l}.wgu"], 2 * I, out=out, epilogue=_lib.EPI_SWIGLU, force_kernel=9, a_frag=af, m=M,
                      w_frag=P[f"llama.l{l}.wgu.f"], c_frag=cf)
    sec = _event_time(run, n_rep) / L
    achieved = alg_bytes / sec / 1e9
    traffic, src = _pmc_traffic("pmc_dstream") if M == 64 else (None, None)
    out = {"bound": "hbm", "kernel": f"gemm_dstream_kernel<SWIGLU,MT=4> (B={M} gate/up streaming GEMM, bf16)", "achieved": round(achieved, 1),
           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": src,
           "alg_bytes_per_launch": alg_bytes, "avg_launch_us": round(sec * 1e6, 2), "launches_timed": n_rep * L}
    us_model, us_src = _in_model_us("bench_b64_by_grid.txt", "gemm_dstream_kernel<3", 131072) if M == 64 else (None, None)
    if us_model:
        out.update(avg_launch_us_in_model=us_model, frac_in_model=round(alg_bytes / (us_model * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), in_model_source=us_src)
    return _label_static(out)


def vit_gemm_roofline(model, B: int = 64, n_rep: int = 20):
    """MFMA regime: the ViT fc1 GEMM of the B = 64 workload (M = 64*257 = 16448 rows, N = 4096, K = 1024, quick-GELU epilogue;
    the largest single kernel of the vision stack) through the product dispatch: algorithmic flops = 2 M N K per launch."""
    import torch
    from visualcla import _lib
    v = model.config.vision_config
    D, I = v["hidden_size"], v["intermediate_size"]
    N = (v["image_size"] // v["patch_size"]) ** 2 + 1
    M = B * N
    P = model._packed
    a = torch.randn(M, D, device=model.device).to(torch.bfloat16)
    out = torch.empty(M, I, dtype=torch.bfloat16, device=model.device)
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=model.device)
    L = v["num_hidden_layers"]

    def run():
        for l in range(L):
            _lib.gemm(a, P[f"vit.l{l}.w1"], I, bias=P[f"vit.l{l}.b1"], out=out, epilogue=_lib.EPI_QUICK_GELU, splitk_ws=ws)
    sec = _event_time(run, n_rep) / L
    flops = 2.0 * M * I * D
    tf = flops / sec / 1e12
    out = {"bound": "mfma", "kernel": f"gemm_mfma256_kernel<QUICK_GELU> (ViT fc1, M={M} N={I} K={D}; one launch of 257-row tiles, L2-prefetch form)",
           "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TF, 4), "traffic": None,
           "alg_flops_per_launch": flops, "avg_launch_us": round(sec * 1e6, 2), "launches_timed": n_rep * L}
    # the same launch inside the B = 64 workload (grid = B x N/256 tiles of 512 threads)
    us_model, us_src = _in_model_us("bench_b64_by_grid.txt", "gemm_mfma256_kernel<1", B * (I // 256) * 512) if B == 64 else (None, None)
    if us_model:
        out.update(avg_launch_us_in_model=us_model, frac_in_model=round(flops / (us_model * 1e-6) / 1e12 / MFMA_BF16_PEAK_TF, 4), in_model_source=us_src)
    return _label_static(out)


def fp8_gemm_roofline(model, M: int = 8192, n_rep: int = 5):
    """BASELINE configs[4] regime: the prefill gate/up GEMM (M = 64 x 128 rows, N = 22016, K = 4096) fp8 x fp8 on the fp8 MFMA
    pipe (gemm_mfma256_fp8_kernel, v_mfma_scale_f32_16x16x128_f8f6f4) against the 5 PFLOP/s dense fp8 peak, the per-row
    activation quantisation pass included in the timed region (it is part of every such GEMM in the product)."""
    import torch
    from visualcla import _lib
    t = model.config.text_config
    D, I, L = t["hidden_size"], t["intermediate_size"], t["num_hidden_layers"]
    P = model._packed
    if "llama.l0.wgu.q8" not in P:
        return None
    a = torch.randn(M, D, device=model.device).to(torch.bfloat16)
    out = torch.empty(M, I, dtype=torch.bfloat16, device=model.device)
    nl = min(L, 8)

    def run():
        for l in range(nl):
            aq, as_ = _lib.quant_fp8_rows(a)
            _lib.gemm(None, P[f"llama.l{l}.wgu"], 2 * I, out=out, epilogue=_lib.EPI_SWIGLU, force_kernel=10, a_q8=aq, a_scale=as_,
                      w_q8=P[f"llama.l{l}.wgu.q8"], w_scale=P[f"llama.l{l}.wgu.s8"])
    sec = _event_time(run, n_rep) / nl
    flops = 2.0 * M * 2 * I * D
    tf = flops / sec / 1e12
    return {"bound": "mfma", "kernel": f"gemm_mfma256_fp8_kernel<SWIGLU> (LLaMA gate/up prefill, M={M} N={2 * I} K={D}, fp8 x fp8, + activation quantisation)",
            "achieved": round(tf, 1), "peak": 5000.0, "unit": "TFLOP/s", "frac": round(tf / 5000.0, 4), "traffic": None,
            "alg_flops_per_launch": flops, "avg_launch_us": round(sec * 1e6, 2), "launches_timed": n_rep * nl}


def step_rooflines(b1, b64, cfgd, fp8: bool, kv8: bool = False, other=None):
    """Whole-step figures, so the line cannot quote only its best kernel: decode bytes per step = all LLaMA linear weights +
    lm_head read once (13.36 GB bf16 / 6.68 GB fp8, shared by the batch) + B * ctx * 512 KiB of KV cache (ctx = mean context
    over the decode steps); vision flops = 179.2 GF per image at 224 px (ViT 162.0 + resampler 16.64 + projection 0.54)."""
    T, n_new = cfgd["seq_len"], cfgd["new_tokens"]
    w_bytes = 13.36e9 / (2 if fp8 else 1)
    ctx = T + (n_new + 1) / 2.0
    out = []
    for tag, br in (("B=1", b1), ("B=64", b64)) + ((("B=%d" % other["batch_per_gpu"], other),) if other else ()):
        if not br:
            continue
        B = br["batch_per_gpu"]
        bytes_step = w_bytes + B * ctx * KV_BYTES_PER_TOKEN / (2 if (fp8 and kv8) else 1)
        ach = bytes_step / (br["breakdown_ms"]["decode_ms_per_token_step"] * 1e-3) / 1e9
        out.append({"bound": "hbm", "kernel": f"whole decode step, {tag} per GPU (all kernels + launch gaps)", "achieved": round(ach, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "alg_bytes_per_step": int(bytes_step), "ms_per_step": br["breakdown_ms"]["decode_ms_per_token_step"]})
        if cfgd.get("image_size", 224) == 224:
            fl = B * 179.2e9
            tf = fl / (br["breakdown_ms"]["vision_ms"] * 1e-3) / 1e12
            out.append({"bound": "mfma", "kernel": f"whole vision stack (ViT + resampler + projection), {tag} per GPU", "achieved": round(tf, 1),
                        "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TF, 4), "traffic": None,
                        "alg_flops": fl, "ms": br["breakdown_ms"]["vision_ms"]})
    return out


# ---------------------------------------------------------------------------------------------------------------- fp8 accuracy
def fp8_mode_accuracy(model, prefill: bool, kv_cache: bool, B: int = 2, T: int = 128, n_new: int = 4):
    """How far the fp8 weight path's logits are from the bf16 path's ON THE SAME MODEL AND INPUTS (configs[4] reports this next to its
    throughput): prefill logits (last position) and the first n_new - 1 decode steps, the fp8 run teacher-forced through the bf16 run's
    tokens so that every step sees the same inputs.  Returns cosine similarity and mean |error| in units of the logit standard
    deviation, worst over the compared steps, separately for the prefill and the decode steps.  Leaves the model in bf16 mode."""
    import torch
    from transformers import LogitsProcessorList
    from visualcla.synthetic import make_inputs
    px, ids, mask = make_inputs(model.config, B, T)
    px, ids, mask = px.to(model.device, torch.bfloat16), ids.to(model.device), mask.to(model.device)

    def run(force):
        seen = []

        def grab(ids_, scores):
            seen.append(scores.detach().float().clone())
            return scores

        def teacher(ids_, scores):
            if force is None:
                return scores
            step = len(seen) - 1
            out = torch.full_like(scores, -1e30)
            out.scatter_(1, force[:, step:step + 1], 0.0)
            return out
        toks = model.generate(input_ids=ids, pixel_values=px, attention_mask=mask, max_new_tokens=n_new, do_sample=False, eos_token_id=None,
                              logits_processor=LogitsProcessorList([grab, teacher]))
        return seen, toks
    model.enable_fp8_decode(False)
    base, toks = run(None)
    model.enable_fp8_decode(True, prefill=prefill, kv_cache=kv_cache)
    try:
        got, _ = run(toks)
    finally:
        model.enable_fp8_decode(False)

    def dist(a, b):
        std = b.std().item()
        cos = torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
        return cos, (a - b).abs().mean().item() / std
    pre = dist(got[0], base[0])
    dec = [dist(got[s], base[s]) for s in range(1, n_new)]
    return {"vs": "bf16 path, same model and inputs, teacher-forced", "batch": B, "prompt_len": T,
            "prefill": {"cosine": round(pre[0], 4), "mean_over_sigma": round(pre[1], 4)},
            "decode_steps": {"cosine": round(min(c for c, _ in dec), 4), "mean_over_sigma": round(max(m for _, m in dec), 4), "steps": n_new - 1}}


# ---------------------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(model, prompt_len: int, n_tokens: int):
    """BASELINE configs[0] on this host's cores = "the reference CPU HuggingFace path" north_star names: oracle/hf_cpu_baseline.py in a child
    process (transformers' own CLIPVisionModel + LlamaForCausalLM.generate in fp32 with the restated resampler between them, pinned to one
    memory node, decode thread count swept; kind "hf+port-resampler").  The child is test infrastructure under oracle/, run here and
    nowhere else; if it cannot run (no transformers, not enough host RAM) the restated port below is timed instead (kind "port")."""
    here = os.path.dirname(os.path.abspath(__file__))
    try:
        r = subprocess.run([sys.executable, os.path.join(here, "oracle", "hf_cpu_baseline.py"), "--prompt-len", str(prompt_len), "--tokens", str(n_tokens)],
                           capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and line:
            return json.loads(line[-1])
        why = (r.stderr or r.stdout)[-300:]
    except Exception as e:  # noqa: BLE001
        why = f"{type(e).__name__}: {e}"
    res = cpu_baseline_port(model, prompt_len, n_tokens)
    res["sample"] += f" [HF child failed: {why}]"
    return res


def cpu_baseline_port(model, prompt_len: int, n_tokens: int):
    """fallback of cpu_baseline(): the CPU oracle itself (kind "port": oracle/visualcla_oracle.py = the reference's glue + the
    transformers arithmetic restated, pinned to the reference's own outputs in tests/golden; /root/reference itself does not
    exist on the GPU box) runs ONE request end to end in fp32 -- vision stack, prefill of the T = 128 prompt, `n_tokens` greedy
    decode steps, all timed, nothing extrapolated.  32 threads: one NUMA domain's worth; torch's CPU kernels collapse when spread
    over all 256 SMT threads of the 2-socket host (measured 28 s/token at 256 threads vs 0.8 s at 32)."""
    import torch
    from oracle import visualcla_oracle as O   # the ONLY place this file touches oracle/: the CPU leg is the oracle, timed
    cfg_o = O.cfg_7b()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    t0 = time.time()
    W = model.state_dict()                      # bf16-rounded values, fp32 on the host (~27 GB)
    t_unpack = time.time() - t0
    px, ids, mask = O.make_inputs(cfg_o, 1, prompt_len)
    with torch.no_grad():
        t_begin = time.time()
        img = O.image_embeds(px, W, cfg_o)
        t_vis = time.time() - t_begin
        x = O.embed_and_splice(ids, img, W, cfg_o)
        cache = [None] * cfg_o.text.num_hidden_layers
        t0 = time.time()
        h = O.llama_forward(x, W, cfg_o.text, mask, cache, 0)
        logits = O.lm_head(h[:, -1:], W)[:, 0]
        t_pre = time.time() - t0
        t0 = time.time()
        past = prompt_len
        produced = 1                              # the prefill's argmax is the first new token
        for _ in range(n_tokens - 1):
            nxt = logits.argmax(-1)
            e = W["text_model.model.embed_tokens.weight"][nxt][:, None, :]
            m = torch.ones(1, past + 1, dtype=torch.int64)
            h = O.llama_forward(e, W, cfg_o.text, m, cache, past)
            logits = O.lm_head(h, W)[:, 0]
            past += 1
            produced += 1
        t_dec = time.time() - t0
        total = time.time() - t_begin
    return {"value": round(produced / total, 3), "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "images_per_sec": round(1.0 / total, 4),
            "sample": (f"BASELINE configs[0] in full: 1 image, T={prompt_len} prompt, {produced} greedy tokens in {total:.1f}s "
                       f"(vision stack {t_vis:.2f}s, prefill {t_pre:.2f}s, {produced - 1} decode steps {t_dec:.1f}s = {t_dec / max(produced - 1, 1):.3f}s/token); "
                       f"fp32 oracle on {torch.get_num_threads()} of {os.cpu_count()} host threads, weights unpacked in {t_unpack:.1f}s (not timed)")}


# ---------------------------------------------------------------------------------------------------------------- timing
def timed_workload(step, steps: int, warmup: int, world: int, sync, barrier, allmax):
    for _ in range(warmup):
        step()
    barrier()
    sync()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
    sync()
    barrier()
    dt = time.perf_counter() - t0
    return allmax(dt), out


def plumbing_check(args, rank, world):
    """The launch / shard / all-gather / max-over-ranks path with a stand-in for generate(): token (r, j) of global request r is
    r * 1000 + j.  CPU, gloo.  Prints the same JSON skeleton; `value` is meaningless."""
    import torch
    import torch.distributed as dist
    from visualcla.distributed import gather_tokens, shard_range
    if world > 1:
        dist.init_process_group(backend="gloo")
    B, n_new = max(args.batch, 3), 8
    gB = B * world
    lo, hi = shard_range(gB, rank, world)

    def step():
        toks = (torch.arange(lo, hi)[:, None] * 1000 + torch.arange(n_new)[None, :]).to(torch.int64)
        return gather_tokens(toks, n_total=gB, n_cols=n_new) if world > 1 else toks
    barrier = (lambda: dist.barrier()) if world > 1 else (lambda: None)

    def allmax(dt):
        if world == 1:
            return dt
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    dt, out = timed_workload(step, args.steps, args.warmup, world, lambda: None, barrier, allmax)
    want = torch.arange(gB)[:, None] * 1000 + torch.arange(n_new)[None, :]
    assert torch.equal(out, want), (rank, out, want)
    # the `strong256` leg of the real run: a GLOBAL batch of 256 split over the ranks (shard sizes 256 // world), gathered the same way
    strong = None
    if args.steps_strong > 0 and 256 % world == 0:
        slo, shi = shard_range(256, rank, world)
        assert shi - slo == 256 // world

        def sstep():
            toks = (torch.arange(slo, shi)[:, None] * 1000 + torch.arange(n_new)[None, :]).to(torch.int64)
            return gather_tokens(toks, n_total=256, n_cols=n_new) if world > 1 else toks
        sdt, sout = timed_workload(sstep, args.steps_strong, 1, world, lambda: None, barrier, allmax)
        assert torch.equal(sout, torch.arange(256)[:, None] * 1000 + torch.arange(n_new)[None, :])
        pdt, _ = timed_workload(sstep, 1, 0, world, lambda: None, barrier, allmax)        # stands in for the first-token generate()
        strong = {"batch_per_gpu": 256 // world, "global_batch": 256, "steps": args.steps_strong, "scaling": "strong",
                  "images_per_sec": round(256 * args.steps_strong / sdt, 1), "images_per_sec_prefill": round(256 / pdt, 1)}
    if rank == 0:
        out = {"metric": "plumbing check (no model)", "value": round(gB * n_new * args.steps / dt, 1), "unit": "tokens/s",
               "images_per_sec": round(gB * args.steps / dt, 1), "images_per_sec_prefill": round(gB * args.steps / dt, 1),
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": "weak",
               "config": {"workload": "token-pattern stand-in", "global_batch": gB, "parallelism": f"dp{world}"}}
        if strong:
            out["strong256"] = strong
            if args.steps_strong_c4 > 0:      # the real run times the same split once more in configs[4]'s mode (fp8 W8A16, 336 px)
                out["strong256_fp8_336"] = dict(strong, steps=args.steps_strong_c4, mode="w8a16")
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.plumbing_check:
        return plumbing_check(args, rank, world)

    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import visualcla
    from visualcla.distributed import gather_tokens, shard_range
    from visualcla.synthetic import make_inputs, stub_tokenizer

    cfg = visualcla.visualcla_7b_config()
    model = visualcla.VisualCLAModel.from_random(cfg, device=dev, torch_dtype=torch.bfloat16, seed=0)
    model.tokenizer = stub_tokenizer()
    model.image_at_head = False
    if args.fp8:
        model.enable_fp8_decode(kv_cache=bool(args.fp8_kv))
    if args.image_size != 224:
        model.set_image_size(args.image_size)

    sync = torch.cuda.synchronize
    barrier = (lambda: dist.barrier()) if world > 1 else (lambda: None)

    def allmax(dt):
        if world == 1:
            return dt
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def run_workload(B, steps, warmup):
        """B requests per GPU: every rank builds ONLY its shard of the global batch (request r is seeded by r, so a shard does
        not depend on the world size), generate() + one all-gather of the ids per step.  Two throughput definitions per leg:
        `images_per_sec` = whole requests (vision + prefill + all decode steps), `images_per_sec_prefill` = SURVEY.md 8(d)'s "images through
        ViT + Resampler + projection + prefill" (one generate(max_new_tokens=1) per batch, timed on every rank, max over ranks)."""
        gB = B * world
        lo, hi = shard_range(gB, rank, world)
        px, ids, mask = make_inputs(model.config, hi - lo, args.prompt_len, first_request=lo)
        px, ids, mask = px.to(dev, torch.bfloat16), ids.to(dev), mask.to(dev)
        kw = dict(input_ids=ids, pixel_values=px, attention_mask=mask, max_new_tokens=args.new_tokens, do_sample=False,
                  eos_token_id=None, use_graph=not args.no_graph)
        if args.sample:   # models/visualcla/modeling_utils.py:36-47
            kw.update(do_sample=True, top_p=0.9, top_k=40, temperature=0.5, repetition_penalty=1.1, no_repeat_ngram_size=15)

        def step():
            toks = model.generate(**kw)
            return gather_tokens(toks, n_total=gB, n_cols=args.new_tokens) if world > 1 else toks
        dt, out = timed_workload(step, steps, warmup, world, sync, barrier, allmax)
        assert out.shape == (gB, args.new_tokens), out.shape
        res = {"batch_per_gpu": B, "global_batch": gB, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 2),
               "tokens_per_sec": round(gB * args.new_tokens * steps / dt, 2), "images_per_sec": round(gB * steps / dt, 4)}

        def timed(fn, reps=2):       # barrier + sync on both sides, max over ranks: the same bracket as the main measurement
            fn()
            barrier()
            sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            sync()
            barrier()
            return allmax(time.perf_counter() - t0) / reps * 1e3
        # stage split of one step: vision stack, + splice / prefill / first token, + decode.  The vision stack as generate() runs it: on a side
        # stream with persistent buffers (the engine replays its captured graph there; on the default stream it would issue ~250 launches eagerly)
        vis_stream = torch.cuda.Stream(device=dev)

        def vision_once():
            with torch.cuda.stream(vis_stream):
                model.embed_images(px, _persistent=not args.no_graph)
        t_vis = timed(vision_once, reps=3)
        t_pre = timed(lambda: model.generate(**dict(kw, max_new_tokens=1)), reps=3)
        t_step = dt / steps * 1e3
        res["images_per_sec_prefill"] = round(gB / (t_pre * 1e-3), 3)
        res["breakdown_ms"] = {"vision_ms": round(t_vis, 2), "prefill_first_token_ms": round(t_pre - t_vis, 2),
                               "decode_ms": round(t_step - t_pre, 2),
                               "decode_ms_per_token_step": round((t_step - t_pre) / max(args.new_tokens - 1, 1), 3)}
        return res

    strong = args.global_batch > 0
    if strong:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} is not a multiple of the {world} ranks")
        args.batch = args.global_batch // world
    main_res = run_workload(args.batch, args.steps, args.warmup)
    # the roofline of the headline workload's dominant kernel is measured HERE, straight after its timed region and before the other legs run (the
    # batch-256 legs leave the device in another thermal / clock state: rounds 1 - 6a measured it last and the same kernel read 1 - 2 % slower there)
    main_rl = None
    if rank == 0 and not args.fp8:
        main_rl = (engine_roofline(model, args.prompt_len) or gemv_roofline(model)) if args.batch == 1 else batch_decode_gemm_roofline(model, min(args.batch, 256))
    b64_res = None
    if args.steps_b64 > 0 and args.batch == 1 and not strong:
        b64_res = run_workload(64, args.steps_b64, 1)

    c4_res = None
    if args.steps_c4 > 0 and args.batch == 1 and not strong and not args.fp8 and args.image_size == 224:
        # BASELINE configs[4], one GPU's share (B = 256 / 8): fp8 weight copies, 336-px patching (577 ViT tokens), timed in BOTH numeric modes:
        #   headline  W8A16: fp8 weights in the decode kernels (dequantised in registers), the prefill on the bf16 MFMA tiles -- the analogue of the
        #             reference's weight-only `load_in_8bit` (models/visualcla/modeling_visualcla.py:151-156) and what from_*_pretrained(load_in_8bit=True) selects;
        #   beside it W8A8: the prefill on the fp8 MFMA pipe with per-row e4m3 activations (faster, lossier).
        # Each carries `accuracy` = its logits against the bf16 path of this same model (fp8_mode_accuracy).  Runs on the same model object and is
        # undone afterwards (the position embedding returns to its native values bit for bit, the fp8 copies are dropped).
        kv = bool(args.fp8_kv)
        model.set_image_size(336)
        legs = {}
        for name, prefill in (("w8a16", False), ("w8a8", True)):
            model.enable_fp8_decode(True, prefill=prefill, kv_cache=kv)
            r = run_workload(32, args.steps_c4, 1)
            r["accuracy"] = fp8_mode_accuracy(model, prefill=prefill, kv_cache=kv)       # leaves the model in bf16 mode
            r["dtype"] = ("fp8-e4m3 weights, " + ("W8A16: fp8 weights dequantised in registers in the decode steps, bf16 x bf16 MFMA prefill" if not prefill else
                                                  "W8A8: decode as W8A16, prefill on the fp8 MFMA pipe with per-row e4m3 activations") +
                          (", e4m3 K/V cache" if kv else "") + "; bf16 activations and vision stack")
            legs[name] = r
        model.set_image_size(224)
        c4_res = dict(legs["w8a16"], mode="w8a16", w8a8=legs["w8a8"])

    strong_res = None
    if args.steps_strong > 0 and args.batch == 1 and not strong and not args.fp8 and args.image_size == 224 and not args.sample and 256 % world == 0:
        # north_star: ">= 6x images/sec scaling 1 -> 8 GPUs at batch 256": the GLOBAL batch is fixed at 256 and split over the ranks (B = 256 on
        # one GPU ... B = 32 on each of 8), so the driver's `bench.py --gpus N` series reads that curve off this leg without any flag
        strong_res = run_workload(256 // world, args.steps_strong, 1)
        strong_res["scaling"] = "strong"

    strong_c4_res = None
    if args.steps_strong_c4 > 0 and args.batch == 1 and not strong and not args.fp8 and args.image_size == 224 and not args.sample and 256 % world == 0:
        # configs[4]'s own strong-scaling leg: "fp8 weight path + 336 px, batch 256 on 8 GPUs" -- the global batch of 256 split over the ranks in the SAME
        # numeric mode `config4` headlines (W8A16: fp8 weights in the decode kernels, bf16 MFMA prefill), so an 8-GPU series has its N = 1 denominator
        model.set_image_size(336)
        model.enable_fp8_decode(True, prefill=False, kv_cache=bool(args.fp8_kv))
        strong_c4_res = run_workload(256 // world, args.steps_strong_c4, 1)
        strong_c4_res.update(scaling="strong", mode="w8a16")
        model.enable_fp8_decode(False)
        model.set_image_size(224)

    def workload_name(B, gB, image_size, fp8, prefill_fp8, kv8, sample, strong_gb=0):
        """the workload string of a leg, from the flags actually in force, and the BASELINE configs[i] it is (None: a side measurement)"""
        mode = "bf16" if not fp8 else ("fp8-e4m3 weights (" + ("W8A8: fp8 MFMA prefill, " if prefill_fp8 else "W8A16: bf16 MFMA prefill, ") + "fp8 weights dequantised in registers in the decode steps" +
                                       (", e4m3 K/V cache" if kv8 else "") + ")")
        which = None
        if not sample and args.prompt_len == 128 and args.new_tokens == 128:
            if not fp8 and image_size == 224:
                which = 1 if (B == 1 and gB == 1) else 2 if (B == 64 and gB == 64) else 3 if (B == 64 and gB == 512) else None
            elif fp8 and image_size == 336 and gB == 256:
                which = 4
        tag = f"BASELINE configs[{which}]" if which is not None else "not a BASELINE config"
        if which == 4 and world != 8:
            tag += f" (its batch of 256 on {world} GPU(s) instead of 8)"
        if strong_gb and which is None:
            tag = f"north_star's strong-scaling workload, global batch {strong_gb}" + (" = BASELINE configs[4]'s batch" if fp8 and image_size == 336 and strong_gb == 256 else "")
        return (f"VisualCLA-7B {mode}, batch={B} image(s)/GPU (global {gB}) at {image_size}px, prompt T={args.prompt_len} with 64 image tokens, "
                f"{args.new_tokens}-token {'sampled (reference default generation config, on-device sampler)' if sample else 'greedy'} decode ({tag})")

    if rank == 0:
        B = args.batch
        cfgd = {"workload": workload_name(B, main_res["global_batch"], args.image_size, args.fp8, args.fp8, bool(args.fp8_kv) and args.fp8, args.sample,
                                          args.global_batch if strong else 0) +
                            ("; second workload config2 = bf16, batch=64/GPU at 224px (BASELINE configs[2])" if b64_res else ""),
                "global_batch": main_res["global_batch"], "seq_len": args.prompt_len, "new_tokens": args.new_tokens, "image_size": args.image_size,
                "parallelism": f"dp{world}", "decode": ("hipGraph" if not args.no_graph else "eager") +
                ("; B = 1 steps: one persistent launch each (decode_engine_kernel)" if (B == 1 and not args.fp8 and os.environ.get("VCLA_ENGINE", "1") != "0") else "")}
        res = {
            "metric": f"output tokens/sec ({'sampled' if args.sample else 'greedy'}, VisualCLA-7B {args.image_size}px; images/sec alongside)",
            "value": main_res["tokens_per_sec"], "unit": "tokens/s",
            # images/sec is the throughput figure of BASELINE configs[2] (B = 64 per GPU) when that workload ran, else the main workload's
            "images_per_sec": (b64_res or main_res)["images_per_sec"],
            "images_per_sec_prefill": (b64_res or main_res)["images_per_sec_prefill"],
            "images_per_sec_workload": (f"batch={(b64_res or main_res)['batch_per_gpu']}/GPU; images_per_sec = whole requests (vision + prefill + "
                                        f"{args.new_tokens} decode steps), images_per_sec_prefill = images through ViT + Resampler + projection + prefill "
                                        "(SURVEY.md 8d), one first-token generate() per batch"),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"],
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "bf16" if not args.fp8 else ("fp8-e4m3 weights: W8A16 in the decode steps (dequantised in registers), W8A8 on the fp8 MFMA pipe in the "
                                                  "prefill (per-row e4m3 activations)" + ("; e4m3 K/V cache (unit scale)" if args.fp8_kv else "") +
                                                  "; bf16 activations elsewhere, fp32 accumulate"),
            "data": "synthetic (random-init 7B weights, N(0,1) pixels, synthetic ids)", "config": cfgd,
            "breakdown_ms": main_res.get("breakdown_ms"),
        }
        if b64_res:
            res["config2"] = dict(b64_res, workload=workload_name(64, b64_res["global_batch"], 224, False, False, False, False))
        if strong_res:
            res["strong256"] = dict(strong_res, workload=f"VisualCLA-7B bf16, GLOBAL batch 256 split over {world} GPU(s) = {256 // world} image(s)/GPU, 224 px, T=128, "
                                                          f"{args.new_tokens} greedy tokens (north_star's strong-scaling workload: compare images_per_sec / "
                                                          "images_per_sec_prefill of this leg across --gpus 1 / 2 / 4 / 8)")
        if strong_c4_res:
            res["strong256_fp8_336"] = dict(strong_c4_res, workload=workload_name(256 // world, 256, 336, True, False, bool(args.fp8_kv), False, 256) +
                                            " -- the N = 1 ... 8 series of this leg is configs[4]'s scaling curve, in the mode `config4` headlines")
        if c4_res:
            res["config4"] = dict(c4_res, workload="VisualCLA-7B, fp8 weight path, 336 px (577 ViT tokens), batch=32 image(s)/GPU = 256 / 8, T=128, "
                                                   "128 greedy tokens (BASELINE configs[4], one GPU's share); headline mode w8a16, the W8A8 prefill mode under `w8a8`")
        b1 = main_res if B == 1 else None
        b64 = b64_res if b64_res else (main_res if B == 64 else None)
        rl = []
        if not args.fp8:
            res["roofline"] = main_rl
            rl.append(main_rl)
            if B == 1 and b64:
                r = batch_decode_gemm_roofline(model, 64)
                if r:
                    rl.append(r)
            if args.image_size == 224:
                rl.append(vit_gemm_roofline(model, 64))
        else:
            r = fp8_gemm_roofline(model)
            res["roofline"] = r
            if r:
                rl.append(r)
        rl += step_rooflines(b1, b64, cfgd, args.fp8, bool(args.fp8_kv), main_res if B not in (1, 64) else strong_res)
        res["rooflines"] = rl
        if not args.no_cpu_baseline and world == 1 and args.image_size == 224:   # the CPU baseline is reported by the N=1 run only
            try:
                res["cpu_baseline"] = cpu_baseline(model, args.prompt_len, args.cpu_tokens)
            except Exception as e:  # e.g. host RAM too small for the 27 GB fp32 copy
                res["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
