#!/usr/bin/env python
"""bench.py -- VisualCLA-7B hot path on MI355X: images/sec + output tokens/sec.

One "step" = one pass of the hot path over one batch of synthetic requests:
    ViT-L/14 (224 px) -> Resampler -> projection -> splice -> LLaMA-7B prefill (T=128) -> 128-token greedy decode.
Default workload = BASELINE.json configs[1] ("VisualCLA-7B bf16, batch=1 image, 128-token greedy decode on 1 MI355X"),
per GPU; `--batch 64` selects configs[2].  Weights are random-init of the 7B architecture, inputs synthetic
(SURVEY.md section 8d); inputs are resident in HBM before the timed region.

Multi-GPU (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`): one process per GPU, full replica
each, the batch is sharded by rank (weak scaling: per-GPU batch fixed); the only collective is one RCCL all-gather of
the generated ids per step (inside the timed region).

Prints ONE JSON line on rank 0 with the throughput, a `roofline` object for the dominant kernel (the decode weight-
streaming GEMV, measured live with HIP events) and a `cpu_baseline` object (the CPU oracle timed on this host).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16 MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="images/prompts per GPU (1 = configs[1], 64 = configs[2])")
    ap.add_argument("--prompt-len", type=int, default=128)
    ap.add_argument("--new-tokens", type=int, default=128)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay in decode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp8", action="store_true", help="BASELINE configs[4] weight path: fp8 (e4m3) decode weights, bf16 activations")
    ap.add_argument("--image-size", type=int, default=224, help="336 = BASELINE configs[4] patching (577 ViT tokens, position embedding grown bicubically)")
    ap.add_argument("--sample", action="store_true", help="decode under the reference's DEFAULT_GENERATION_CONFIG (sampling + "
                    "penalties, on-device sampler) instead of greedy; a side measurement, not BASELINE's metric")
    ap.add_argument("--cpu-tokens", type=int, default=3, help="decode tokens in the bounded CPU sample")
    return ap.parse_args()


def gemv_roofline(model, n_rep: int = 20):
    """Dominant kernel of the B=1 workload: the gate/up SwiGLU GEMV with fused RMSNorm
    (gemv_kernel<bf16,bf16,M=1,EPI_SWIGLU,R=4>; ~34 % of the decode time, 43 % of the weight bytes).  One launch streams
    W_gu [2*11008, 4096] bf16 exactly once: algorithmic bytes = 180.4 MB.  The 32 layers' matrices are launched back to
    back (5.8 GB footprint, so nothing is served from the 256 MB Infinity Cache) between two HIP events on the current
    stream; achieved = bytes / mean launch duration (inter-launch gaps included -> a conservative figure)."""
    from visualcla import _lib
    t = model.config.text_config
    D, I = t["hidden_size"], t["intermediate_size"]
    P = model._packed
    dev = model.device
    x = torch.randn(1, D, device=dev).to(torch.bfloat16)
    layers = t["num_hidden_layers"]
    out = torch.empty(1, I, dtype=torch.bfloat16, device=dev)
    alg_bytes = 2 * I * D * 2

    def run():
        for l in range(layers):
            _lib.gemm(x, P[f"llama.l{l}.wgu"], 2 * I, out=out, epilogue=_lib.EPI_SWIGLU, force_kernel=2,
                      norm_gamma=P[f"llama.l{l}.ln2.g"], norm_eps=1e-6)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n_rep):
        run()
    e1.record()
    torch.cuda.synchronize()
    total_ms = e0.elapsed_time(e1)
    launches = n_rep * layers
    avg_us = total_ms * 1e3 / launches
    achieved = alg_bytes / (avg_us * 1e-6) / 1e9
    # HBM bytes per launch from the PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in their own runs, FETCH_SIZE doubled
    # per the gfx950 correction; tools/gpu_check.sh pmc -> profiles/r01_pmc_gemv1_*.txt).  Not collectable inside this run.
    traffic, src = None, None
    try:
        import re
        tot = 0.0
        for nm in ("fetch_size", "write_size"):
            m = re.search(r"-> ([0-9.]+) MB per launch", open(os.path.join(ROOT, "profiles", f"r01_pmc_gemv1_{nm}.txt")).read())
            tot += float(m.group(1)) * 1e6
        traffic, src = int(tot), "profiles/r01_pmc_gemv1_{fetch,write}_size.txt (separate rocprofv3 --pmc passes)"
    except Exception:
        pass
    return {"bound": "hbm", "kernel": "gemv1_kernel<R=2,U=4,WPB=8,SWIGLU> (gate/up GEMV + fused RMSNorm, bf16)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_source": src, "alg_bytes_per_launch": alg_bytes, "avg_launch_us": round(avg_us, 2),
            "launches_timed": launches}


def cpu_baseline(model, prompt_len: int, new_tokens: int, sample_tokens: int):
    """The CPU oracle (kind 'port': oracle/visualcla_oracle.py, fp32, torch CPU kernels on all host cores) on a bounded
    sample of the same request: 1 image through the vision stack + prefill of the T=128 prompt + `sample_tokens` decode
    steps; tokens/s is scaled to the full 128-token request as 128 / (t_vision + t_prefill + 128 * t_step)."""
    from oracle import visualcla_oracle as O   # the ONLY place this file touches oracle/: the CPU leg is the oracle, timed
    cfg_o = O.cfg_7b()
    # one NUMA domain's worth of threads: torch's CPU kernels collapse when spread over all 256 SMT threads of the
    # GPU box's 2-socket host (measured: 28 s/token at 256 threads)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    t0 = time.time()
    W = model.state_dict()                      # bf16-rounded values, fp32 on the host (~27 GB)
    t_unpack = time.time() - t0
    px, ids, mask = O.make_inputs(cfg_o, 1, prompt_len)
    with torch.no_grad():
        t0 = time.time()
        img = O.image_embeds(px, W, cfg_o)
        t_vis = time.time() - t0
        x = O.embed_and_splice(ids, img, W, cfg_o)
        cache = [None] * cfg_o.text.num_hidden_layers
        t0 = time.time()
        h = O.llama_forward(x, W, cfg_o.text, mask, cache, 0)
        logits = O.lm_head(h[:, -1:], W)[:, 0]
        t_pre = time.time() - t0
        t0 = time.time()
        past = prompt_len
        for _ in range(sample_tokens):
            nxt = logits.argmax(-1)
            e = W["text_model.model.embed_tokens.weight"][nxt][:, None, :]
            m = torch.ones(1, past + 1, dtype=torch.int64)
            h = O.llama_forward(e, W, cfg_o.text, m, cache, past)
            logits = O.lm_head(h, W)[:, 0]
            past += 1
        t_step = (time.time() - t0) / max(sample_tokens, 1)
    total = t_vis + t_pre + new_tokens * t_step
    return {"value": round(new_tokens / total, 3), "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "images_per_sec": round(1.0 / total, 4),
            "sample": (f"B=1: vision stack {t_vis:.2f}s + prefill T={prompt_len} {t_pre:.2f}s + {sample_tokens} decode steps "
                       f"at {t_step:.3f}s/token, scaled to {new_tokens} tokens; fp32 oracle, weights unpacked in {t_unpack:.1f}s")}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
        model.enable_fp8_decode()
    if args.image_size != 224:
        model.set_image_size(args.image_size)

    B = args.batch
    gB = B * world
    lo, hi = shard_range(gB, rank, world)
    px, ids, mask = make_inputs(model.config, gB, args.prompt_len)      # same global batch on every rank; take my shard
    px, ids, mask = px[lo:hi].to(dev, torch.bfloat16), ids[lo:hi].to(dev), mask[lo:hi].to(dev)
    kw = dict(input_ids=ids, pixel_values=px, attention_mask=mask, max_new_tokens=args.new_tokens, do_sample=False,
              eos_token_id=None, use_graph=not args.no_graph)
    if args.sample:   # models/visualcla/modeling_utils.py:36-47
        kw.update(do_sample=True, top_p=0.9, top_k=40, temperature=0.5, repetition_penalty=1.1, no_repeat_ngram_size=15)

    def step():
        toks = model.generate(**kw)
        return gather_tokens(toks) if world > 1 else toks

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert out.shape == (gB, args.new_tokens), out.shape

    def timed(fn, reps=2):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    breakdown = None
    if rank == 0:
        # stage split of one step (outside the timed region): vision stack, + splice/prefill/first token, + decode
        t_vis = timed(lambda: model.embed_images(px))
        kw1 = dict(kw, max_new_tokens=1)
        t_pre = timed(lambda: model.generate(**kw1))
        breakdown = {"vision_ms": round(t_vis, 2), "prefill_first_token_ms": round(t_pre - t_vis, 2),
                     "decode_ms": round(dt / args.steps * 1e3 - t_pre, 2),
                     "decode_ms_per_token_step": round((dt / args.steps * 1e3 - t_pre) / max(args.new_tokens - 1, 1), 3)}

    if rank == 0:
        tokens = gB * args.new_tokens * args.steps
        images = gB * args.steps
        res = {
            "metric": f"output tokens/sec ({'sampled' if args.sample else 'greedy'}, VisualCLA-7B {args.image_size}px; images/sec alongside)",
            "value": round(tokens / dt, 2), "unit": "tokens/s",
            "images_per_sec": round(images / dt, 4),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if not args.fp8 else "bf16 activations / fp32 accumulate, fp8-e4m3 decode weights", "data": "synthetic (random-init 7B weights, N(0,1) pixels, synthetic ids)",
            "config": {"workload": (f"VisualCLA-7B bf16, batch={B} image(s)/GPU at {args.image_size}px, prompt T={args.prompt_len} with 64 image tokens, "
                                    f"{args.new_tokens}-token {'sampled (reference default generation config, on-device sampler)' if args.sample else 'greedy'} decode "
                                    f"(BASELINE configs[{1 if B == 1 else 2}])"),
                       "global_batch": gB, "seq_len": args.prompt_len, "new_tokens": args.new_tokens,
                       "parallelism": f"dp{world}", "decode": "hipGraph" if not args.no_graph else "eager"},
        }
        res["breakdown_ms"] = breakdown
        res["roofline"] = gemv_roofline(model) if not args.fp8 else None
        if not args.no_cpu_baseline and world == 1 and args.image_size == 224:   # the CPU baseline is reported by the N=1 run only
            try:
                res["cpu_baseline"] = cpu_baseline(model, args.prompt_len, args.new_tokens, args.cpu_tokens)
            except Exception as e:  # e.g. host RAM too small for the 27 GB fp32 copy
                res["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
