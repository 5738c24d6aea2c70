"""CPU-side checks: the C-ABI library loads and exports every declared symbol, weight packing round-trips,
prompt assembly follows the reference template, and the data-parallel shard/gather logic works on gloo (world 2)."""
import json
import os
import re
import subprocess
import sys
from types import SimpleNamespace

import pytest
import torch

from oracle import visualcla_oracle as O
from tests.helpers import stub_tokenizer, to_vcla_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from visualcla import _lib
    lib = _lib.load()                     # raises if the .so is missing or lacks a symbol bound in _lib.SYMBOLS
    header = open(os.path.join(ROOT, "include", "visualcla_hip.h")).read()
    declared = set(re.findall(r"\b(vcla_[a-z0-9_]+)\s*\(", header))
    declared -= {"vcla_gemm_args", "vcla_attn_args", "vcla_model_cfg", "vcla_ctx"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.vcla_version() == 5


def test_no_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import visualcla
    with pytest.raises(Exception) as e:
        visualcla.VisualCLAModel(to_vcla_config(O.cfg_tiny()))
    assert "no CPU fallback" in str(e.value)


def test_argument_validation_without_gpu():
    """status-code / ValueError convention is reachable without launching anything"""
    from visualcla import _lib
    a = torch.zeros(4, 100, dtype=torch.bfloat16)
    w = torch.zeros(128, 100, dtype=torch.bfloat16)
    with pytest.raises(ValueError, match="BAD_SHAPE"):
        _lib_gemm_cpu(_lib, a, w, 128)


def _lib_gemm_cpu(_lib, a, w, n):
    import ctypes as C
    out = torch.empty(a.shape[0], n, dtype=torch.bfloat16)
    args = _lib.GemmArgs()
    args.A, args.lda, args.W, args.C, args.ldc = a.data_ptr(), a.stride(0), w.data_ptr(), out.data_ptr(), out.stride(0)
    args.M, args.N, args.K = a.shape[0], n, a.shape[1]
    _lib.check(_lib.load().vcla_gemm(C.byref(args), 1, None))


@pytest.mark.parametrize("mk", [O.cfg_tiny, O.cfg_small])
def test_pack_unpack_roundtrip(mk):
    from visualcla.weights import pack_state_dict, unpack_state_dict, pad_to
    cfg = mk()
    W = O.make_weights(cfg, seed=0)
    vc = to_vcla_config(cfg)
    packed = pack_state_dict(W, vc, "cpu", torch.bfloat16)
    t = cfg.text
    assert packed["llama.l0.wgu"].shape == (pad_to(2 * t.intermediate_size, 128), t.hidden_size)
    assert packed["vit.patch_w"].shape[1] % 64 == 0
    # interleave: packed rows 0..15 = gate rows 0..15, rows 16..31 = up rows 0..15
    assert torch.equal(packed["llama.l0.wgu"][:16].float(), W["text_model.model.layers.0.mlp.gate_proj.weight"][:16])
    assert torch.equal(packed["llama.l0.wgu"][16:32].float(), W["text_model.model.layers.0.mlp.up_proj.weight"][:16])
    back = unpack_state_dict(packed, vc)
    assert set(back) == set(W) - {"visual_resampler.pooler.dense.weight", "visual_resampler.pooler.dense.bias"}
    for k, v in back.items():
        assert torch.equal(v.reshape(W[k].shape), W[k]), k
    # transformers-5 flat CLIP key layout is accepted too
    flat = {k.replace("vision_model.vision_model.", "vision_model."): v for k, v in W.items()}
    p2 = pack_state_dict(flat, vc, "cpu", torch.bfloat16)
    assert torch.equal(p2["vit.l0.wqkv"], packed["vit.l0.wqkv"])


def test_fragment_major_layout():
    """element (n, k) of W sits at [n // 16][k // 32][((k % 32) // 8) * 16 + n % 16][k % 8]"""
    from visualcla.weights import to_fragment_major, from_fragment_major
    w = torch.arange(32 * 64, dtype=torch.float32).view(32, 64).to(torch.bfloat16)
    f = to_fragment_major(w)
    assert f.shape == (2, 2, 64, 8)
    for n, k in ((0, 0), (5, 9), (17, 40), (31, 63), (16, 31)):
        assert f[n // 16, k // 32, ((k % 32) // 8) * 16 + n % 16, k % 8] == w[n, k]
    assert torch.equal(from_fragment_major(f), w)


def test_rope_tables_match_oracle():
    from visualcla.weights import rope_tables
    cos, sin = rope_tables(64, 128, 10000.0)
    c, s = O.llama_rope_tables(torch.arange(64), 128, 10000.0, torch.float32)
    assert torch.equal(cos, c[:, :64]) and torch.equal(sin, s[:, :64])


class _CharTok:
    """tiny deterministic tokenizer: one id per character, special strings map to single ids"""
    bos_token, img_start_token, img_end_token, img_token = "<s>", "<img>", "</img>", "<img_token>"

    def __call__(self, text, return_tensors=None, add_special_tokens=False):
        ids, i = [], 0
        spec = {"<s>": 1, "<img>": 300, "</img>": 301, "<img_token>": 303}
        while i < len(text):
            for k, v in spec.items():
                if text.startswith(k, i):
                    ids.append(v); i += len(k); break
            else:
                ids.append(3 + (ord(text[i]) % 250)); i += 1
        return SimpleNamespace(input_ids=torch.tensor([ids]), attention_mask=torch.ones(1, len(ids), dtype=torch.int64))


def test_prompt_template_and_history():
    from visualcla.modeling_utils import encoding_text
    tok = _CharTok()
    enc = encoding_text([], "what is this?", 4, tok)
    ids = enc.input_ids[0].tolist()
    assert ids[0] == 1 and ids.count(300) == 1 and ids.count(303) == 4 and ids.count(301) == 1
    p0 = ids.index(300)
    assert ids[p0 + 1:p0 + 5] == [303] * 4 and ids[p0 + 5] == 301
    hist = [{"type": "instruction", "value": "a", "first_instruction": True}, {"type": "response", "value": "b"}]
    enc2 = encoding_text(hist, "c", 4, tok)
    assert enc2.input_ids[0].tolist().count(300) == 1          # image slot only in the first instruction
    with pytest.raises(ValueError):
        encoding_text([{"type": "bogus", "value": "x"}], "c", 4, tok)


def test_shard_range_covers_everything():
    from visualcla.distributed import shard_range
    for n in (0, 1, 7, 8, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


_GLOO_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "visual-chinese-llama-alpaca_amd"))
from visualcla.distributed import gather_tokens, shard_range
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n = 5
full = torch.arange(n * 3).view(n, 3)
lo, hi = shard_range(n, rank, world)
mine = full[lo:hi, : 3 - rank]            # ragged in both dims (rank 1 stopped one token early)
got = gather_tokens(mine, pad_id=-1)
want = full.clone()
lo1, hi1 = shard_range(n, 1, world)
want[lo1:hi1, 2] = -1
assert torch.equal(got, want), (rank, got, want)
# the serving path: shapes known up front (n_total requests, max_new_tokens columns) -> ONE collective, no size exchange
calls = []
orig = dist.all_gather_into_tensor
dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
got = gather_tokens(mine, n_total=n, n_cols=3, pad_id=-1)
assert torch.equal(got, want) and len(calls) == 1, (rank, got, want, calls)
dist.barrier()
print("ok", rank)
"""


def test_data_parallel_gather_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29617", str(script), ROOT],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_bench_self_spawns_ranks_and_gathers_once():
    """`python bench.py --gpus 2` with no launcher around it must start 2 ranks itself and report n_gpus = 2: the driver's
    scaling command.  --plumbing-check runs exactly that launch / shard / all-gather / max-over-ranks path on CPU (gloo)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--plumbing-check"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout
    res = json.loads(line[0])
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["config"]["parallelism"] == "dp2" and res["config"]["global_batch"] == 6
    # the default line carries north_star's strong-scaling workload (global batch 256 split over the ranks) and both images/sec definitions
    assert res["images_per_sec"] > 0 and res["images_per_sec_prefill"] > 0
    s256 = res["strong256"]
    assert s256["global_batch"] == 256 and s256["batch_per_gpu"] == 128 and s256["scaling"] == "strong"
    assert s256["images_per_sec"] > 0 and s256["images_per_sec_prefill"] > 0 and s256["steps"] >= 2       # (not a single sample)
    # ... and configs[4]'s own strong-scaling leg (fp8 W8A16, 336 px, the same global batch of 256): its N = 1 run is the denominator of the 8-GPU claim
    c4s = res["strong256_fp8_336"]
    assert c4s["global_batch"] == 256 and c4s["batch_per_gpu"] == 128 and c4s["scaling"] == "strong" and c4s["mode"] == "w8a16"


def test_synthetic_shards_are_slices_of_the_global_batch():
    """a rank builds only its own requests; they must be the same tensors a single process would slice out of the full batch"""
    import visualcla
    from visualcla.synthetic import make_inputs
    cfg = visualcla.visualcla_7b_config()
    cfg.vision_config = dict(cfg.vision_config, image_size=56)
    px, ids, mask = make_inputs(cfg, 5, 128)
    px2, ids2, mask2 = make_inputs(cfg, 2, 128, first_request=3)
    assert torch.equal(px[3:5], px2) and torch.equal(ids[3:5], ids2) and torch.equal(mask[3:5], mask2)


def test_python_constants_match_the_header():
    from visualcla import _lib
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "visualcla_hip.h")).read()
    for name, val in (("VCLA_SAMPLE_MAX_TOP_K", _lib.SAMPLE_MAX_TOP_K), ("VCLA_SAMPLE_MAX_EOS", _lib.SAMPLE_MAX_EOS),
                      ("VCLA_SAMPLE_KEPT_LD", _lib.SAMPLE_KEPT_LD), ("VCLA_SAMPLE_MAX_VOCAB", _lib.SAMPLE_MAX_VOCAB)):
        m = re.search(rf"#define {name} (\d+)", hdr)
        assert m and int(m.group(1)) == val, name


def test_fold_lora_formula_modules_to_save_and_errors():
    from visualcla.weights import fold_lora
    g = torch.Generator().manual_seed(0)
    base = {"text_model.model.layers.0.self_attn.q_proj.weight": torch.randn(8, 8, generator=g),
            "text_model.model.layers.0.self_attn.v_proj.weight": torch.randn(8, 8, generator=g).half(),
            "text_model.model.embed_tokens.weight": torch.randn(10, 8, generator=g),
            "vision_model.encoder.layers.0.self_attn.q_proj.weight": torch.randn(8, 8, generator=g),      # flat (transformers 5.x)
            "image_projection_layer.weight": torch.zeros(8, 8)}
    want = {k: v.clone().float() for k, v in base.items()}
    r, alpha = 2, 8
    A = {k: torch.randn(r, 8, generator=g) for k in "qvc"}
    Bm = {k: torch.randn(8, r, generator=g) for k in "qvc"}
    new_embed, new_proj = torch.randn(14, 8, generator=g), torch.randn(8, 8, generator=g)
    adapter = {
        "base_model.model.text_model.model.layers.0.self_attn.q_proj.lora_A.weight": A["q"],
        "base_model.model.text_model.model.layers.0.self_attn.q_proj.lora_B.weight": Bm["q"],
        "base_model.model.text_model.model.layers.0.self_attn.v_proj.lora_A.default.weight": A["v"],       # newer peft naming
        "base_model.model.text_model.model.layers.0.self_attn.v_proj.lora_B.default.weight": Bm["v"],
        "base_model.model.vision_model.vision_model.encoder.layers.0.self_attn.q_proj.lora_A.weight": A["c"],   # 4.x nesting
        "base_model.model.vision_model.vision_model.encoder.layers.0.self_attn.q_proj.lora_B.weight": Bm["c"],
        "base_model.model.text_model.model.embed_tokens.weight": new_embed,                                  # modules_to_save
        "base_model.model.image_projection_layer.modules_to_save.default.weight": new_proj,
    }
    out = fold_lora(base, adapter, {"r": r, "lora_alpha": alpha})
    s = alpha / r
    torch.testing.assert_close(out["text_model.model.layers.0.self_attn.q_proj.weight"], want["text_model.model.layers.0.self_attn.q_proj.weight"] + s * Bm["q"] @ A["q"])
    torch.testing.assert_close(out["text_model.model.layers.0.self_attn.v_proj.weight"], want["text_model.model.layers.0.self_attn.v_proj.weight"] + s * Bm["v"] @ A["v"])
    torch.testing.assert_close(out["vision_model.encoder.layers.0.self_attn.q_proj.weight"], want["vision_model.encoder.layers.0.self_attn.q_proj.weight"] + s * Bm["c"] @ A["c"])
    assert torch.equal(out["text_model.model.embed_tokens.weight"], new_embed) and torch.equal(out["image_projection_layer.weight"], new_proj)
    # fan_in_fan_out stores the delta transposed
    b2 = {"w.weight": torch.zeros(8, 8)}
    fold_lora(b2, {"w.lora_A.weight": A["q"], "w.lora_B.weight": Bm["q"]}, {"r": r, "lora_alpha": alpha, "fan_in_fan_out": True})
    torch.testing.assert_close(b2["w.weight"], (s * Bm["q"] @ A["q"]).t())
    with pytest.raises(KeyError):
        fold_lora({"w.weight": torch.zeros(8, 8)}, {"nope.lora_A.weight": A["q"], "nope.lora_B.weight": Bm["q"]}, {"r": r, "lora_alpha": alpha})
    with pytest.raises(KeyError):
        fold_lora({"w.weight": torch.zeros(8, 8)}, {"w.lora_A.weight": A["q"]}, {"r": r, "lora_alpha": alpha})
    with pytest.raises(ValueError):
        fold_lora({"w.weight": torch.zeros(8, 8)}, {"w.lora_A.weight": A["q"], "w.lora_B.weight": Bm["q"]}, {"r": 4, "lora_alpha": alpha})


def test_tgwebui_pipeline_statics_and_missing_settings():
    from visualcla import tgwebui as T
    P = T.VisualCLA_7B_Pipeline
    assert (P.name(), P.placeholder_token_id(), P.visualcla_projector_shape(), P.num_image_embeds()) == ("visualcla-7b", 49957, (1024, 4096), 64)
    assert (P.image_start(), P.image_end(), P.image_placeholder()) == ("<img>", "</img>", "<img_token>")
    assert T.available_pipelines == ["visualcla-7b"] and T.get_pipeline("llava-7b", {}) is None
    assert T.get_pipeline_from_model_name("llama-13b", {}) is None
    with pytest.raises(KeyError):
        T.get_pipeline("visualcla-7b", {})


def test_benchmark_input_generator_matches_the_oracle_recipe():
    """bench.py draws its requests from visualcla.synthetic (product side); same seeds -> same tensors as the oracle's generator"""
    import visualcla
    from visualcla.synthetic import make_inputs, stub_tokenizer
    cfg_o = O.cfg_7b()
    want = O.make_inputs(cfg_o, 3, 128)
    got = make_inputs(visualcla.visualcla_7b_config(), 3, 128)
    for a, b in zip(got, want):
        assert a.dtype == b.dtype and torch.equal(a, b)
    tok = stub_tokenizer()
    assert (tok.img_start_token_id, tok.img_end_token_id, tok.img_token_id) == (cfg_o.img_start_token_id, cfg_o.img_end_token_id, cfg_o.img_token_id)
    small = to_vcla_config(O.cfg_tiny())
    c = O.cfg_tiny()
    got = make_inputs(small, 2, 24, img_ids=(c.img_start_token_id, c.img_end_token_id, c.img_token_id))
    for a, b in zip(got, O.make_inputs(c, 2, 24)):
        assert torch.equal(a, b)
    px = make_inputs(visualcla.visualcla_7b_config(), 1, 128, image_size=336)[0]
    assert px.shape == (1, 3, 336, 336)


def test_header_is_plain_c_and_a_c_program_links_the_library(tmp_path):
    """the drop-in boundary is a C ABI: compile a C99 translation unit against include/visualcla_hip.h with warnings as errors,
    link it to libvisualcla_hip.so and run the entry points that validate arguments before touching a device"""
    root = os.path.join(os.path.dirname(__file__), "..")
    from visualcla import _lib
    exe = str(tmp_path / "c_abi_check")
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c_abi_check.c"),
           "-o", exe, "-L", libdir, "-l:libvisualcla_hip.so", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "abi" in r.stdout and "last error:" in r.stdout


def test_no_compute_kernel_uses_scratch():
    """Every kernel of the built library except the sampler must have an EMPTY private segment (no register spills, no local arrays in
    scratch).  Round 3 measured what a spill costs here: a 256 x 256 GEMM instance that spilled 20 registers gave FLAKY results (2 failures
    in 3 passes of 481 parity tests; 4 x 481 green once the spills were gone), and a streaming-GEMM instance with scratch slowed every K
    loop of its launch.  Reads the code objects out of the .so (llvm-objcopy / clang-offload-bundler / llvm-readelf from the ROCm LLVM)."""
    import re
    import shutil
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    so = os.path.join(ROOT, "visual-chinese-llama-alpaca_amd", "visualcla", "libvisualcla_hip.so")
    if not (os.path.exists(so) and all(os.path.exists(os.path.join(llvm, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"))):
        pytest.skip("library not built or ROCm LLVM tools absent")
    with tempfile.TemporaryDirectory() as td:
        fb = os.path.join(td, "fat.bin")
        subprocess.check_call([f"{llvm}/llvm-objcopy", "--dump-section", f".hip_fatbin={fb}", so])
        data = open(fb, "rb").read()
        offs = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), data)]
        assert offs, "no offload bundle in .hip_fatbin"
        n_kernels, bad = 0, []
        for k, o in enumerate(offs):
            bf, co = os.path.join(td, f"b{k}.bin"), os.path.join(td, f"b{k}.co")
            with open(bf, "wb") as f:
                f.write(data[o:offs[k + 1] if k + 1 < len(offs) else len(data)])
            subprocess.check_call([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={bf}",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
            notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
            for blk in notes.split("- .agpr_count")[1:]:
                name, priv = re.search(r"\.name:\s+(\S+)", blk), re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
                assert name and priv
                n_kernels += 1
                if int(priv.group(1)) > 0 and "sample_kernel" not in name.group(1):     # the sampler keeps its top-k candidates in a local array
                    bad.append((name.group(1), int(priv.group(1))))
    assert n_kernels > 300, n_kernels
    assert not bad, f"kernels with a private segment (spills / scratch arrays): {bad}"


@pytest.mark.parametrize("sym,vm", [("_Z19attn_vit_dma_kernel14vcla_attn_args", 12), ("_Z20attn_vit_long_kernelILi9EEv14vcla_attn_args", 17)])
def test_vit_attention_asm_register_loads_are_untouched_until_their_wait(sym, vm):
    """attn_vit_dma_kernel (257 tokens) and attn_vit_long_kernel (577 tokens) load their Q fragments with inline-asm `global_load_dwordx4` that hipcc
    does not count (the hand-counted vmcnt in front of key tile 0 completes them together with the LDS-DMA pieces).  hipcc treats an asm load's
    destination as written at the end of the statement, so nothing but register allocation keeps it from copying / spilling / reusing those
    registers before the data lands (cdna_hip_programming.md section 5).  This test audits the SHIPPED code object: between the 8 loads and the
    `s_waitcnt vmcnt(12 / 17)` no instruction may name one of their 32 destination registers."""
    import re
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    so = os.path.join(ROOT, "visual-chinese-llama-alpaca_amd", "visualcla", "libvisualcla_hip.so")
    if not (os.path.exists(so) and all(os.path.exists(os.path.join(llvm, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump", "llvm-readelf"))):
        pytest.skip("library not built or ROCm LLVM tools absent")
    with tempfile.TemporaryDirectory() as td:
        fb = os.path.join(td, "fat.bin")
        subprocess.check_call([f"{llvm}/llvm-objcopy", "--dump-section", f".hip_fatbin={fb}", so])
        data = open(fb, "rb").read()
        offs = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), data)]
        body = None
        for k, o in enumerate(offs):
            bf, co = os.path.join(td, f"b{k}.bin"), os.path.join(td, f"b{k}.co")
            with open(bf, "wb") as f:
                f.write(data[o:offs[k + 1] if k + 1 < len(offs) else len(data)])
            subprocess.check_call([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={bf}",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
            if sym not in subprocess.run([f"{llvm}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout:
                continue
            dis = subprocess.run([f"{llvm}/llvm-objdump", "-d", "--mcpu=gfx950", f"--disassemble-symbols={sym}", co], capture_output=True, text=True, check=True).stdout
            body = dis[dis.index(f"<{sym}>:"):]
            break
    assert body is not None, f"{sym} not found in the library"
    lines = [ln.split("//")[0] for ln in body.splitlines()[1:]]
    loads, first, wait = [], None, None
    for i, ln in enumerate(lines):
        m = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off\s*$", ln.strip())
        if m and wait is None:
            loads.append((int(m.group(1)), int(m.group(2))))
            first = i if first is None else first
        if wait is None and re.search(rf"s_waitcnt vmcnt\({vm}\)", ln):
            wait = i
    assert len(loads) == 8 and wait is not None and first < wait, (loads, first, wait)
    regs = {r for a, b in loads for r in range(a, b + 1)}
    assert len(regs) == 32
    touched = []
    for ln in lines[first:wait]:
        t = ln.strip()
        if re.search(r"global_load_dwordx4 v\[\d+:\d+\], v\[\d+:\d+\], off\s*$", t):
            continue                                     # the loads themselves (their ADDRESS registers are checked by the other loads' lines below)
        used = set()
        for m in re.finditer(r"\bv(\d+)\b", t):
            used.add(int(m.group(1)))
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]", t):
            used.update(range(int(m.group(1)), int(m.group(2)) + 1))
        if used & regs:
            touched.append(t)
    assert not touched, "instructions naming a Q destination register before the counted wait:\n" + "\n".join(touched[:10])


def test_hf_cpu_baseline_child_runs_and_reports_the_contract_fields():
    """bench.py's cpu_baseline leg = oracle/hf_cpu_baseline.py in a child process (transformers' CLIPVisionModel + LlamaForCausalLM.generate with
    the restated resampler between them).  Run it at the small geometry: one JSON line with value / unit / cores / kind / sample."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "hf_cpu_baseline.py"), "--geometry", "small", "--prompt-len", "48", "--tokens", "4",
                        "--sweep", "1,2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["kind"] == "hf+port-resampler" and d["unit"] == "tokens/s" and d["value"] > 0 and d["cores"] in (1, 2)
    assert set(d["thread_sweep_decode_s_per_token"]) == {"1", "2"} and "LlamaForCausalLM.generate" in d["sample"]


def test_check_request_validates_in_one_pass_on_cpu():
    """VisualCLAModel._check_request (every data-dependent validation of a request, one host synchronisation) is plain tensor logic: exercised on
    the CPU through the oracle-backed stand-in of tests/repl_stub (no HIP involved).  Precedence and messages follow the reference: vocabulary
    first, then the image slot (modeling_visualcla.py:300-302 / :366-367), then the mask."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "repl_stub"))
    from oracle_backed import OracleBackedModel
    cfg = O.cfg_tiny()
    m = OracleBackedModel(to_vcla_config(cfg), {})
    m.tokenizer = stub_tokenizer(cfg)
    m.image_at_head = False
    Q = cfg.resampler.num_query_tokens
    px, ids, mask = O.make_inputs(cfg, 2, 24)
    pos, am = m._check_request(ids, mask, Q, for_generate=False)
    p0 = int((ids[0] == cfg.img_start_token_id).nonzero()[0])
    assert pos.dtype == torch.int32 and pos.tolist() == [p0, p0] and am is None            # all-ones mask -> no key mask at all
    lp = mask.clone(); lp[1, :3] = 0
    pos, am = m._check_request(ids, lp, Q, for_generate=True)
    assert am is not None and torch.equal(am, lp)
    # a row without <img>: -1 (its embeds pass through); forward additionally asks for an <img_token>
    no_slot = ids.clone(); no_slot[1] = torch.randint(3, 300, (24,))
    assert m._check_request(no_slot, mask, Q, for_generate=True)[0].tolist() == [p0, -1]
    only_start = no_slot.clone(); only_start[1, 2] = cfg.img_start_token_id               # <img> without the slot: generate must refuse, forward lets it pass
    assert m._check_request(only_start, mask, Q, for_generate=False)[0].tolist() == [p0, -1]
    with pytest.raises(ValueError, match="Num of patch"):
        m._check_request(only_start, mask, Q, for_generate=True)
    bad = ids.clone(); bad[0, p0 + Q + 1] = 5
    with pytest.raises(ValueError, match="Num of patch"):
        m._check_request(bad, mask, Q, for_generate=False)
    oov = bad.clone(); oov[1, 0] = cfg.text.vocab_size                                     # vocabulary error wins over the slot error
    with pytest.raises(ValueError, match="outside the vocabulary"):
        m._check_request(oov, mask, Q, for_generate=False)
    hole = mask.clone(); hole[0, 5] = 0
    with pytest.raises(ValueError, match="between visible tokens"):                        # generate: HF's mask-derived positions would differ -> refused
        m._check_request(ids, hole, Q, for_generate=True)
    assert torch.equal(m._check_request(ids, hole, Q, for_generate=False)[1], hole)        # forward: arange positions, the mask only removes keys (as the reference)
    lab = ids.clone(); lab[:, :5] = -100
    m._check_request(ids, mask, Q, for_generate=False, labels=lab)
    lab[1, 7] = cfg.text.vocab_size
    with pytest.raises(ValueError, match="labels contain ids outside"):
        m._check_request(ids, mask, Q, for_generate=False, labels=lab)
    right = mask.clone(); right[1, 20:] = 0
    assert m._check_request(ids, right, Q, for_generate=False)[1] is not None              # right padding: accepted
    # image_at_head: the image columns are prepended to the mask (modeling_visualcla.py:308-310); a left-padded text mask then has an interior hole
    m.image_at_head = True
    txt = torch.randint(3, 300, (2, 10)); tm = torch.ones_like(txt)
    pos, am = m._check_request(txt, tm, Q, for_generate=False)
    assert pos is None and am is None
    tm[1, :2] = 0
    with pytest.raises(ValueError, match="between visible tokens"):
        m._check_request(txt, tm, Q, for_generate=True)
    am = m._check_request(txt, tm, Q, for_generate=False)[1]
    assert am.shape == (2, Q + 10) and bool(am[:, :Q].all()) and am[1, Q:Q + 2].tolist() == [0, 0]
    assert m._check_request(txt, tm, 0, for_generate=False)[1] is not None                 # text-only (no image): plain left padding is fine


def test_beam_search_bookkeeping_equals_transformers_on_random_configs():
    """visualcla.beam_search (host bookkeeping of generate(num_beams > 1)) against transformers' own beam search -- what the reference's generate() forwards
    to (models/visualcla/modeling_visualcla.py:382-391) -- on a tiny random LLaMA driven by `inputs_embeds`: 100 random draws over batch, left padding, beams,
    length limits, eos sets the model really produces (one, two, five ids), pad ids (unset / 0 / another), length penalties (incl. 0 and negative),
    early_stopping in {False, True, "never"}, several returned hypotheses, and the config-selected processors built by visualcla.logits_processors
    (repetition penalty, no-repeat-ngram, eos floor, bad words, suppressed tokens, forced eos, a prefix_allowed_tokens_fn that depends on the prompt
    index).  Ids must be EQUAL.  The step
    function here re-runs the full forward of the re-ordered sequences (no cache): the cache gather is covered by the golden-fixture tests."""
    import random
    import warnings
    sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd"))
    from transformers import GenerationConfig, LlamaConfig, LlamaForCausalLM
    from visualcla.beam_search import beam_search
    from visualcla.logits_processors import build_logits_processors
    torch.manual_seed(0)
    V, H = 23, 32
    cfg = LlamaConfig(vocab_size=V, hidden_size=H, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                      max_position_embeddings=64, pad_token_id=None, bos_token_id=None, eos_token_id=None)
    m = LlamaForCausalLM(cfg).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(6.0)                                    # sharper distributions: the beams diverge and the eos ids get produced

    def ours(emb, mask, nb, L, eos, pad, lp, es, nrs, procs):
        B = emb.shape[0]
        x, mk = emb.repeat_interleave(nb, 0), mask.repeat_interleave(nb, 0)
        state = {"seq": torch.zeros(B * nb, 0, dtype=torch.long)}

        def fwd():
            s = state["seq"]
            e = torch.cat([x, m.get_input_embeddings()(s)], 1)
            am = torch.cat([mk, torch.ones(B * nb, s.shape[1], dtype=mk.dtype)], 1)
            return m(inputs_embeds=e, attention_mask=am, position_ids=(am.cumsum(-1) - 1).clamp(min=0)).logits[:, -1].float()

        def step(tok, rows):
            state["seq"] = torch.cat([state["seq"].index_select(0, rows), tok[:, None]], 1)
            return fwd()
        return beam_search(fwd(), step, B, nb, L, eos_ids=eos, pad_token_id=pad, length_penalty=lp, early_stopping=es, num_return_sequences=nrs, processors=procs)

    rng = random.Random(1)
    hit_eos = finished_early = 0
    for case in range(100):
        B, T, nb, L = rng.choice([1, 2, 3]), rng.choice([3, 5, 8]), rng.choice([2, 3, 4, 5]), rng.choice([1, 2, 4, 7, 10])
        eos = rng.choice([(), (3,), (3, 7), (1, 2, 3, 4, 5)])
        pad = rng.choice([None, 0, 9])
        lp, es = rng.choice([1.0, 0.0, 0.6, 2.0, -1.0]), rng.choice([False, True, "never"])
        nrs = rng.choice([1, 1, nb, max(1, nb - 1)])
        rp, ng = rng.choice([None, None, 1.3]), rng.choice([None, None, 2])
        emb = torch.randn(B, T, H, generator=torch.Generator().manual_seed(case))
        mask = torch.ones(B, T, dtype=torch.long)
        for b in range(B):
            mask[b, :rng.choice([0, 0, 1, 2])] = 0
        kw = dict(num_beams=nb, max_new_tokens=L, do_sample=False, length_penalty=lp, early_stopping=es, num_return_sequences=nrs,
                  eos_token_id=list(eos) if eos else None, pad_token_id=pad)
        if rp:
            kw["repetition_penalty"] = rp
        if ng:
            kw["no_repeat_ngram_size"] = ng
        # the processors come from the package's own config -> processor mapping, with the fields a beam request may carry
        fn = (lambda b, sent: [t for t in range(V) if (t + b) % 5 != 0]) if rng.random() < 0.3 else None      # depends on the PROMPT index: checks the beam -> prompt mapping
        if eos and rng.random() < 0.3:
            kw["min_new_tokens"] = 2
        if rng.random() < 0.3:
            kw["bad_words_ids"] = [[4], [6, 8]]
        if rng.random() < 0.2:
            kw["suppress_tokens"] = [10]
        if eos and rng.random() < 0.2:
            kw["forced_eos_token_id"] = eos[0]
        procs = build_logits_processors(GenerationConfig(**kw), list(eos), "cpu", prompt_len=T, n_new=L, prefix_allowed_tokens_fn=fn)
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = m.generate(inputs_embeds=emb, attention_mask=mask, prefix_allowed_tokens_fn=fn, **kw)
            got = ours(emb, mask, nb, L, eos, pad, lp, es, nrs, procs)
        assert ref.shape == got.shape and torch.equal(ref, got), (case, kw, ref, got)
        hit_eos += int(bool(eos) and bool(torch.isin(ref, torch.tensor(list(eos) or [-5])).any()))
        finished_early += int(ref.shape[1] < L)
    assert hit_eos >= 10 and finished_early >= 4             # the draw really exercises finished hypotheses


def test_logits_processors_equal_transformers_on_random_configs():
    """visualcla.logits_processors (GenerationConfig -> processor list + token budget of one generate() call) against transformers' generate driven by
    `inputs_embeds` -- what the reference's generate() forwards its config to (models/visualcla/modeling_visualcla.py:382-391).  A tiny random LLaMA runs HF's
    generate with `output_scores`; the sequences it produced are replayed step by step through THIS package's processor list and the processed scores must be
    the ones HF reported (same -inf pattern, values to 1e-4), for 70 random draws over: max_new_tokens / max_length (counts the prompt) / neither, eos sets,
    min_new_tokens and min_length (less the prompt), repetition penalty, no-repeat-ngram, bad words, sequence bias, suppressed tokens (always / at the start),
    forced bos / eos, exponential length decay, inf / nan removal, renormalisation, a caller's processor, prefix_allowed_tokens_fn, and when sampling:
    temperature, top-k, top-p, min-p, typical, epsilon and eta cut-offs."""
    import copy
    import random
    import warnings
    sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd"))
    from transformers import GenerationConfig, LlamaConfig, LlamaForCausalLM, LogitsProcessor, LogitsProcessorList
    from visualcla.logits_processors import build_logits_processors, new_token_budget
    torch.manual_seed(0)
    V, H = 29, 32
    m = LlamaForCausalLM(LlamaConfig(vocab_size=V, hidden_size=H, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                                     max_position_embeddings=64, pad_token_id=None, bos_token_id=None, eos_token_id=None)).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(5.0)

    class Bump(LogitsProcessor):                              # a caller's own processor
        def __call__(self, ids, scores):
            s = scores.clone()
            s[:, 11] += 0.5
            return s

    rng = random.Random(3)
    checked = 0
    for case in range(70):
        B, T = rng.choice([1, 2, 3]), rng.choice([3, 6])
        kw = {}
        if rng.random() < 0.7:
            kw["max_new_tokens"] = rng.choice([1, 3, 6, 9])
        elif rng.random() < 0.7:
            kw["max_length"] = T + rng.choice([1, 4, 8])
        eos = rng.choice([None, 3, [3, 7]])
        kw.update(eos_token_id=eos, pad_token_id=rng.choice([None, 0]))
        for name, prob, val in (("min_new_tokens", 0.3, rng.choice([1, 2, 4])), ("min_length", 0.3, rng.choice([T + 2, 2, T + 5])), ("repetition_penalty", 0.3, 1.2),
                                ("no_repeat_ngram_size", 0.3, 2), ("bad_words_ids", 0.3, [[5], [6, 8]]), ("sequence_bias", 0.2, [[[4], 2.0], [[9, 10], -3.0]]),
                                ("suppress_tokens", 0.2, [12, 13]), ("begin_suppress_tokens", 0.2, [14, 1]), ("forced_bos_token_id", 0.15, 2),
                                ("remove_invalid_values", 0.2, True), ("renormalize_logits", 0.3, True), ("encoder_repetition_penalty", 0.2, 1.3)):
            if rng.random() < prob:
                kw[name] = val
        if eos is not None and rng.random() < 0.2:
            kw["forced_eos_token_id"] = 3
        if eos is not None and rng.random() < 0.2:
            kw["exponential_decay_length_penalty"] = (1, 1.5)
        kw["do_sample"] = rng.random() < 0.5
        if kw["do_sample"]:
            for name, prob, val in (("temperature", 0.5, rng.choice([0.5, 1.7])), ("top_k", 0.5, rng.choice([3, 10])), ("top_p", 0.5, 0.8), ("min_p", 0.3, 0.05),
                                    ("typical_p", 0.3, 0.7), ("epsilon_cutoff", 0.3, 0.02), ("eta_cutoff", 0.3, 0.03)):
                if rng.random() < prob:
                    kw[name] = val
        extra = [Bump()] if rng.random() < 0.3 else []
        fn = (lambda b, sent: list(range(1, V - 2))) if rng.random() < 0.25 else None
        emb = torch.randn(B, T, H, generator=torch.Generator().manual_seed(case))
        mask = torch.ones(B, T, dtype=torch.long)
        gc = GenerationConfig(**kw)
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.manual_seed(case)
            try:
                ref = m.generate(inputs_embeds=emb, attention_mask=mask, generation_config=copy.deepcopy(gc), logits_processor=LogitsProcessorList(extra),
                                 prefix_allowed_tokens_fn=fn, return_dict_in_generate=True, output_scores=True)
            except RuntimeError:                              # a draw whose filters leave torch.multinomial nothing to sample from upstream
                continue
            seq, scores = ref.sequences, ref.scores
            n_new = new_token_budget(gc, T)
            procs = build_logits_processors(gc, [] if eos is None else ([eos] if isinstance(eos, int) else list(eos)), "cpu", prompt_len=T, n_new=n_new,
                                            extra=extra, prefix_allowed_tokens_fn=fn)
            assert len(scores) == seq.shape[1] <= n_new, (case, kw)
            assert seq.shape[1] == n_new or eos is not None, (case, kw)
            for i in range(seq.shape[1]):
                e = torch.cat([emb, m.get_input_embeddings()(seq[:, :i])], 1)
                s = m(inputs_embeds=e, attention_mask=torch.cat([mask, torch.ones(B, i, dtype=torch.long)], 1)).logits[:, -1].float()
                for p in procs:
                    s = p(seq[:, :i], s)
                assert torch.equal(torch.isinf(s), torch.isinf(scores[i])), (case, i, kw, [type(p).__name__ for p in procs])
                fin = ~torch.isinf(s)
                assert torch.allclose(s[fin], scores[i][fin], rtol=1e-4, atol=1e-4, equal_nan=True), (case, i, kw, [type(p).__name__ for p in procs])   # upstream's own nan (length decay on a masked eos) included
        checked += 1
    assert checked >= 60


def test_generation_config_fields_are_honoured_or_refused_never_dropped():
    """the length rules of HF generate under `inputs_embeds` (generation/utils.py `_prepare_generated_length`), which fields send a request to the host-driven
    path, and that every generation feature without an implementation is refused BY NAME (as is an unknown keyword, HF's `_validate_model_kwargs`)"""
    sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd"))
    from transformers import GenerationConfig
    from visualcla.logits_processors import min_token_floor, needs_host_processors, new_token_budget, refuse_unsupported
    G = GenerationConfig
    assert new_token_budget(G(max_new_tokens=7, max_length=3), 100) == 7          # max_new_tokens wins
    assert new_token_budget(G(max_length=140), 128) == 12                          # an explicit max_length counts the prompt
    assert new_token_budget(G(), 128) == 20
    for ml in (128, 100):
        with pytest.raises(ValueError, match="max_length"):
            new_token_budget(G(max_length=ml), 128)
    assert min_token_floor(G(min_new_tokens=5, min_length=500), 128) == 5
    assert min_token_floor(G(min_length=130), 128) == 2 and min_token_floor(G(min_length=0), 128) == 0 and min_token_floor(G(), 128) == 0
    assert not needs_host_processors(G(do_sample=True, top_k=40, top_p=0.9, temperature=0.5, repetition_penalty=1.1, no_repeat_ngram_size=15, min_length=0))
    assert not needs_host_processors(G(do_sample=False, min_p=0.1, typical_p=0.5))          # warpers without sampling do nothing upstream either
    for kw in (dict(bad_words_ids=[[3]]), dict(suppress_tokens=[3]), dict(begin_suppress_tokens=[3]), dict(forced_eos_token_id=2), dict(renormalize_logits=True),
               dict(sequence_bias=[[[4], 1.0]]), dict(do_sample=True, min_p=0.1), dict(do_sample=True, typical_p=0.5), dict(do_sample=True, eta_cutoff=0.1),
               dict(exponential_decay_length_penalty=(2, 1.1)), dict(remove_invalid_values=True)):
        assert needs_host_processors(G(**kw)), kw
    refuse_unsupported(G(max_new_tokens=3, do_sample=True, top_k=5, num_beams=4, length_penalty=0.5, use_cache=True), {})
    for kw, word in ((dict(guidance_scale=1.5), "guidance_scale"), (dict(penalty_alpha=0.6, top_k=4), "penalty_alpha"), (dict(return_dict_in_generate=True), "return_dict"),
                     (dict(return_dict_in_generate=True, output_scores=True), "output_scores"), (dict(stop_strings=["a"]), "stop_strings")):
        with pytest.raises(ValueError, match=word):
            refuse_unsupported(G(**kw), {})
    with pytest.raises(ValueError, match="streamer"):
        refuse_unsupported(G(), {"streamer": object()})


def test_ring_gemm_loops_hold_no_vmem_the_hand_count_does_not_know():
    """gemm_ring_kernel (csrc/gemm_ring.hip) waits for its LDS-DMA stages with hand-counted `s_waitcnt vmcnt((NS - 2) * PP)` over inline-asm
    `global_load_lds_dwordx4` that hipcc does not see: vmcnt retires in order, so ONE compiler-emitted VMEM instruction (an epilogue load hoisted over the
    loop, a scratch spill) between the first DMA and the last would shift every count and a stage would be read before it lands.  Audits every shipped
    instantiation: up to its last DMA instruction the kernel issues no other global / buffer / scratch / flat instruction, and the waits it contains are
    exactly the counts the source derives from the template parameters (PP = pieces per wave and stage, dummies included)."""
    import re
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    so = os.path.join(ROOT, "visual-chinese-llama-alpaca_amd", "visualcla", "libvisualcla_hip.so")
    if not (os.path.exists(so) and all(os.path.exists(os.path.join(llvm, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump", "llvm-readelf"))):
        pytest.skip("library not built or ROCm LLVM tools absent")
    seen = 0
    with tempfile.TemporaryDirectory() as td:
        fb = os.path.join(td, "fat.bin")
        subprocess.check_call([f"{llvm}/llvm-objcopy", "--dump-section", f".hip_fatbin={fb}", so])
        data = open(fb, "rb").read()
        offs = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), data)]
        for k, o in enumerate(offs):
            bf, co = os.path.join(td, f"b{k}.bin"), os.path.join(td, f"b{k}.co")
            with open(bf, "wb") as f:
                f.write(data[o:offs[k + 1] if k + 1 < len(offs) else len(data)])
            subprocess.check_call([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={bf}",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
            notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
            syms = sorted(set(re.findall(r"\.name:\s+(_Z16gemm_ring_kernel\S+)", notes)))
            if not syms:
                continue
            dis = subprocess.run([f"{llvm}/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout
            for sym in syms:
                m = re.match(r"_Z16gemm_ring_kernelILi(\d+)E[tf]Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELb([01])E", sym)
                assert m, sym
                _, BM, BN, _, _, NS, KS, W8, _ = (int(x) for x in m.groups())
                i = dis.index(f"<{sym}>:")
                body = [ln.split("//")[0].strip() for ln in dis[i:dis.find("\n\n", i)].splitlines()[1:]]
                vmem = [(n, ln) for n, ln in enumerate(body) if re.match(r"(global_|buffer_|scratch_|flat_)", ln)]
                dma = [n for n, ln in vmem if ln.startswith("global_load_lds_dwordx4")]
                pp = KS * (-(-(BM // 8) // 8) + -(-(BN // (16 if W8 else 8)) // 8))          # 1 KiB pieces per slab: BM / 8 of A, BN / 8 of bf16 W (BN / 16 of e4m3), over 8 waves
                assert len(dma) >= 2 * pp, (sym, len(dma))
                strangers = [ln for n, ln in vmem if n < dma[-1] and not ln.startswith("global_load_lds_dwordx4")]
                assert not strangers, f"{sym}: VMEM instructions the hand count does not know, before the last DMA: {strangers[:4]}"
                waits = {int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", "\n".join(body[:dma[-1] + 1]))}
                want = {(NS - 2) * pp} | ({(NS - 1) * pp} if W8 else set())
                assert want <= waits <= want | {0}, (sym, waits, want)
                seen += 1
    assert seen >= 20, seen


def test_generation_config_resolution_follows_transformers_priority():
    """user keywords > the passed generation_config > the model's own generation config (every field left at None, not only the special tokens):
    hf generation/utils.py `_prepare_generation_config`, which the reference reaches through text_model.generate.  Compared field by field with what a
    tiny HF LLaMA resolves for the same three inputs."""
    import warnings
    sys.path.insert(0, os.path.join(ROOT, "visual-chinese-llama-alpaca_amd"))
    from transformers import GenerationConfig, LlamaConfig, LlamaForCausalLM
    from visualcla.modeling_visualcla import VisualCLAModel
    own = GenerationConfig(bos_token_id=1, eos_token_id=2, pad_token_id=0, temperature=0.6, top_p=0.9, repetition_penalty=1.05)
    hf = LlamaForCausalLM(LlamaConfig(vocab_size=16, hidden_size=16, intermediate_size=32, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=2))
    hf.generation_config = own
    stub = SimpleNamespace(generation_config=own)
    fields = ("max_new_tokens", "max_length", "min_length", "do_sample", "temperature", "top_k", "top_p", "repetition_penalty", "no_repeat_ngram_size",
              "num_beams", "eos_token_id", "bos_token_id", "pad_token_id", "length_penalty", "bad_words_ids")
    from visualcla.modeling_visualcla import _HF_GLOBAL_GENERATION_DEFAULTS
    if hasattr(own, "_get_default_generation_params"):
        upstream = own._get_default_generation_params()
        assert all(upstream[k] == v for k, v in _HF_GLOBAL_GENERATION_DEFAULTS.items()), {k: (v, upstream.get(k)) for k, v in _HF_GLOBAL_GENERATION_DEFAULTS.items()}
    for passed, kw in ((None, {}), (None, dict(max_new_tokens=5, do_sample=True, top_k=7)), (None, dict(do_sample=True)),
                       (GenerationConfig(max_new_tokens=9, do_sample=True, top_k=40, temperature=0.5, eos_token_id=None), {}),
                       (GenerationConfig(max_new_tokens=9, top_p=0.5, no_repeat_ngram_size=3), dict(eos_token_id=None, temperature=1.3, num_beams=2)),
                       (GenerationConfig(eos_token_id=[2, 5], pad_token_id=7, bad_words_ids=[[3]]), dict(max_length=30))):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want, _ = hf._prepare_generation_config(passed, **dict(kw))
        got = VisualCLAModel._resolve_generation_config(stub, passed, dict(kw))
        for f in fields:
            w, g = getattr(want, f, None), getattr(got, f, None)
            if f in ("max_length", "min_length") and g is None:
                assert w == {"max_length": 20, "min_length": 0}[f]        # "not set" stays None here: visualcla.logits_processors' length rules test for it
                continue
            assert g == w, (f, passed, kw, g, w)                         # incl. the global defaults (top_k = 50!) and an explicit eos_token_id=None keyword


def test_engine_stream_is_every_cu_s_weights_in_consumption_order():
    """weights.add_engine_stream (the persistent B = 1 decode step's weight twin, csrc/decode_engine.hip): walking CU c's slots exactly as the kernel's
    consumers do -- qkv two rows per slot, o_proj / down_proj K-major (16 rows x 512 inputs, down_proj's inputs in mailbox-granule order), gate/up one
    SwiGLU unit per slot, lm_head two rows per slot -- must reproduce the plain matrix products, for an even and an odd number of units per CU."""
    from visualcla import weights as W
    D, H, V, L = W.ENGINE_D, W.ENGINE_H, 1100, 1
    gen = torch.Generator().manual_seed(0)

    def pw(n, k):
        m = torch.zeros(W.pad_to(n, 128), k, dtype=torch.bfloat16)
        m[:n] = (torch.randn(n, k, generator=gen) * 0.02).to(torch.bfloat16)
        return m
    x = torch.randn(D, generator=gen).bfloat16().float()
    for inter in (512, 768):                    # 2 and 3 units per CU (the odd one leaves a zero in every CU's last granule)
        gate, up = pw(inter, D)[:inter], pw(inter, D)[:inter]
        packed = {"llama.l0.wqkv": pw(3 * D, D), "llama.l0.wo": pw(D, D), "llama.l0.wgu": W._pack_w(W.interleave_gate_up(gate, up), "cpu"),
                  "llama.l0.wd": pw(D, inter), "llama.lm_head": pw(V, D), "llama.l0.ln1.g": torch.ones(D), "llama.l0.ln2.g": torch.full((D,), 2.0),
                  "llama.norm.g": torch.full((D,), 3.0)}
        # (the kernel needs >= 8 slots per operator for its register preloads; the LAYOUT is what is checked here, so build it directly)
        g = dict(upc=inter // 256, gpc=(inter // 256 + 1) // 2, s_lm=(V + 511) // 512)
        g.update(slots_layer=32 + g["upc"] + g["gpc"], slots_total=32 + g["upc"] + g["gpc"] + g["s_lm"])
        orig = W.engine_geometry
        W.engine_geometry = lambda *a_: g
        try:
            W.add_engine_stream(packed, D, H, inter, V, L)
        finally:
            W.engine_geometry = orig
        # in memory: [slot][CU][16 KiB] -- slot g of CU c at (g * 256 + c) * 16 KiB (the loaders of all CUs sweep one moving window); read here per CU
        assert packed["llama.engine.w"].shape == (g["slots_total"], 256, 2 * D) and packed["llama.engine.w"].is_contiguous()
        st = packed["llama.engine.w"].float().permute(1, 0, 2)
        assert torch.equal(packed["llama.engine.g"][:, 0], torch.tensor([1.0, 2.0, 3.0]))
        upc, gpc = g["upc"], g["gpc"]
        cu = torch.arange(256)
        # qkv: slot j of CU c = rows part*D + (c // 8)*128 + (c % 8)*16 + 2 (j % 8), + 1 with part = j // 8
        ref = packed["llama.l0.wqkv"][:3 * D].float() @ x
        got = (st[:, 0:24].reshape(256, 24, 2, D) @ x)
        j = torch.arange(24)
        rows = (j[None, :] // 8) * D + (cu[:, None] // 8) * 128 + (cu[:, None] % 8) * 16 + 2 * (j[None, :] % 8)
        assert torch.allclose(got[..., 0], ref[rows], atol=1e-4) and torch.allclose(got[..., 1], ref[rows + 1], atol=1e-4)
        # o_proj, K-major: slot j = rows 16 c .. 16 c + 15 against inputs 512 j .. 512 j + 511
        ref = packed["llama.l0.wo"][:D].float() @ x
        got = (st[:, 24:32].reshape(256, 8, 16, 512) * x.view(1, 8, 1, 512)).sum(dim=(1, 3))
        assert torch.allclose(got.reshape(-1), ref, atol=1e-4)
        # gate/up: slot j of CU c = (gate row, up row) of unit upc * c + j; its activation goes to granule gpc * c + j // 2, half j % 2
        gu = st[:, 32:32 + upc].reshape(256, upc, 2, D) @ x
        u = upc * cu[:, None] + torch.arange(upc)[None, :]
        assert torch.allclose(gu[..., 0], (gate.float() @ x)[u], atol=1e-4) and torch.allclose(gu[..., 1], (up.float() @ x)[u], atol=1e-4)
        act = torch.zeros(256, gpc, 2)
        act.view(256, 2 * gpc)[:, :upc] = gu[..., 0] * gu[..., 1]
        # down_proj, K-major over the mailbox order: slot j multiplies granules 256 j .. 256 j + 255
        ref = packed["llama.l0.wd"][:D, :inter].float() @ (gate.float() @ x * (up.float() @ x))
        got = (st[:, 32 + upc:32 + upc + gpc].reshape(256, gpc, 16, 512) * act.reshape(-1).view(1, gpc, 1, 512)).sum(dim=(1, 3))
        assert torch.allclose(got.reshape(-1), ref, atol=2e-4)
        assert torch.equal(W.engine_down_kmap(inter).view(256, gpc, 2)[:, :, :][..., 0][:, 0], upc * cu)
        # lm_head: slot j of CU c = rows 2 s_lm c + 2 j, + 1; rows past the vocabulary are zero
        lm = (st[:, g["slots_layer"]:].reshape(256, g["s_lm"], 2, D) @ x).reshape(-1)
        assert torch.allclose(lm[:V], packed["llama.lm_head"][:V].float() @ x, atol=1e-4) and float(lm[V:].abs().max()) == 0.0
    # the geometry the kernel accepts: LLaMA-7B yes, other widths / too few slots per operator no (those models keep the launch path)
    assert W.engine_geometry(4096, 32, 11008, 49958, 32) == dict(upc=43, gpc=22, s_lm=98, slots_layer=97, slots_total=32 * 97 + 98)
    assert W.engine_geometry(5120, 40, 13824, 49958, 40) is None and W.engine_geometry(4096, 32, 11008 + 64, 49958, 32) is None
    assert W.engine_geometry(4096, 32, 1024, 49958, 2) is None and W.engine_geometry(4096, 32, 11008, 1000, 2) is None


def test_graft_entry_build_is_what_the_driver_runs():
    """`__graft_entry__.build()` -- the driver's "does it build" check -- must pass on this tree: make (incremental), import, and the library's ABI version
    against the header's (round 6 shipped most of its commits with a stale `== 4` there: nothing in the suite ran it)"""
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT, capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "build ok" in r.stdout
