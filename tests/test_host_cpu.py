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
    assert lib.vcla_version() == 4


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
    assert s256["images_per_sec"] > 0 and s256["images_per_sec_prefill"] > 0


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
