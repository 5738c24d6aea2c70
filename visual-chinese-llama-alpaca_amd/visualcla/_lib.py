"""ctypes binding of libvisualcla_hip.so (include/visualcla_hip.h).

The product path has no CPU fallback: if the shared library is missing, or the device is
not a gfx950, every entry point raises.  Tensors stay torch-owned; only raw device pointers,
sizes and the current HIP stream cross the boundary.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

VCLA_F32, VCLA_BF16 = 0, 1
EPI_NONE, EPI_QUICK_GELU, EPI_GELU_ERF, EPI_SWIGLU = 0, 1, 2, 3

_ERR_NAMES = {1: "BAD_SHAPE", 2: "BAD_DTYPE", 3: "UNSUPPORTED_ARCH", 4: "HIP", 5: "BAD_ARG", 6: "WORKSPACE",
              7: "MISSING_TENSOR"}

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VCLA_LIB", os.path.join(_HERE, "libvisualcla_hip.so"))


class VclaError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("W", C.c_void_p), ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("ldr", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("epilogue", C.c_int), ("out_f32", C.c_int),
        ("c_group_rows", C.c_int), ("c_group_stride", C.c_int), ("c_row_offset", C.c_int),
        ("force_kernel", C.c_int),
        ("norm_gamma", C.c_void_p), ("norm_eps", C.c_float),
        ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_size_t),
        ("W_frag", C.c_void_p),
        ("W_q8", C.c_void_p), ("W_q8_frag", C.c_void_p), ("w_scale", C.c_void_p),
        ("post_norm_gamma", C.c_void_p), ("post_norm_eps", C.c_float), ("post_norm_out", C.c_void_p), ("post_norm_ld", C.c_int64),
        ("A_frag", C.c_void_p), ("C_frag", C.c_void_p),
        ("c_frag_gamma", C.c_void_p), ("c_row_ssq", C.c_void_p), ("a_row_ssq", C.c_void_p), ("a_row_ssq_parts", C.c_int),
        ("a_norm_eps", C.c_float),
        ("A_q8", C.c_void_p), ("a_scale", C.c_void_p),
        ("ds_splitk", C.c_int),
        ("ds_raw_partials", C.c_int),
        ("A_slab", C.c_void_p), ("a_slab_rows", C.c_int64), ("W_slab", C.c_void_p), ("W_q8_slab", C.c_void_p),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p),
        ("q_bs", C.c_int64), ("q_hs", C.c_int64), ("q_rs", C.c_int64),
        ("k_bs", C.c_int64), ("k_hs", C.c_int64), ("k_rs", C.c_int64),
        ("v_bs", C.c_int64), ("v_hs", C.c_int64), ("v_rs", C.c_int64),
        ("o_bs", C.c_int64), ("o_hs", C.c_int64), ("o_rs", C.c_int64),
        ("B", C.c_int), ("H", C.c_int), ("Tq", C.c_int), ("Tk", C.c_int), ("D", C.c_int),
        ("scale", C.c_float), ("causal", C.c_int),
        ("key_mask", C.c_void_p), ("key_mask_ld", C.c_int64),
        ("tk_dev", C.c_void_p), ("tk_dev_add", C.c_int),
        ("force_kernel", C.c_int),
    ]


class ModelCfg(C.Structure):
    _fields_ = [
        ("act_dtype", C.c_int),
        ("v_hidden", C.c_int), ("v_layers", C.c_int), ("v_heads", C.c_int), ("v_inter", C.c_int),
        ("v_patch", C.c_int), ("v_image", C.c_int), ("v_channels", C.c_int), ("v_eps", C.c_float),
        ("r_hidden", C.c_int), ("r_layers", C.c_int), ("r_heads", C.c_int), ("r_inter", C.c_int),
        ("r_queries", C.c_int), ("r_eps", C.c_float),
        ("t_hidden", C.c_int), ("t_layers", C.c_int), ("t_heads", C.c_int), ("t_inter", C.c_int),
        ("t_vocab", C.c_int), ("t_max_pos", C.c_int), ("t_eps", C.c_float), ("t_rope_theta", C.c_float),
        ("t_fp8_mfma", C.c_int),
        ("t_kv_fp8", C.c_int),
    ]


SAMPLE_MAX_TOP_K, SAMPLE_MAX_EOS, SAMPLE_KEPT_LD, SAMPLE_MAX_HIST, SAMPLE_MAX_VOCAB = 256, 4, 512, 4096, 53248


class SampleArgs(C.Structure):
    _fields_ = [
        ("repetition_penalty", C.c_float), ("no_repeat_ngram_size", C.c_int), ("min_new_tokens", C.c_int),
        ("n_eos", C.c_int), ("eos_ids", C.c_int * SAMPLE_MAX_EOS),
        ("temperature", C.c_float), ("top_k", C.c_int), ("top_p", C.c_double), ("min_tokens_to_keep", C.c_int),
        ("uniforms", C.c_void_p), ("history", C.c_void_p),
        ("kept_ids", C.c_void_p), ("kept_probs", C.c_void_p), ("n_kept", C.c_void_p),
    ]


def sample_args(repetition_penalty=1.0, no_repeat_ngram_size=0, min_new_tokens=0, eos_ids=(), temperature=1.0, top_k=1,
                top_p=1.0, min_tokens_to_keep=1, uniforms=None, history=None, kept_ids=None, kept_probs=None, n_kept=None):
    a = SampleArgs()
    a.repetition_penalty, a.no_repeat_ngram_size, a.min_new_tokens = float(repetition_penalty), int(no_repeat_ngram_size), int(min_new_tokens)
    eos_ids = list(eos_ids)[:SAMPLE_MAX_EOS]
    a.n_eos = len(eos_ids)
    for i, e in enumerate(eos_ids):
        a.eos_ids[i] = int(e)
    a.temperature, a.top_k, a.top_p, a.min_tokens_to_keep = float(temperature), int(top_k), float(top_p), int(min_tokens_to_keep)
    a.uniforms, a.history = ptr(uniforms), ptr(history)
    a.kept_ids, a.kept_probs, a.n_kept = ptr(kept_ids), ptr(kept_probs), ptr(n_kept)
    return a


# every symbol include/visualcla_hip.h declares: name -> (restype, argtypes)
_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
SYMBOLS = {
    "vcla_version": (_i, []),
    "vcla_last_error": (C.c_char_p, []),
    "vcla_device_check": (_i, []),
    "vcla_layernorm": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _f, _i, _vp]),
    "vcla_rmsnorm": (_i, [_vp, _i64, _vp, _vp, _i64, _i, _i, _f, _i, _vp]),
    "vcla_gemm": (_i, [C.POINTER(GemmArgs), _i, _vp]),
    "vcla_rmsnorm_pack": (_i, [_vp, _i64, _vp, _vp, _i, _i, _f, _vp]),
    "vcla_quant_fp8_rows": (_i, [_vp, _i64, _vp, _vp, _i, _i, _vp]),
    "vcla_im2col": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "vcla_vit_assemble": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "vcla_attention": (_i, [C.POINTER(AttnArgs), _i, _vp]),
    "vcla_image_preprocess": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, C.c_double, C.POINTER(C.c_float),
                                   C.POINTER(C.c_float), _vp, _i, _vp]),
    "vcla_image_preprocess_batch": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, C.c_double, C.POINTER(C.c_float),
                                         C.POINTER(C.c_float), _vp, _i, _vp]),
    "vcla_check_request": (_i, [_vp, _i, _i, _i, _i, _i64, _i64, _i64, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "vcla_embed_splice": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "vcla_rope_kv_append": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "vcla_attn_decode_fused": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i64, _f, _i, _i, _vp]),
    "vcla_attn_decode_fused_parts": (_i, [_vp, _i64, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i64, _f, _i, _i, _vp]),
    "vcla_argmax": (_i, [_vp, _i64, _vp, _i, _i, _vp]),
    "vcla_causal_lm_loss": (_i, [_vp, _i64, _vp, _i, _i, _i, _i64, _vp, _vp, _vp]),
    "vcla_sample": (_i, [_vp, _i64, _i, _i, _i, _vp, C.POINTER(SampleArgs), _vp, _vp]),
    "vcla_ctx_create": (_i, [C.POINTER(ModelCfg), C.POINTER(_vp)]),
    "vcla_ctx_destroy": (None, [_vp]),
    "vcla_ctx_set_tensor": (_i, [_vp, C.c_char_p, _vp, _sz]),
    "vcla_ctx_finalize": (_i, [_vp]),
    "vcla_vision_workspace_bytes": (_sz, [_vp, _i]),
    "vcla_llama_workspace_bytes": (_sz, [_vp, _i, _i]),
    "vcla_kv_cache_bytes": (_sz, [_vp, _i, _i]),
    "vcla_vision_forward": (_i, [_vp, _vp, _vp, _i, _vp, _sz, _vp, _vp, _vp]),
    "vcla_llama_prefill": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _sz, _vp, _vp]),
    "vcla_llama_decode_step": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vcla_llama_decode_loop": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _sz, _i, _vp]),
    "vcla_llama_decode_status": (_i, [_vp, _i, _vp, _sz, _vp]),
    "vcla_llama_decode_loop_sampled": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _sz, _i, C.POINTER(SampleArgs), _i, _vp]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared library (once).  Raises VclaError if it is absent -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VclaError(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py build` "
            f"(or `make -C visual-chinese-llama-alpaca_amd/csrc`). The VisualCLA HIP path has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().vcla_last_error().decode("utf-8", "replace")
        kind = _ERR_NAMES.get(rc, str(rc))
        if rc in (1, 2, 5):
            raise ValueError(f"visualcla_hip[{kind}]: {msg}")
        raise VclaError(f"visualcla_hip[{kind}]: {msg}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return VCLA_F32
    if dt == torch.bfloat16:
        return VCLA_BF16
    raise ValueError(f"activation dtype must be float32 or bfloat16, got {dt}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_device() -> None:
    """Fail loudly when there is no usable MI355X."""
    if not torch.cuda.is_available():
        raise VclaError("no HIP device visible: the VisualCLA HIP path needs an MI355X (gfx950); there is no CPU fallback")
    check(load().vcla_device_check())


# ------------------------------------------------------------------ primitive wrappers (used by tests / tools)
def layernorm(x, gamma, beta, eps, out=None):
    lib = load()
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    x2 = x.reshape(rows, cols)
    assert x2.stride(1) == 1
    out = torch.empty_like(x2) if out is None else out
    check(lib.vcla_layernorm(ptr(x2), x2.stride(0), ptr(gamma), ptr(beta), ptr(out), out.stride(0), rows, cols,
                             eps, dtype_code(x.dtype), stream_ptr()))
    return out.view(x.shape)


def rmsnorm(x, gamma, eps, out=None):
    lib = load()
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    x2 = x.reshape(rows, cols)
    out = torch.empty_like(x2) if out is None else out
    check(lib.vcla_rmsnorm(ptr(x2), x2.stride(0), ptr(gamma), ptr(out), out.stride(0), rows, cols, eps,
                           dtype_code(x.dtype), stream_ptr()))
    return out.view(x.shape)


def to_frag(a: torch.Tensor) -> torch.Tensor:
    """[M, K] bf16 row-major -> the fragment-major activation layout of vcla_gemm_args.A_frag, [K/32, ceil(M/16), 64, 8]
    (host-side twin of vcla_rmsnorm_pack(gamma=NULL); rows past M are zero)."""
    M, K = a.shape
    mt = (M + 15) // 16
    ap = torch.zeros(mt * 16, K, dtype=a.dtype, device=a.device)
    ap[:M] = a
    return ap.view(mt, 16, K // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous().view(K // 32, mt, 64, 8)


def from_frag(f: torch.Tensor, M: int) -> torch.Tensor:
    """inverse of to_frag: [K/32, MT, 64, 8] -> [M, K]"""
    ks, mt = f.shape[0], f.shape[1]
    return f.view(ks, mt, 4, 16, 8).permute(1, 3, 0, 2, 4).contiguous().view(mt * 16, ks * 32)[:M]


def rmsnorm_pack(x, gamma, eps, out=None):
    """[M <= 64, K] bf16 -> fragment-major RMSNorm(x) (gamma None: plain re-layout), [K/32, ceil(M/16), 64, 8]"""
    M, K = x.shape
    if out is None:
        out = torch.zeros(K // 32, (M + 15) // 16, 64, 8, dtype=torch.bfloat16, device=x.device)
    check(load().vcla_rmsnorm_pack(ptr(x), x.stride(0), ptr(gamma), ptr(out), M, K, float(eps), stream_ptr()))
    return out


def quant_fp8_rows(x):
    """[M, K] bf16 -> (uint8 [M, K] e4m3fn bits, fp32 [M] per-row scale): the activation operand of the fp8 MFMA GEMM"""
    M, K = x.shape
    q = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    sc = torch.empty(M, dtype=torch.float32, device=x.device)
    check(load().vcla_quant_fp8_rows(ptr(x), x.stride(0), ptr(q), ptr(sc), M, K, stream_ptr()))
    return q, sc


def gemm(a, w_packed, n, bias=None, residual=None, epilogue=EPI_NONE, out_f32=False, out=None, force_kernel=0,
         group_rows=0, group_stride=0, row_offset=0, norm_gamma=None, norm_eps=0.0, splitk_ws=None, w_frag=None, w_q8=None, w_q8_frag=None, w_scale=None,
         post_norm_gamma=None, post_norm_eps=0.0, post_norm_out=None, a_frag=None, c_frag=None, m=None,
         c_frag_gamma=None, c_row_ssq=None, a_row_ssq=None, a_norm_eps=0.0, a_q8=None, a_scale=None, ds_splitk=0, ds_raw_partials=False,
         a_slab=None, w_slab=None, w_q8_slab=None, k=None):
    """a [M, K] (fp32 | bf16, row-major), w_packed [N_pad, K] bf16 -> [M, N_out].  a_frag ([K/32, MT, 64, 8], with m = M) selects
    the streaming decode kernel; c_frag (same layout over N_out) receives a fragment-major copy of the output."""
    lib = load()
    if a is None and a_q8 is not None:
        M, K = a_q8.shape
        adt, adev = torch.bfloat16, a_q8.device
    elif a is None and a_slab is not None:        # slab-major activations [K/64, rows, 64]
        M, K = int(m), a_slab.shape[0] * 64
        adt, adev = torch.bfloat16, a_slab.device
    elif a is None:
        M, K = int(m), a_frag.shape[0] * 32
        adt, adev = torch.bfloat16, a_frag.device
    else:
        M, K = a.shape
        adt, adev = a.dtype, a.device
    n_out = n // 2 if epilogue == EPI_SWIGLU else n
    if out is None:
        odt = torch.float32 if (out_f32 or adt == torch.float32) else torch.bfloat16
        out = torch.empty(M, n_out, dtype=odt, device=adev)
    args = GemmArgs()
    args.A, args.lda = ptr(a), (a.stride(0) if a is not None else 0)
    args.A_frag, args.C_frag = ptr(a_frag), ptr(c_frag)
    args.A_q8, args.a_scale = ptr(a_q8), ptr(a_scale)
    args.A_slab, args.a_slab_rows = ptr(a_slab), (a_slab.shape[1] if a_slab is not None else 0)
    args.W_slab, args.W_q8_slab = ptr(w_slab), ptr(w_q8_slab)
    args.ds_splitk = int(ds_splitk)
    args.ds_raw_partials = int(bool(ds_raw_partials))
    args.c_frag_gamma, args.c_row_ssq, args.a_row_ssq = ptr(c_frag_gamma), ptr(c_row_ssq), ptr(a_row_ssq)
    args.a_row_ssq_parts, args.a_norm_eps = (a_row_ssq.shape[1] if a_row_ssq is not None else 0), float(a_norm_eps)
    args.W, args.bias = ptr(w_packed), ptr(bias)
    args.residual, args.ldr = ptr(residual), (residual.stride(0) if residual is not None else 0)
    args.C, args.ldc = ptr(out), out.stride(0)
    args.M, args.N, args.K = M, n, K
    args.epilogue, args.out_f32 = epilogue, int(bool(out_f32))
    args.c_group_rows, args.c_group_stride, args.c_row_offset = group_rows, group_stride, row_offset
    args.force_kernel = force_kernel
    args.norm_gamma, args.norm_eps = ptr(norm_gamma), float(norm_eps)
    args.W_frag = ptr(w_frag)
    args.W_q8, args.W_q8_frag, args.w_scale = ptr(w_q8), ptr(w_q8_frag), ptr(w_scale)
    args.splitk_ws = ptr(splitk_ws)
    args.splitk_ws_bytes = splitk_ws.numel() * splitk_ws.element_size() if splitk_ws is not None else 0
    if post_norm_gamma is not None:
        args.post_norm_gamma, args.post_norm_eps = ptr(post_norm_gamma), float(post_norm_eps)
        args.post_norm_out, args.post_norm_ld = ptr(post_norm_out), post_norm_out.stride(0)
    check(lib.vcla_gemm(C.byref(args), dtype_code(adt), stream_ptr()))
    return out


def attention(q, k, v, scale, causal=False, key_mask=None, out=None, force_kernel=0):
    """q [B,H,Tq,D], k/v [B,H,Tk,D] (any strides with unit last stride) -> o [B,Tq,H*D]."""
    lib = load()
    B, H, Tq, D = q.shape
    Tk = k.shape[2]
    assert q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1
    if out is None:
        out = torch.empty(B, Tq, H * D, dtype=q.dtype, device=q.device)
    a = AttnArgs()
    a.q, a.k, a.v, a.o = ptr(q), ptr(k), ptr(v), ptr(out)
    a.q_bs, a.q_hs, a.q_rs = q.stride(0), q.stride(1), q.stride(2)
    a.k_bs, a.k_hs, a.k_rs = k.stride(0), k.stride(1), k.stride(2)
    a.v_bs, a.v_hs, a.v_rs = v.stride(0), v.stride(1), v.stride(2)
    a.o_bs, a.o_hs, a.o_rs = out.stride(0), D, out.stride(1)
    a.B, a.H, a.Tq, a.Tk, a.D = B, H, Tq, Tk, D
    a.scale, a.causal = float(scale), int(bool(causal))
    a.key_mask, a.key_mask_ld = ptr(key_mask), (key_mask.stride(0) if key_mask is not None else 0)
    a.tk_dev, a.tk_dev_add, a.force_kernel = None, 0, force_kernel
    check(lib.vcla_attention(C.byref(a), dtype_code(q.dtype), stream_ptr()))
    return out


def sample(logits, args: SampleArgs, n_hist: int = 0):
    """logits [B, V] fp32 (modified in place) -> next tokens [B] int64; args.history / args.uniforms as the header documents"""
    assert logits.dtype == torch.float32 and logits.dim() == 2 and logits.stride(1) == 1
    out = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
    check(load().vcla_sample(logits.data_ptr(), logits.stride(0), logits.shape[0], logits.shape[1], int(n_hist), None, C.byref(args),
                             out.data_ptr(), stream_ptr()))
    return out


def causal_lm_loss(logits, labels, ignore_index: int = -100):
    """logits fp32 [B, T, V] (contiguous rows), labels int64 [B, T] -> scalar fp32 tensor: mean shifted cross-entropy (HF ForCausalLMLoss)"""
    assert logits.dtype == torch.float32 and logits.dim() == 3 and logits.stride(2) == 1 and logits.stride(0) == logits.shape[1] * logits.stride(1)
    B, T, V = logits.shape
    labels = labels.to(device=logits.device, dtype=torch.int64).contiguous()
    if labels.shape != (B, T):
        raise ValueError(f"labels {tuple(labels.shape)} do not match logits {(B, T)}")
    ws = torch.empty(B * T, dtype=torch.float32, device=logits.device)
    out = torch.empty(1, dtype=torch.float32, device=logits.device)
    check(load().vcla_causal_lm_loss(ptr(logits), logits.stride(1), ptr(labels), B, T, V, int(ignore_index), ptr(ws), ptr(out), stream_ptr()))
    return out[0]


def argmax(logits):
    lib = load()
    B, V = logits.shape
    out = torch.empty(B, dtype=torch.int64, device=logits.device)
    check(lib.vcla_argmax(ptr(logits), logits.stride(0), ptr(out), B, V, stream_ptr()))
    return out
