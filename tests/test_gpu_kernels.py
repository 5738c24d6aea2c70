"""Per-kernel parity: every C-ABI primitive against the oracle's restatement of the same op (fp32 CPU),
on seeded inputs.  Integer/index outputs must match exactly; floating point within the tolerance written
next to each check.  All calls go through libvisualcla_hip.so via ctypes."""
import ctypes as C
import math
import os

import pytest
import torch

from oracle import visualcla_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def lib():
    from visualcla import _lib
    _lib.require_device()
    return _lib


def _report(line: str):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_report.txt", "a") as f:
        f.write(line + "\n")


def bf16r(t):
    return t.to(torch.bfloat16).float()


def _cmp(name, got, ref, atol, rtol=0.0):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    worst = (err - bound).max().item()
    _report(f"{name}: max_abs_err={err.max().item():.3e} ref_absmax={ref.abs().max().item():.3e}")
    assert worst <= 0, f"{name}: max err {err.max().item():.3e} exceeds atol={atol} rtol={rtol}"


# ------------------------------------------------------------------ norms
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", [(5, 128), (257, 1024), (33, 4096), (3, 100), (6, 512), (1030, 1536), (2, 2048)])
def test_layernorm(lib, dtype, rows, cols):
    g = torch.Generator().manual_seed(rows * 7 + cols)
    x = torch.randn(rows, cols, generator=g) * 2 + 0.3
    gamma, beta = torch.randn(cols, generator=g), torch.randn(cols, generator=g)
    xd = x.to(dtype)
    ref = O.layer_norm(xd.float(), gamma, beta, 1e-5)
    got = lib.layernorm(xd.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5)
    # fp32: summation order only; bf16: one output rounding (2^-9 relative)
    _cmp(f"layernorm[{dtype},{rows}x{cols}]", got, ref, atol=2e-5 if dtype == torch.float32 else 1e-3, rtol=0 if dtype == torch.float32 else 4e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", [(4, 256), (128, 4096), (7, 512)])
def test_rmsnorm(lib, dtype, rows, cols):
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * 1.5).to(dtype)
    gamma = 1 + 0.1 * torch.randn(cols, generator=g)
    ref = O.llama_rmsnorm(x, gamma, 1e-6).float()     # oracle applies HF's rounding order in `dtype`
    got = lib.rmsnorm(x.to(DEV), gamma.to(DEV), 1e-6)
    _cmp(f"rmsnorm[{dtype},{rows}x{cols}]", got, ref, atol=2e-5 if dtype == torch.float32 else 1e-3, rtol=0 if dtype == torch.float32 else 8e-3)


def test_layernorm_strided_rows(lib):
    # last-token selection in prefill: rows picked with a stride larger than cols
    x = torch.randn(6, 4, 256)
    gamma, beta = torch.randn(256), torch.randn(256)
    xd, gd, bd = x.to(DEV), gamma.to(DEV), beta.to(DEV)   # keep device tensors alive across the call
    view = xd[:, 3, :]                       # stride 1024
    out = torch.empty(6, 256, device=DEV)
    lib.check(lib.load().vcla_layernorm(view.data_ptr(), view.stride(0), gd.data_ptr(), bd.data_ptr(),
                                        out.data_ptr(), 256, 6, 256, 1e-5, 0, lib.stream_ptr()))
    torch.cuda.synchronize()
    _cmp("layernorm[strided]", out, O.layer_norm(x[:, 3, :], gamma, beta, 1e-5), atol=2e-5)


# ------------------------------------------------------------------ GEMM
def _pack(w):
    from visualcla.weights import _pack_w
    return _pack_w(w, DEV)


def _gemm_ref(a, w, bias, epi, residual):
    y = a.float() @ w.float().t()
    if epi == 3:
        from visualcla.weights import interleave_gate_up  # noqa: F401  (documented layout)
        n = w.shape[0]
        yb = y + (bias if bias is not None else 0)
        blocks = yb.view(y.shape[0], n // 32, 2, 16)
        y = torch.nn.functional.silu(blocks[:, :, 0]) * blocks[:, :, 1]
        y = y.reshape(y.shape[0], n // 2)
    else:
        if bias is not None:
            y = y + bias
        if epi == 1:
            y = O.quick_gelu(y)
        elif epi == 2:
            y = O.gelu_erf(y)
    if residual is not None:
        y = y + residual.float()
    return y


GEMM_SHAPES = [
    # M, N, K
    (1, 256, 128), (3, 1000, 256), (8, 4096, 512), (16, 128, 64), (130, 200, 192), (257, 384, 128),
    (514, 1024, 1024), (64, 320, 640), (700, 1000, 2048), (2, 512, 4096), (33, 4096, 1408), (128, 288, 2048), (1, 4096, 11008),
]


@pytest.mark.parametrize("kernel", ["mfma", "mfma_ws", "mfma256", "mfma256b", "skinny", "panel", "panel_ws", "panel_frag", "gemv", "gemv_generic", "gemv32", "f32"])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm(lib, kernel, epi, M, N, K):
    if kernel.startswith("gemv") and M > 8:
        pytest.skip("gemv kernel is for M <= 8")
    if kernel in ("skinny", "panel", "panel_ws", "panel_frag") and M > 128:
        pytest.skip("skinny / panel kernels are for M <= 128")
    if epi == 3:
        N = (N + 31) // 32 * 32
    g = torch.Generator().manual_seed(M * 1000 + N + K + epi)
    dtype = torch.float32 if kernel in ("f32", "gemv32") else torch.bfloat16
    a = bf16r(torch.randn(M, K, generator=g))
    w = bf16r(torch.randn(N, K, generator=g) * 0.05)
    bias = bf16r(torch.randn(N, generator=g) * 0.1)
    n_out = N // 2 if epi == 3 else N
    res = bf16r(torch.randn(M, n_out, generator=g))
    ref = _gemm_ref(a, w, bias, epi, res)
    fk = {"mfma": 1, "mfma_ws": 1, "mfma256": 4, "mfma256b": 5, "skinny": 7, "panel": 8, "panel_ws": 8, "panel_frag": 8, "gemv": 2, "gemv_generic": 6, "gemv32": 2, "f32": 3}[kernel]
    # "mfma_ws": with a workspace the 128-tile kernel splits K when the problem has few tiles (e.g. 257 x 384 x 128 does not, 514 x 1024 x 1024 does)
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV) if kernel in ("panel_ws", "panel_frag", "mfma_ws") else None
    wp = _pack(w)
    wf = None
    if kernel == "panel_frag":
        from visualcla.weights import to_fragment_major, from_fragment_major
        wf = to_fragment_major(wp)
        assert torch.equal(from_fragment_major(wf), wp)
    got = lib.gemm(a.to(DEV, dtype), wp, N, bias=bias.to(DEV), residual=res.to(DEV, dtype), epilogue=epi,
                   force_kernel=fk, splitk_ws=ws, w_frag=wf)
    # inputs are exactly representable; products are exact in fp32; only accumulation order differs (+ one bf16
    # output rounding in bf16 mode)
    if dtype == torch.float32:
        _cmp(f"gemm[{kernel},epi{epi},{M}x{N}x{K}]", got, ref, atol=1e-4, rtol=1e-5)
    else:
        _cmp(f"gemm[{kernel},epi{epi},{M}x{N}x{K}]", got, ref, atol=2e-3, rtol=8e-3)


@pytest.mark.parametrize("M,N,K,epi", [(257, 1024, 4096, 0), (257, 4096, 1024, 1), (257, 3072, 1024, 0), (321, 2048, 1024, 2), (200, 4096, 11008, 0),
                                       (257, 1024, 1024, 0), (300, 22016, 4096, 3)])
def test_gemm_mfma128_splitk_single_image_shapes(lib, M, N, K, epi):
    """the ViT / resampler GEMMs of ONE image (M = 257 / 321) and a 200-token LLaMA prefill: few 128 x 128 tiles, long K -> K slices
    + reduce launch (launch_mfma); same function as the unsplit kernel, deterministic, fp32 output included"""
    g = torch.Generator().manual_seed(M + N + K + epi)
    a = bf16r(torch.randn(M, K, generator=g))
    w = bf16r(torch.randn(N, K, generator=g) * 0.05)
    bias = bf16r(torch.randn(N, generator=g) * 0.1)
    n_out = N // 2 if epi == 3 else N
    res = bf16r(torch.randn(M, n_out, generator=g))
    ref = _gemm_ref(a, w, bias, epi, res)
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
    wp = _pack(w)
    outs = [lib.gemm(a.to(DEV, torch.bfloat16), wp, N, bias=bias.to(DEV), residual=res.to(DEV, torch.bfloat16), epilogue=epi, force_kernel=1,
                     splitk_ws=ws) for _ in range(2)]
    _cmp(f"gemm_mfma128_splitk[{M}x{N}x{K},epi{epi}]", outs[0], ref, atol=3e-3 if K > 8192 else 2e-3, rtol=8e-3)
    assert torch.equal(outs[0], outs[1])
    f32 = lib.gemm(a.to(DEV, torch.bfloat16), wp, N, bias=bias.to(DEV), epilogue=epi, force_kernel=1, splitk_ws=ws, out_f32=True)
    _cmp(f"gemm_mfma128_splitk_f32[{M}x{N}x{K},epi{epi}]", f32, _gemm_ref(a, w, bias, epi, None), atol=2e-4, rtol=2e-5)


@pytest.mark.parametrize("M,N,K,epi,fk", [(257 * 3, 768, 192, 0, 4), (257 * 5, 1000, 1024, 1, 4), (257 * 2, 3072, 1024, 3, 4), (257 * 9, 1024, 4096, 2, 5),
                                          (257 * 64, 1024, 1024, 0, 0), (257 * 64, 4096, 1024, 1, 0), (257 * 32, 1024, 4096, 0, 0), (257 * 7, 2048, 512, 0, 0),
                                          # a LLaMA prefill whose B * T happens to be a multiple of 257 takes the same tiles: SwiGLU / in-place residual at those widths
                                          (257 * 16, 22016, 4096, 3, 4), (257 * 16, 4096, 11008, 0, 4), (257 * 16, 22016, 4096, 3, 0)])
def test_gemm_tile257(lib, M, N, K, epi, fk):
    """M = B * 257 (whole ViT sequences): the 256 x 256 kernel runs 257-row tiles -- the 257th row as a 17th MFMA strip whose other 15
    rows (the next tile's first rows) are computed and NOT stored -- instead of 256-row rounds + a tail launch.  Every row of every tile
    against the fp32 reference, bias / activation / in-place residual / SwiGLU / ragged N / fp32 output; fk = 0 is the engine's dispatch."""
    g = torch.Generator().manual_seed(M + N + K + epi)
    a = bf16r(torch.randn(M, K, generator=g))
    w = bf16r(torch.randn(N, K, generator=g) * 0.05)
    bias = bf16r(torch.randn(N, generator=g) * 0.1)
    n_out = N // 2 if epi == 3 else N
    res = bf16r(torch.randn(M, n_out, generator=g))
    ref = _gemm_ref(a, w, bias, epi, res)
    wp = _pack(w)
    xd = res.to(DEV, torch.bfloat16)
    lib.gemm(a.to(DEV, torch.bfloat16), wp, N, bias=bias.to(DEV), residual=xd, out=xd, epilogue=epi, force_kernel=fk)
    _cmp(f"gemm_tile257[{M}x{N}x{K},epi{epi},k{fk}]", xd, ref, atol=3e-3 if K > 2048 else 2e-3, rtol=8e-3)
    f32 = lib.gemm(a.to(DEV, torch.bfloat16), wp, N, bias=bias.to(DEV), epilogue=epi, force_kernel=fk, out_f32=True)
    _cmp(f"gemm_tile257_f32[{M}x{N}x{K},epi{epi},k{fk}]", f32, _gemm_ref(a, w, bias, epi, None), atol=2e-4, rtol=2e-5)


RING_SHAPES = [
    # M, N, K, epi -- the four LLaMA-7B decode GEMMs at 129 - 256 rows (tiles 128 x 96, 64 x 64, 256 x 96, 64 x 64) + lm_head, then ragged / tiny shapes
    (256, 12288, 4096, 0), (256, 4096, 4096, 0), (256, 22016, 4096, 3), (256, 4096, 11008, 0), (200, 12288, 4096, 0), (129, 4096, 4096, 0),
    (193, 22016, 4096, 3), (255, 4096, 11008, 0), (256, 49958, 4096, 0),
    (130, 200, 192, 0), (256, 96, 64, 0), (131, 1000, 128, 3), (250, 333, 704, 0), (256, 5120, 13824, 0), (144, 64, 4096, 0),
]


@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
@pytest.mark.parametrize("M,N,K,epi", RING_SHAPES)
def test_gemm_ring(lib, M, N, K, epi, cfg):
    """kernel 11 (gemm_ring.hip: full-K tiles fed by an LDS-DMA ring, 129 - 256 rows): every row and column against the fp32 reference with bias,
    in-place residual, SwiGLU, ragged M / N, fp32 output; cfg = 0 is the kernel's own tile choice (force_kernel 11), 1 / 2 / 3 force the 256 x 96 /
    128 x 96 / 64 x 64 tile (force_kernel 12 / 13 / 14)."""
    if cfg > 1 and epi == 3:
        pytest.skip("SwiGLU exists on the 256 x 96 tile only")
    if cfg and (N > 30000 or (M, N, K) in ((200, 12288, 4096), (193, 22016, 4096), (255, 4096, 11008))):
        pytest.skip("forced tiles: one instance of each LLaMA shape is enough")
    if epi == 3:
        N = (N + 31) // 32 * 32
    g = torch.Generator().manual_seed(M * 1000 + N + K + epi)
    a = bf16r(torch.randn(M, K, generator=g))
    w = bf16r(torch.randn(N, K, generator=g) * 0.05)
    bias = bf16r(torch.randn(N, generator=g) * 0.1)
    n_out = N // 2 if epi == 3 else N
    res = bf16r(torch.randn(M, n_out, generator=g))
    ref = _gemm_ref(a, w, bias, epi, res)
    wp = _pack(w)
    fk = 11 + cfg
    xd = res.to(DEV, torch.bfloat16)
    lib.gemm(a.to(DEV, torch.bfloat16), wp, N, bias=bias.to(DEV), residual=xd, out=xd, epilogue=epi, force_kernel=fk)
    _cmp(f"gemm_ring[{M}x{N}x{K},epi{epi},cfg{cfg}]", xd, ref, atol=3e-3 if K > 2048 else 2e-3, rtol=8e-3)
    again = res.to(DEV, torch.bfloat16)
    lib.gemm(a.to(DEV, torch.bfloat16), wp, N, bias=bias.to(DEV), residual=again, out=again, epilogue=epi, force_kernel=fk)
    assert torch.equal(xd, again)                    # fixed summation order: the same bits on every run
    if epi == 0:
        f32 = lib.gemm(a.to(DEV, torch.bfloat16), wp, N, bias=bias.to(DEV), force_kernel=fk, out_f32=True)
        _cmp(f"gemm_ring_f32[{M}x{N}x{K},cfg{cfg}]", f32, _gemm_ref(a, w, bias, 0, None), atol=3e-4, rtol=2e-5)
    if cfg == 0:                                     # the dispatch picks this kernel for 129 - 256 rows
        auto = lib.gemm(a.to(DEV, torch.bfloat16), wp, N, bias=bias.to(DEV), residual=res.to(DEV, torch.bfloat16), epilogue=epi)
        assert torch.equal(auto, xd)


@pytest.mark.parametrize("M,N,K,epi,f32out", [(256, 12288, 4096, 0, False), (256, 4096, 4096, 0, False), (256, 22016, 4096, 3, False), (256, 4096, 11008, 0, False),
                                              (160, 22016, 4096, 3, False), (129, 4096, 11008, 0, False), (256, 49958, 4096, 0, True), (130, 200, 192, 0, False),
                                              (250, 352, 704, 3, False)])
def test_gemm_ring_fp8_weights(lib, M, N, K, epi, f32out):
    """W8A16 at 129 - 256 rows (BASELINE configs[4], decode batches of its N = 1 leg): the ring kernel stages the e4m3 rows as they are and widens them
    in registers -- EXACTLY the function of the dequantised weights (e4m3 -> bf16 is exact, fp32 accumulate, row scale in the epilogue), the same
    function the M = 1 GEMV and the M <= 128 panel kernel compute"""
    from visualcla.weights import quantize_fp8_rows, dequantize_fp8_rows
    g = torch.Generator().manual_seed(M + N + K + epi + 3)
    a = bf16r(torch.randn(M, K, generator=g))
    wp = _pack(bf16r(torch.randn(N, K, generator=g) * 0.05))
    q, sc = quantize_fp8_rows(wp)
    wdq = dequantize_fp8_rows(q, sc)[:N].cpu()
    n_out = N // 2 if epi == 3 else N
    res = None if f32out else bf16r(torch.randn(M, n_out, generator=g))
    ref = _gemm_ref(a, wdq, None, epi, res)
    got = lib.gemm(a.to(DEV, torch.bfloat16), wp, N, residual=None if res is None else res.to(DEV, torch.bfloat16), epilogue=epi, out_f32=f32out,
                   w_q8=q, w_scale=sc, force_kernel=11)
    _cmp(f"gemm_ring_fp8[{M}x{N}x{K},epi{epi}]", got, ref, atol=3e-4 if f32out else 4e-3, rtol=2e-5 if f32out else 8e-3)


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("M,N,K,epi,fk", [(256, 4096, 4096, 0, 11), (256, 22016, 4096, 3, 11), (200, 12288, 4096, 0, 11), (255, 4096, 11008, 0, 11), (131, 1000, 128, 3, 12),
                                          (250, 333, 704, 0, 13), (144, 64, 4096, 0, 14), (130, 200, 192, 0, 11)])
def test_gemm_ring_slab_major_operands(lib, M, N, K, epi, fk, fp8):
    """slab-major A / W / W_q8 ([K/64][rows][64]: the K slab of 8 rows is one contiguous KiB -- the DMA-friendly layout, vcla_gemm_args.A_slab):
    bit-identical to the same kernel on the row-major operands (same LDS image, same arithmetic), ragged row counts included"""
    from visualcla.weights import to_slab_major, from_slab_major, quantize_fp8_rows
    if epi == 3:
        N = (N + 31) // 32 * 32
    g = torch.Generator().manual_seed(M + N + K + epi)
    a = bf16r(torch.randn(M, K, generator=g)).to(DEV, torch.bfloat16)
    wp = _pack(bf16r(torch.randn(N, K, generator=g) * 0.05))
    n_out = N // 2 if epi == 3 else N
    res = bf16r(torch.randn(M, n_out, generator=g)).to(DEV, torch.bfloat16)
    a_rows = (M + 7) // 8 * 8 + 8                               # a_slab_rows > M: rows beyond M are never read
    a_pad = torch.zeros(a_rows, K, dtype=torch.bfloat16, device=DEV)
    a_pad[:M] = a
    a_sl = to_slab_major(a_pad)
    assert torch.equal(from_slab_major(a_sl)[:M], a)
    if fp8:
        q, sc = quantize_fp8_rows(wp)
        base = lib.gemm(a, wp, N, residual=res, epilogue=epi, force_kernel=fk, w_q8=q, w_scale=sc)
        got = lib.gemm(None, wp, N, residual=res, epilogue=epi, force_kernel=fk, w_q8_slab=to_slab_major(q), w_scale=sc, a_slab=a_sl, m=M)
        mixed = lib.gemm(a, wp, N, residual=res, epilogue=epi, force_kernel=fk, w_q8_slab=to_slab_major(q), w_scale=sc)
    else:
        base = lib.gemm(a, wp, N, residual=res, epilogue=epi, force_kernel=fk)
        got = lib.gemm(None, wp, N, residual=res, epilogue=epi, force_kernel=fk, w_slab=to_slab_major(wp), a_slab=a_sl, m=M)
        mixed = lib.gemm(None, wp, N, residual=res, epilogue=epi, force_kernel=fk, a_slab=a_sl, m=M)
    assert torch.equal(got, base) and torch.equal(mixed, base)


@pytest.mark.parametrize("M,N,K,epi,fk", [(256, 4096, 4096, 0, 11), (256, 22016, 4096, 3, 11), (200, 12288, 4096, 0, 11), (255, 4096, 11008, 0, 11), (131, 1024, 128, 3, 12),
                                          (250, 384, 704, 0, 13), (144, 128, 4096, 0, 14), (130, 256, 192, 0, 11)])
def test_gemm_ring_fragment_major_weights(lib, M, N, K, epi, fk):
    """the ring kernel with the weight pieces taken from the fragment-major twin (W_frag: one contiguous KiB per 16 x 32 fragment, lane-ordered in LDS):
    bit-identical to the row-major pieces (same fragments, same MFMA order), every tile, ragged M"""
    from visualcla.weights import to_fragment_major
    g = torch.Generator().manual_seed(M + N + K + epi + 1)
    a = bf16r(torch.randn(M, K, generator=g)).to(DEV, torch.bfloat16)
    wp = _pack(bf16r(torch.randn(N, K, generator=g) * 0.05))
    n_out = N // 2 if epi == 3 else N
    res = bf16r(torch.randn(M, n_out, generator=g)).to(DEV, torch.bfloat16)
    import os as _os
    base = lib.gemm(a, wp, N, residual=res, epilogue=epi, force_kernel=fk)
    got = lib.gemm(a, wp, N, residual=res, epilogue=epi, force_kernel=fk, w_frag=to_fragment_major(wp))
    assert torch.equal(got, base)


@pytest.mark.parametrize("kernel", ["mfma", "gemv"])
def test_gemm_f32_output_and_identity(lib, kernel):
    """A = I (asymmetric W) catches operand/row-column swaps; fp32 output keeps the full accumulator."""
    M = 8 if kernel == "gemv" else 192
    K, N = 192, 333
    a = torch.zeros(M, K)
    a[torch.arange(M), torch.arange(M) % K] = 1.0
    g = torch.Generator().manual_seed(5)
    w = bf16r(torch.randn(N, K, generator=g))
    ref = a @ w.t()
    got = lib.gemm(a.to(DEV, torch.bfloat16), _pack(w), N, out_f32=True, force_kernel=1 if kernel == "mfma" else 2)
    assert got.dtype == torch.float32
    _cmp(f"gemm_identity[{kernel}]", got, ref, atol=0.0)     # exact: a single non-zero product per output


def test_gemm_row_remap(lib):
    # resampler K/V assembly: rows of two GEMMs interleave into one [B, Q+N, 2D] buffer
    B, Q, Nn, D = 3, 16, 50, 128
    g = torch.Generator().manual_seed(9)
    lat, img = bf16r(torch.randn(B * Q, D, generator=g)), bf16r(torch.randn(B * Nn, D, generator=g))
    w = bf16r(torch.randn(2 * D, D, generator=g) * 0.05)
    ref = torch.cat([(lat @ w.t()).view(B, Q, 2 * D), (img @ w.t()).view(B, Nn, 2 * D)], dim=1)
    for dtype, fk in ((torch.bfloat16, 1), (torch.float32, 3)):
        out = torch.zeros(B * (Q + Nn), 2 * D, dtype=dtype, device=DEV)
        wp = _pack(w)
        lib.gemm(lat.to(DEV, dtype), wp, 2 * D, out=out, force_kernel=fk, group_rows=Q, group_stride=Q + Nn, row_offset=0)
        lib.gemm(img.to(DEV, dtype), wp, 2 * D, out=out, force_kernel=fk, group_rows=Nn, group_stride=Q + Nn, row_offset=Q)
        _cmp(f"gemm_remap[{dtype}]", out.view(B, Q + Nn, 2 * D), ref, atol=1e-4 if dtype == torch.float32 else 2e-3, rtol=8e-3)


def test_gemm_in_place_residual(lib):
    M, N, K = 200, 256, 128
    g = torch.Generator().manual_seed(11)
    a, w, x = bf16r(torch.randn(M, K, generator=g)), bf16r(torch.randn(N, K, generator=g) * 0.05), bf16r(torch.randn(M, N, generator=g))
    ref = x + a @ w.t()
    xd = x.to(DEV, torch.bfloat16)
    lib.gemm(a.to(DEV, torch.bfloat16), _pack(w), N, residual=xd, out=xd, force_kernel=1)
    _cmp("gemm_inplace_residual", xd, ref, atol=2e-3, rtol=8e-3)
    xd = x.to(DEV, torch.bfloat16)
    lib.gemm(a.to(DEV, torch.bfloat16), _pack(w), N, residual=xd, out=xd, force_kernel=4)
    _cmp("gemm256_inplace_residual", xd, ref, atol=2e-3, rtol=8e-3)


def test_gemm_full_size_shapes_vs_torch(lib):
    """BASELINE-size GEMMs (ViT fc1 at B=8, LLaMA gate/up at M=1024) against torch's own GPU matmul in fp32."""
    g = torch.Generator(device=DEV).manual_seed(3)
    for (M, N, K, epi) in [(8 * 257, 4096, 1024, 1), (1024, 22016, 4096, 3), (1024, 4096, 11008, 0), (130, 49958, 4096, 0)]:
        a = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device=DEV) * 0.02).to(torch.bfloat16)
        ref = _gemm_ref(a.float(), w.float(), None, epi, None)
        wp = torch.zeros((N + 127) // 128 * 128, K, dtype=torch.bfloat16, device=DEV)
        wp[:N] = w
        for fk in (1, 4):
            got = lib.gemm(a, wp, N, epilogue=epi, out_f32=True, force_kernel=fk)
            _cmp(f"gemm_full[k{fk},{M}x{N}x{K},epi{epi}]", got, ref, atol=3e-3, rtol=2e-3)


def test_gemm_error_conventions(lib):
    a = torch.zeros(4, 100, dtype=torch.bfloat16, device=DEV)        # K not a multiple of 64
    w = torch.zeros(128, 100, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(ValueError):
        lib.gemm(a, w, 128)


# ------------------------------------------------------------------ attention
def _attn_ref(q, k, v, scale, causal, key_mask):
    B, H, Tq, D = q.shape
    Tk = k.shape[2]
    s = (q.float() @ k.float().transpose(-1, -2)) * scale
    if causal:
        qpos = torch.arange(Tq)[:, None] + (Tk - Tq)
        s = s.masked_fill(torch.arange(Tk)[None, :] > qpos, float("-inf"))
    if key_mask is not None:
        s = s.masked_fill(key_mask[:, None, None, :Tk] == 0, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return (p @ v.float()).transpose(1, 2).reshape(B, Tq, H * D)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,Tq,Tk,D,causal", [
    (2, 4, 17, 17, 32, False), (2, 3, 257, 257, 64, False), (1, 2, 64, 321, 64, False),
    (2, 4, 48, 48, 128, True), (2, 2, 1, 77, 128, True), (1, 2, 5, 130, 64, True), (1, 1, 130, 130, 128, True),
])
def test_attention_generic(lib, dtype, B, H, Tq, Tk, D, causal):
    g = torch.Generator().manual_seed(B + H + Tq + Tk + D)
    q, k, v = (bf16r(torch.randn(B, H, T, D, generator=g)) for T in (Tq, Tk, Tk))
    ref = _attn_ref(q, k, v, 1 / math.sqrt(D), causal, None)
    got = lib.attention(q.to(DEV, dtype), k.to(DEV, dtype), v.to(DEV, dtype), 1 / math.sqrt(D), causal=causal, force_kernel=1)
    # bf16: probabilities and the output are rounded to bf16 (as HF does), |v| ~ 1
    _cmp(f"attn_generic[{dtype},B{B}H{H}Tq{Tq}Tk{Tk}D{D}c{int(causal)}]", got, ref, atol=2e-5 if dtype == torch.float32 else 1.5e-2)


def test_attention_key_mask_and_strides(lib):
    # fused-qkv strides (ViT layout) + left-padding mask
    B, H, T, D = 2, 4, 40, 64
    g = torch.Generator().manual_seed(1)
    qkv = bf16r(torch.randn(B, T, 3 * H * D, generator=g))
    km = torch.ones(B, T, dtype=torch.int32)
    km[1, :7] = 0
    q, k, v = (qkv[..., i * H * D:(i + 1) * H * D].view(B, T, H, D).transpose(1, 2) for i in range(3))
    ref = _attn_ref(q, k, v, 0.125, True, km)
    qd = qkv.to(DEV)
    qv, kv, vv = (qd[..., i * H * D:(i + 1) * H * D].view(B, T, H, D).transpose(1, 2) for i in range(3))
    got = lib.attention(qv, kv, vv, 0.125, causal=True, key_mask=km.to(DEV), force_kernel=1)
    # rows whose every visible key is masked are don't-care (HF gives them uniform garbage); compare the rest
    valid = torch.ones(B, T, dtype=torch.bool)
    valid[1, :7] = False
    _cmp("attn_mask_strided", got[valid.to(DEV)], ref[valid], atol=2e-5)


# ------------------------------------------------------------------ embed / rope / argmax / im2col
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embed_splice(lib, dtype):
    B, T, Q, D, V = 3, 20, 5, 64, 97
    g = torch.Generator().manual_seed(2)
    table = bf16r(torch.randn(V, D, generator=g))
    img = bf16r(torch.randn(B, Q, D, generator=g))
    ids = torch.randint(0, V, (B, T), generator=g)
    pos = torch.tensor([3, -1, 14], dtype=torch.int32)
    ref = table[ids].clone()
    for b in range(B):
        if pos[b] >= 0:
            ref[b, pos[b] + 1:pos[b] + 1 + Q] = img[b]
    out = torch.empty(B, T, D, dtype=dtype, device=DEV)
    L = lib.load()
    ids_d, tab_d, img_d, pos_d = ids.to(DEV), table.to(DEV, torch.bfloat16), img.to(DEV, dtype), pos.to(DEV)
    lib.check(L.vcla_embed_splice(ids_d.data_ptr(), tab_d.data_ptr(), img_d.data_ptr(), pos_d.data_ptr(), out.data_ptr(),
                                  B, T, Q, D, V, lib.dtype_code(dtype), lib.stream_ptr()))
    torch.cuda.synchronize()
    _cmp(f"embed_splice[{dtype}]", out, ref, atol=0.0)       # pure data movement: bit exact


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rope_kv_append(lib, dtype):
    B, T, H, d, ctx, pos0 = 2, 5, 3, 64, 16, 4
    g = torch.Generator().manual_seed(4)
    qkv = bf16r(torch.randn(B, T, 3, H, d, generator=g))
    from visualcla.weights import rope_tables
    cos, sin = rope_tables(32, d, 10000.0)
    positions = torch.arange(pos0, pos0 + T)
    c, s = O.llama_rope_tables(positions, d, 10000.0, torch.float32)
    if dtype == torch.bfloat16:
        c, s = bf16r(c), bf16r(s)
    qr = O.apply_rope(qkv[:, :, 0].transpose(1, 2), c, s)          # [B,H,T,d]
    kr = O.apply_rope(qkv[:, :, 1].transpose(1, 2), c, s)
    buf = qkv.reshape(B * T, 3 * H * d).to(DEV, dtype).contiguous()
    kc = torch.zeros(B, H, ctx, d, dtype=dtype, device=DEV)
    vc = torch.zeros(B, H, ctx, d, dtype=dtype, device=DEV)
    pos_dev = torch.tensor([1], dtype=torch.int32, device=DEV)
    L = lib.load()
    cos_d, sin_d = cos.to(DEV), sin.to(DEV)
    lib.check(L.vcla_rope_kv_append(buf.data_ptr(), kc.data_ptr(), vc.data_ptr(), cos_d.data_ptr(), sin_d.data_ptr(),
                                    B, T, H, d, ctx, pos0 - 1, pos_dev.data_ptr(), lib.dtype_code(dtype), lib.stream_ptr()))
    torch.cuda.synchronize()
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    _cmp(f"rope_q[{dtype}]", buf.view(B, T, 3, H, d)[:, :, 0].transpose(1, 2), qr, atol=tol)
    _cmp(f"rope_k_cache[{dtype}]", kc[:, :, pos0:pos0 + T], kr, atol=tol)
    _cmp(f"v_cache[{dtype}]", vc[:, :, pos0:pos0 + T], qkv[:, :, 2].transpose(1, 2), atol=0.0)
    assert float(kc[:, :, :pos0].abs().max()) == 0.0 and float(kc[:, :, pos0 + T:].abs().max()) == 0.0


def test_argmax_first_max(lib):
    g = torch.Generator().manual_seed(6)
    x = torch.randn(5, 49958, generator=g)
    x[2, 100] = x[2, 40000] = 99.0          # tie -> lowest index
    x[3, -1] = 123.0
    got = lib.argmax(x.to(DEV)).cpu()
    assert torch.equal(got, x.argmax(dim=-1)), (got, x.argmax(dim=-1))
    assert got[2] == 100 and got[3] == 49957


@pytest.mark.parametrize("B,T,V", [(2, 12, 320), (3, 7, 49958), (1, 1, 1000), (4, 33, 1000)])
def test_causal_lm_loss_matches_the_hf_formula(lib, B, T, V):
    """vcla_causal_lm_loss == hf ForCausalLMLoss (oracle.causal_lm_loss restates it): shifted targets, ignore_index rows skipped, mean; the
    same bits on repeated runs; nan when nothing is supervised (torch's mean over zero targets)"""
    g = torch.Generator().manual_seed(B + T + V)
    logits = torch.randn(B, T, V, generator=g) * 3.0
    labels = torch.randint(0, V, (B, T), generator=g)
    labels[:, 0] = -100
    if T > 3:
        labels[0, 2] = -100
    want = O.causal_lm_loss(logits, labels) if T > 1 else torch.tensor(float("nan"))
    lg_d = logits.to(DEV)
    got = lib.causal_lm_loss(lg_d, labels.to(DEV))
    again = lib.causal_lm_loss(lg_d, labels.to(DEV))
    torch.cuda.synchronize()
    if T > 1:
        assert abs(float(got) - float(want)) <= 2e-5 * max(1.0, abs(float(want))), (float(got), float(want))
        assert float(got) == float(again)
    else:
        assert math.isnan(float(got))
    none = lib.causal_lm_loss(lg_d, torch.full((B, T), -100, dtype=torch.int64, device=DEV))
    assert math.isnan(float(none))


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_causal_lm_loss_propagates_non_finite_logits(lib, bad):
    """A NaN or +inf logit in a SUPERVISED row makes the loss nan, as torch's cross_entropy / HF's ForCausalLMLoss do (round 4 clamped such a
    row to 0 = "perfect prediction"); in an ignored row it changes nothing; a -inf logit is an ordinary masked class (finite loss unless it is
    the target, then +inf)."""
    B, T, V = 2, 6, 300
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(B, T, V, generator=g) * 2.0
    labels = torch.randint(0, V, (B, T), generator=g)
    labels[:, 0] = -100
    labels[1, 4] = -100                                   # -> row (1, 3) has no target
    clean = float(lib.causal_lm_loss(logits.to(DEV), labels.to(DEV)))
    poisoned = logits.clone()
    poisoned[0, 2, 17] = bad                              # row (0, 2) is scored against labels[0, 3]
    want = torch.nn.functional.cross_entropy(poisoned[:, :-1].reshape(-1, V), labels[:, 1:].reshape(-1), ignore_index=-100)
    got = float(lib.causal_lm_loss(poisoned.to(DEV), labels.to(DEV)))
    assert math.isnan(float(want)) and math.isnan(got), (float(want), got)
    ignored = logits.clone()
    ignored[1, 3, 5] = bad                                # a row without a target: not read at all
    ignored[0, T - 1, 9] = bad                            # the last position never has one
    assert float(lib.causal_lm_loss(ignored.to(DEV), labels.to(DEV))) == clean
    masked = logits.clone()
    masked[0, 2, (int(labels[0, 3]) + 1) % V] = float("-inf")
    want = float(torch.nn.functional.cross_entropy(masked[:, :-1].reshape(-1, V), labels[:, 1:].reshape(-1), ignore_index=-100))
    got = float(lib.causal_lm_loss(masked.to(DEV), labels.to(DEV)))
    assert math.isfinite(got) and abs(got - want) <= 2e-5 * max(1.0, abs(want))
    masked[0, 2, int(labels[0, 3])] = float("-inf")
    assert float(lib.causal_lm_loss(masked.to(DEV), labels.to(DEV))) == float("inf")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_patch_embed_pipeline(lib, dtype):
    """im2col + GEMM + class/pos assembly + pre-LN == oracle.clip_embeddings."""
    cfg = O.cfg_small().vision
    W = O.make_weights(O.cfg_small(), seed=0)
    px = bf16r(torch.randn(2, 3, cfg.image_size, cfg.image_size, generator=torch.Generator().manual_seed(8)))
    ref = O.clip_embeddings(px, W, cfg)
    p = "vision_model.vision_model."
    D, P = cfg.hidden_size, cfg.patch_size
    kreal, kpad = 3 * P * P, (3 * P * P + 63) // 64 * 64
    L = lib.load()
    np_ = cfg.num_patches
    patches = torch.empty(2 * np_, kpad, dtype=dtype, device=DEV)
    px_d = px.to(DEV, dtype).contiguous()
    lib.check(L.vcla_im2col(px_d.data_ptr(), patches.data_ptr(), 2, 3, cfg.image_size, cfg.image_size,
                            P, kpad, lib.dtype_code(dtype), lib.stream_ptr()))
    wp = torch.nn.functional.pad(W[p + "embeddings.patch_embedding.weight"].reshape(D, kreal), (0, kpad - kreal))
    pe = lib.gemm(patches, _pack(wp), D)
    out = torch.empty(2 * (np_ + 1), D, dtype=dtype, device=DEV)
    f = lambda n: W[p + n].float().to(DEV).contiguous()
    cls, pos, gm, bt = f("embeddings.class_embedding"), f("embeddings.position_embedding.weight"), f("pre_layrnorm.weight"), f("pre_layrnorm.bias")
    lib.check(L.vcla_vit_assemble(pe.data_ptr(), cls.data_ptr(), pos.data_ptr(), gm.data_ptr(), bt.data_ptr(), out.data_ptr(),
                                  2, np_, D, cfg.layer_norm_eps, lib.dtype_code(dtype), lib.stream_ptr()))
    _cmp(f"patch_embed[{dtype}]", out.view(2, np_ + 1, D), ref, atol=1e-4 if dtype == torch.float32 else 6e-2)


# ------------------------------------------------------------------ fused RMSNorm + GEMV (decode path)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,epi", [(1, 4096, 512, 0), (3, 320, 256, 0), (8, 2048, 1408, 3), (1, 12288, 4096, 0), (2, 22016, 4096, 3)])
def test_gemv_fused_rmsnorm(lib, dtype, M, N, K, epi):
    g = torch.Generator().manual_seed(M + N + K)
    x = bf16r(torch.randn(M, K, generator=g) * 1.3)
    gamma = bf16r(1 + 0.1 * torch.randn(K, generator=g))
    w = bf16r(torch.randn(N, K, generator=g) * 0.03)
    res = bf16r(torch.randn(M, N // 2 if epi == 3 else N, generator=g))
    h = O.llama_rmsnorm(x, gamma, 1e-6)                      # fp32 reference of the norm
    ref = _gemm_ref(h, w, None, epi, res)
    got = lib.gemm(x.to(DEV, dtype), _pack(w), N, residual=res.to(DEV, dtype), epilogue=epi, norm_gamma=gamma.to(DEV),
                   norm_eps=1e-6)
    _cmp(f"gemv_fused_norm[{dtype},{M}x{N}x{K},epi{epi}]", got, ref, atol=2e-4 if dtype == torch.float32 else 3e-3,
         rtol=1e-5 if dtype == torch.float32 else 8e-3)


# ------------------------------------------------------------------ fused decode attention (RoPE + append + attend)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,d,pos", [(2, 4, 128, 37), (1, 3, 64, 200), (3, 2, 32, 5), (1, 2, 128, 0), (2, 2, 128, 700),
                                       # B*H >= 512 selects the row-cooperative score loop (batch decode)
                                       (32, 16, 128, 190), (64, 8, 64, 3), (16, 32, 32, 65), (16, 32, 128, 0),
                                       # B*H >= 1024 at d = 128: 2-wave workgroups (one round of workgroups at B = 64)
                                       (64, 32, 128, 190), (40, 32, 128, 70), (33, 32, 128, 0), (64, 16, 128, 333)])
def test_attn_decode_fused(lib, dtype, B, H, d, pos):
    from visualcla.weights import rope_tables
    ctx = max(64, (pos + 64) // 64 * 64)
    g = torch.Generator().manual_seed(B + H + d + pos)
    qkv = bf16r(torch.randn(B, 3, H, d, generator=g))
    kc = bf16r(torch.randn(B, H, ctx, d, generator=g))
    vc = bf16r(torch.randn(B, H, ctx, d, generator=g))
    km = torch.ones(B, ctx, dtype=torch.int32)
    if pos > 3:
        km[0, 1:3] = 0
    cos, sin = rope_tables(1024, d, 10000.0)
    c, s = O.llama_rope_tables(torch.tensor([pos]), d, 10000.0, torch.float32)
    rnd = (lambda t: t) if dtype == torch.float32 else bf16r
    c, s = rnd(c), rnd(s)
    q = rnd(O.apply_rope(qkv[:, 0][:, :, None, :], c, s))          # [B,H,1,d]
    kn = rnd(O.apply_rope(qkv[:, 1][:, :, None, :], c, s))
    K = torch.cat([kc[:, :, :pos], kn], dim=2)
    V = torch.cat([vc[:, :, :pos], qkv[:, 2][:, :, None, :]], dim=2)
    ref = _attn_ref(q, K, V, 1 / math.sqrt(d), True, km)
    kc_d, vc_d = kc.to(DEV, dtype), vc.to(DEV, dtype)
    qkv_d = qkv.reshape(B, 3 * H * d).to(DEV, dtype).contiguous()
    out = torch.empty(B, H * d, dtype=dtype, device=DEV)
    dev_part = min(2, pos)               # position = pos0 + *pos_dev, split between the host and the device counter
    pos_dev = torch.tensor([dev_part], dtype=torch.int32, device=DEV)
    cos_d, sin_d, km_d = cos.to(DEV), sin.to(DEV), km.to(DEV)
    L = lib.load()
    lib.check(L.vcla_attn_decode_fused(qkv_d.data_ptr(), kc_d.data_ptr(), vc_d.data_ptr(), cos_d.data_ptr(), sin_d.data_ptr(),
                                       out.data_ptr(), B, H, d, ctx, pos - dev_part, pos_dev.data_ptr(), km_d.data_ptr(), ctx,
                                       1 / math.sqrt(d), lib.dtype_code(dtype), 0, lib.stream_ptr()))
    torch.cuda.synchronize()
    _cmp(f"attn_decode_fused[{dtype},B{B}H{H}d{d}pos{pos}]", out.view(B, 1, H * d), ref, atol=3e-5 if dtype == torch.float32 else 1.5e-2)
    # the cache must now hold the roped key / raw value at `pos`, everything else untouched
    _cmp("decode_k_append", kc_d[:, :, pos], kn[:, :, 0], atol=1e-6 if dtype == torch.float32 else 2e-2)
    _cmp("decode_v_append", vc_d[:, :, pos], qkv[:, 2], atol=0.0)
    assert torch.equal(kc_d[:, :, :pos].float().cpu(), kc[:, :, :pos]) and torch.equal(kc_d[:, :, pos + 1:].float().cpu(), kc[:, :, pos + 1:])


KV_FP8 = 0x100     # include/visualcla_hip.h VCLA_KV_FP8


def _fp8rt(t):
    """e4m3fn rounding (what the cache stores) of a float tensor, and the bytes"""
    q = t.float().to(torch.float8_e4m3fn)
    return q.float(), q.view(torch.uint8)


@pytest.mark.parametrize("B,H,d,pos,frag", [(1, 4, 128, 0, 0), (2, 3, 128, 37, 0), (64, 32, 128, 191, 1), (3, 2, 64, 70, 0), (16, 32, 128, 254, 1), (2, 2, 128, 300, 0)])
def test_attn_decode_fused_fp8_cache(lib, B, H, d, pos, frag):
    """VCLA_KV_FP8: the cache rows are e4m3 bytes (16 elements per 16-byte lane load, converted in registers).  Reference = fp32 attention
    over the DEQUANTISED cache + the e4m3 rounding of the new token's roped key / value -- the only rounding the kernel adds on top is the
    bf16 output; the appended bytes must equal torch's own float8_e4m3fn conversion."""
    from visualcla.weights import rope_tables
    ctx = max(64, (pos + 64) // 64 * 64)
    g = torch.Generator().manual_seed(B + H + d + pos)
    qkv = bf16r(torch.randn(B, 3, H, d, generator=g))
    kc, kc_b = _fp8rt(torch.randn(B, H, ctx, d, generator=g))
    vc, vc_b = _fp8rt(torch.randn(B, H, ctx, d, generator=g))
    km = torch.ones(B, ctx, dtype=torch.int32)
    if pos > 3:
        km[0, 1:3] = 0
    cos, sin = rope_tables(1024, d, 10000.0)
    c, s = O.llama_rope_tables(torch.tensor([pos]), d, 10000.0, torch.float32)
    c, s = bf16r(c), bf16r(s)
    q = bf16r(O.apply_rope(qkv[:, 0][:, :, None, :], c, s))          # [B,H,1,d]
    kn, kn_b = _fp8rt(bf16r(O.apply_rope(qkv[:, 1][:, :, None, :], c, s)))
    vn, vn_b = _fp8rt(qkv[:, 2][:, :, None, :])
    K = torch.cat([kc[:, :, :pos], kn], dim=2)
    V = torch.cat([vc[:, :, :pos], vn], dim=2)
    ref = _attn_ref(q, K, V, 1 / math.sqrt(d), True, km)
    kc_d, vc_d = kc_b.to(DEV).contiguous(), vc_b.to(DEV).contiguous()
    qkv_d = qkv.reshape(B, 3 * H * d).to(DEV, torch.bfloat16).contiguous()
    out = torch.zeros((H * d) // 32, (B + 15) // 16, 64, 8, dtype=torch.bfloat16, device=DEV) if frag else torch.empty(B, H * d, dtype=torch.bfloat16, device=DEV)
    dev_part = min(2, pos)
    pos_dev = torch.tensor([dev_part], dtype=torch.int32, device=DEV)
    cos_d, sin_d, km_d = cos.to(DEV), sin.to(DEV), km.to(DEV)
    L = lib.load()
    lib.check(L.vcla_attn_decode_fused(qkv_d.data_ptr(), kc_d.data_ptr(), vc_d.data_ptr(), cos_d.data_ptr(), sin_d.data_ptr(),
                                       out.data_ptr(), B, H, d, ctx, pos - dev_part, pos_dev.data_ptr(), km_d.data_ptr(), ctx,
                                       1 / math.sqrt(d), lib.dtype_code(torch.bfloat16) | KV_FP8, frag, lib.stream_ptr()))
    torch.cuda.synchronize()
    got = lib.from_frag(out, B) if frag else out
    _cmp(f"attn_decode_fused_fp8kv[B{B}H{H}d{d}pos{pos}]", got.reshape(B, 1, H * d), ref, atol=1.5e-2)
    assert torch.equal(kc_d[:, :, pos].cpu(), kn_b[:, :, 0]) and torch.equal(vc_d[:, :, pos].cpu(), vn_b[:, :, 0])
    assert torch.equal(kc_d[:, :, :pos].cpu(), kc_b[:, :, :pos]) and torch.equal(vc_d[:, :, pos + 1:].cpu(), vc_b[:, :, pos + 1:])
    with pytest.raises((ValueError, lib.VclaError)):     # fp32 activations have no fp8-cache form
        lib.check(L.vcla_attn_decode_fused(qkv_d.data_ptr(), kc_d.data_ptr(), vc_d.data_ptr(), cos_d.data_ptr(), sin_d.data_ptr(), out.data_ptr(), B, H, d,
                                           ctx, pos, None, None, 0, 1 / math.sqrt(d), lib.dtype_code(torch.float32) | KV_FP8, 0, lib.stream_ptr()))


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("M,N,K", [(64, 12288, 4096), (48, 1536, 1024), (64, 4096, 2048), (33, 1000, 1408)])
def test_gemm_dstream_raw_partials(lib, M, N, K, fp8):
    """ds_raw_partials: split-K without the reduce launch -- the workspace holds the RAW fp32 slices [S][M][N] (no bias, no fp8 scale, no rstd)
    for a consumer that sums them (the decode attention).  (64, 12288, 4096) = the LLaMA qkv shape: 6 tiles per pair of workgroups, the wide
    kernel; the others fall back to the 4-tile kernel."""
    from visualcla.weights import to_fragment_major, quantize_fp8_rows, to_fragment_pair_major_fp8
    g = torch.Generator().manual_seed(M + N + K)
    a = bf16r(torch.randn(M, K, generator=g))
    w = bf16r(torch.randn(N, K, generator=g) * 0.05)
    wp = _pack(w)
    kw = dict(w_frag=to_fragment_major(wp))
    scale = None
    wref = w
    if fp8:
        q, sc = quantize_fp8_rows(wp)
        wref = q[:N].view(torch.float8_e4m3fn).float().cpu()      # the raw slices are in units of the fp8 codes
        scale = sc[:N].float().cpu()
        kw = dict(w_q8_frag=to_fragment_pair_major_fp8(q), w_scale=sc)
    af = lib.to_frag(a.to(DEV, torch.bfloat16))
    ws = torch.full((32 << 18,), float("nan"), dtype=torch.float32, device=DEV)
    out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
    ssq_in = torch.ones(M, 16, dtype=torch.float32, device=DEV)          # accepted and NOT applied
    lib.gemm(None, wp, N, out=out, force_kernel=9, a_frag=af, m=M, splitk_ws=ws.view(torch.uint8), ds_splitk=2, ds_raw_partials=True, a_row_ssq=ssq_in,
             a_norm_eps=1e-6, **kw)
    torch.cuda.synchronize()
    parts = ws[: 2 * M * N].view(2, M, N).cpu()
    assert torch.isfinite(parts).all()
    ref = a @ wref.t()
    _cmp(f"gemm_dstream_raw_partials[{M}x{N}x{K},fp8={fp8}]", parts.sum(0), ref, atol=(2e-3 if not fp8 else 0.5), rtol=1e-4)
    assert float((out.float() - 7.0).abs().max()) == 0.0                 # C untouched
    if scale is not None:
        assert scale.shape[0] == N


@pytest.mark.parametrize("kv8", [False, True])
@pytest.mark.parametrize("B,H,d,pos,with_norm,with_wscale", [(33, 32, 128, 150, True, False), (64, 32, 128, 191, True, True), (40, 32, 64, 5, False, False)])
def test_attn_decode_fused_parts(lib, B, H, d, pos, with_norm, with_wscale, kv8):
    """vcla_attn_decode_fused_parts: q / k / v arrive as TWO raw fp32 K slices of the qkv projection (+ the row's 16 partial sums of squares
    for the deferred RMSNorm, + fp8 weight scales); the kernel must do exactly what reduce launch + vcla_attn_decode_fused do: same output
    bits, same appended cache rows"""
    from visualcla.weights import rope_tables
    ctx = (pos + 64) // 64 * 64
    g = torch.Generator().manual_seed(B + H + d + pos)
    HD = H * d
    p0 = torch.randn(B, 3 * HD, generator=g) * 2
    p1 = torch.randn(B, 3 * HD, generator=g) * 2
    ssq = torch.rand(B, 16, generator=g) * 300 + 10 if with_norm else None
    wsc = (torch.rand(3 * HD, generator=g) * 0.5 + 0.75) if with_wscale else None
    eps = 1e-6
    parts = torch.stack([p0, p1]).to(DEV).contiguous()
    qkv_d = parts[0] + parts[1]                   # the same fp32 arithmetic as the kernel, on the device
    if with_norm:
        qkv_d = qkv_d * torch.rsqrt(ssq.to(DEV).sum(1, keepdim=True) / HD + eps)
    if with_wscale:
        qkv_d = qkv_d * wsc.to(DEV)
    qkv_bf = qkv_d.to(torch.bfloat16).contiguous()
    mk = (lambda: torch.randn(B, H, ctx, d, generator=g).to(torch.float8_e4m3fn).view(torch.uint8).to(DEV)) if kv8 else (lambda: torch.randn(B, H, ctx, d, generator=g).to(DEV, torch.bfloat16))
    kc, vc = mk(), mk()
    kc2, vc2 = kc.clone(), vc.clone()
    cos, sin = rope_tables(1024, d, 10000.0)
    cos_d, sin_d = cos.to(DEV), sin.to(DEV)
    km = torch.ones(B, ctx, dtype=torch.int32, device=DEV)
    km[0, 1:3] = 0
    dt = lib.dtype_code(torch.bfloat16) | (KV_FP8 if kv8 else 0)
    MT = (B + 15) // 16
    out_a = torch.zeros(HD // 32, MT, 64, 8, dtype=torch.bfloat16, device=DEV)
    out_b = torch.zeros_like(out_a)
    L = lib.load()
    ssq_d = ssq.to(DEV).contiguous() if with_norm else None
    wsc_d = wsc.to(DEV).contiguous() if with_wscale else None
    lib.check(L.vcla_attn_decode_fused(qkv_bf.data_ptr(), kc.data_ptr(), vc.data_ptr(), cos_d.data_ptr(), sin_d.data_ptr(), out_a.data_ptr(), B, H, d, ctx, pos,
                                       None, km.data_ptr(), ctx, 1 / math.sqrt(d), dt, 1, lib.stream_ptr()))
    lib.check(L.vcla_attn_decode_fused_parts(parts.data_ptr(), B * 3 * HD, lib.ptr(ssq_d), lib.ptr(wsc_d), eps, kc2.data_ptr(), vc2.data_ptr(), cos_d.data_ptr(),
                                             sin_d.data_ptr(), out_b.data_ptr(), B, H, d, ctx, pos, None, km.data_ptr(), ctx, 1 / math.sqrt(d), dt, 1, lib.stream_ptr()))
    torch.cuda.synchronize()
    # the device's own fp32 product (parts sum * rstd * scale) may differ from the kernel's by an ulp before the bf16 rounding: compare with a bf16-step tolerance,
    # and demand bit equality wherever the rounded inputs agree (checked through the appended cache rows)
    same_k = torch.equal(kc2[:, :, pos], kc[:, :, pos]) and torch.equal(vc2[:, :, pos], vc[:, :, pos])
    _cmp(f"attn_decode_parts[B{B}H{H}d{d}pos{pos}kv8={kv8}]", lib.from_frag(out_b, B), lib.from_frag(out_a, B).float(), atol=0.0 if same_k else 2e-2)
    _cmp("attn_decode_parts.k_append", kc2[:, :, pos].float() if not kv8 else kc2[:, :, pos].view(torch.float8_e4m3fn).float(),
         kc[:, :, pos].float() if not kv8 else kc[:, :, pos].view(torch.float8_e4m3fn).float(), atol=0.13 if kv8 else 3.2e-2, rtol=0.13 if kv8 else 8e-3)
    assert torch.equal(kc2[:, :, :pos], kc[:, :, :pos]) and torch.equal(vc2[:, :, pos + 1:], vc[:, :, pos + 1:])


def test_rope_kv_append_fp8_cache(lib):
    """prefill side of VCLA_KV_FP8: q and k rotated IN PLACE in the qkv buffer (bf16, what the prompt's own attention reads), the cache
    receives the e4m3 bytes of the rotated k and of v"""
    from visualcla.weights import rope_tables
    B, T, H, d, ctx, pos0 = 2, 5, 3, 128, 64, 0
    g = torch.Generator().manual_seed(77)
    qkv = bf16r(torch.randn(B, T, 3, H, d, generator=g))
    cos, sin = rope_tables(256, d, 10000.0)
    c, s = O.llama_rope_tables(torch.arange(pos0, pos0 + T), d, 10000.0, torch.float32)
    c, s = bf16r(c), bf16r(s)
    qr = bf16r(O.apply_rope(qkv[:, :, 0].permute(0, 2, 1, 3), c, s))      # [B,H,T,d]
    kr = bf16r(O.apply_rope(qkv[:, :, 1].permute(0, 2, 1, 3), c, s))
    qkv_d = qkv.reshape(B * T, 3 * H * d).to(DEV, torch.bfloat16).contiguous()
    kc_d = torch.full((B, H, ctx, d), 0x55, dtype=torch.uint8, device=DEV)
    vc_d = torch.full((B, H, ctx, d), 0x55, dtype=torch.uint8, device=DEV)
    L = lib.load()
    cos_d, sin_d = cos.to(DEV), sin.to(DEV)        # (named: a temporary would be freed -- and its block reused -- before the launch)
    lib.check(L.vcla_rope_kv_append(qkv_d.data_ptr(), kc_d.data_ptr(), vc_d.data_ptr(), cos_d.data_ptr(), sin_d.data_ptr(), B, T, H, d, ctx, pos0,
                                    None, lib.dtype_code(torch.bfloat16) | KV_FP8, lib.stream_ptr()))
    torch.cuda.synchronize()
    back = qkv_d.float().cpu().view(B, T, 3, H, d)
    _cmp("rope_kv_fp8.q_in_place", back[:, :, 0].permute(0, 2, 1, 3), qr, atol=0.0)
    _cmp("rope_kv_fp8.k_in_place", back[:, :, 1].permute(0, 2, 1, 3), kr, atol=0.0)
    _cmp("rope_kv_fp8.v_untouched", back[:, :, 2], qkv[:, :, 2], atol=0.0)
    kb, vb = _fp8rt(kr)[1], _fp8rt(qkv[:, :, 2].permute(0, 2, 1, 3))[1]
    nk, nv = int((kc_d[:, :, :T].cpu() != kb).sum()), int((vc_d[:, :, :T].cpu() != vb).sum())
    assert nk == 0 and nv == 0, (nk, nv)
    assert int((kc_d[:, :, T:] != 0x55).sum()) == 0 and int((vc_d[:, :, T:] != 0x55).sum()) == 0


def test_fp8_cache_writers_saturate(lib):
    """|k| or |v| beyond e4m3's 448: v_cvt_pk_fp8_f32 does not clamp (the byte would be NaN and poison every later step of that
    (sequence, head)); the cache writers saturate first -- prefill append and decode append both store the e4m3 bytes of the value
    clamped to +-448, and the decode step's own attention output stays finite"""
    from visualcla.weights import rope_tables
    B, T, H, d, ctx = 2, 4, 2, 128, 64
    g = torch.Generator().manual_seed(5)
    qkv = bf16r(torch.randn(B, T, 3, H, d, generator=g))
    qkv[0, 1, 1, 0, 3], qkv[1, 2, 1, 1, 70] = 1000.0, -3000.0           # keys (|.| survives the rotation: > 448 after RoPE too)
    qkv[0, 0, 2, 1, 5], qkv[1, 3, 2, 0, 127] = 2048.0, -600.0           # values
    cos, sin = rope_tables(256, d, 10000.0)
    c, s = O.llama_rope_tables(torch.arange(T), d, 10000.0, torch.float32)
    kr = bf16r(O.apply_rope(qkv[:, :, 1].permute(0, 2, 1, 3), bf16r(c), bf16r(s)))
    assert float(kr.abs().max()) > 448
    sat = lambda t: _fp8rt(t.clamp(-448.0, 448.0))[1]
    qkv_d = qkv.reshape(B * T, 3 * H * d).to(DEV, torch.bfloat16).contiguous()
    kc_d = torch.zeros(B, H, ctx, d, dtype=torch.uint8, device=DEV)
    vc_d = torch.zeros(B, H, ctx, d, dtype=torch.uint8, device=DEV)
    L = lib.load()
    cos_d, sin_d = cos.to(DEV), sin.to(DEV)
    lib.check(L.vcla_rope_kv_append(qkv_d.data_ptr(), kc_d.data_ptr(), vc_d.data_ptr(), cos_d.data_ptr(), sin_d.data_ptr(), B, T, H, d, ctx, 0,
                                    None, lib.dtype_code(torch.bfloat16) | KV_FP8, lib.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(kc_d[:, :, :T].cpu(), sat(kr)) and torch.equal(vc_d[:, :, :T].cpu(), sat(qkv[:, :, 2].permute(0, 2, 1, 3)))
    assert not bool(torch.isnan(kc_d.view(torch.float8_e4m3fn).float()).any()) and not bool(torch.isnan(vc_d.view(torch.float8_e4m3fn).float()).any())
    # decode append at position T with an outlier in the new token's key and value
    step = bf16r(torch.randn(B, 3, H, d, generator=g))
    step[0, 1, 0, 9], step[1, 2, 1, 64] = -900.0, 5000.0
    c1, s1 = O.llama_rope_tables(torch.tensor([T]), d, 10000.0, torch.float32)
    kn = bf16r(O.apply_rope(step[:, 1][:, :, None, :], bf16r(c1), bf16r(s1)))
    step_d = step.reshape(B, 3 * H * d).to(DEV, torch.bfloat16).contiguous()
    out = torch.empty(B, H * d, dtype=torch.bfloat16, device=DEV)
    lib.check(L.vcla_attn_decode_fused(step_d.data_ptr(), kc_d.data_ptr(), vc_d.data_ptr(), cos_d.data_ptr(), sin_d.data_ptr(), out.data_ptr(), B, H, d, ctx,
                                       T, None, None, 0, 1 / math.sqrt(d), lib.dtype_code(torch.bfloat16) | KV_FP8, 0, lib.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(kc_d[:, :, T].cpu(), sat(kn[:, :, 0])) and torch.equal(vc_d[:, :, T].cpu(), sat(step[:, 2]))
    assert bool(torch.isfinite(out.float()).all())
    # and the step computed the function of the SATURATED cache (the new token enters its own softmax as the stored byte)
    K = torch.cat([kc_d[:, :, :T].cpu().view(torch.float8_e4m3fn).float(), sat(kn).view(torch.float8_e4m3fn).float()], dim=2)
    V = torch.cat([vc_d[:, :, :T].cpu().view(torch.float8_e4m3fn).float(), sat(step[:, 2][:, :, None, :]).view(torch.float8_e4m3fn).float()], dim=2)
    q = bf16r(O.apply_rope(step[:, 0][:, :, None, :], bf16r(c1), bf16r(s1)))
    ref = _attn_ref(q, K, V, 1 / math.sqrt(d), True, torch.ones(B, ctx, dtype=torch.int32))
    _cmp("attn_decode_fused_fp8kv_saturated", out.view(B, 1, H * d), ref, atol=4.0, rtol=2e-2)


# ------------------------------------------------------------------ MFMA flash attention (bf16)
@pytest.mark.parametrize("B,H,Tq,Tk,D,causal", [
    (2, 3, 257, 257, 64, False), (1, 2, 64, 321, 64, False), (2, 4, 48, 48, 128, True), (1, 2, 130, 130, 128, True),
    (1, 2, 128, 128, 128, True), (2, 2, 100, 228, 128, True), (1, 1, 16, 16, 64, False), (1, 2, 577, 577, 64, False),
    (1, 1, 300, 1000, 128, True),
    # bidirectional d = 64: 2 / 4 / 9 waves per workgroup by sequence length (attention_mfma.hip)
    (3, 2, 129, 129, 64, False), (1, 1, 288, 288, 64, False), (2, 2, 100, 100, 64, False), (1, 2, 289, 200, 64, False), (2, 1, 33, 64, 64, False),
])
def test_attention_mfma(lib, B, H, Tq, Tk, D, causal):
    g = torch.Generator().manual_seed(B * 3 + H + Tq + Tk + D)
    q, k, v = (bf16r(torch.randn(B, H, T, D, generator=g)) for T in (Tq, Tk, Tk))
    ref = _attn_ref(q, k, v, 1 / math.sqrt(D), causal, None)
    dt = torch.bfloat16
    got = lib.attention(q.to(DEV, dt), k.to(DEV, dt), v.to(DEV, dt), 1 / math.sqrt(D), causal=causal, force_kernel=2)
    gen = lib.attention(q.to(DEV, dt), k.to(DEV, dt), v.to(DEV, dt), 1 / math.sqrt(D), causal=causal, force_kernel=1)
    _cmp(f"attn_mfma[B{B}H{H}Tq{Tq}Tk{Tk}D{D}c{int(causal)}]", got, ref, atol=1.5e-2)
    _cmp(f"attn_mfma_vs_generic[Tq{Tq}Tk{Tk}D{D}]", got, gen.float(), atol=1.6e-2)


@pytest.mark.parametrize("B,H,T", [(64, 16, 257), (3, 2, 257), (1, 1, 257), (2, 4, 65), (40, 4, 65), (32, 16, 577), (2, 3, 577), (1, 1, 577), (9, 16, 577)])
def test_attention_vit_whole_sequence(lib, B, H, T):
    """force_kernel 3: the whole-sequence ViT self-attention (4 waves x 64 query rows on MFMA, the 257th key and the 257th query row on
    the VALU, the last row merged from per-wave partials) against the fp32 reference, through the strided fused-qkv layout the engine
    uses; and the automatic dispatch (B * H >= 128) must give the same bits as the forced kernel.  T = 577 (336-px images, round 5): the
    8-wave form with the whole 148 KB of K / V parked in LDS and the 37 q-tiles taken in two passes (attn_vit_long_kernel)."""
    D = 64
    g = torch.Generator().manual_seed(B * 7 + H + T)
    qkv = bf16r(torch.randn(B, T, 3, H, D, generator=g) * 1.5)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))          # [B, H, T, D]
    ref = _attn_ref(q, k, v, 1 / math.sqrt(D), False, None)
    qkv_d = qkv.to(DEV, torch.bfloat16).contiguous()
    L = lib.load()
    outs = []
    for fk in (3, 0, 2):
        out = torch.full((B, T, H * D), 7.0, dtype=torch.bfloat16, device=DEV)
        a = lib.AttnArgs()
        e = 2
        base = qkv_d.data_ptr()
        a.q, a.k, a.v, a.o = base, base + H * D * e, base + 2 * H * D * e, out.data_ptr()
        a.q_bs = a.k_bs = a.v_bs = T * 3 * H * D
        a.q_hs = a.k_hs = a.v_hs = D
        a.q_rs = a.k_rs = a.v_rs = 3 * H * D
        a.o_bs, a.o_hs, a.o_rs = T * H * D, D, H * D
        a.B, a.H, a.Tq, a.Tk, a.D = B, H, T, T, D
        a.scale, a.causal, a.force_kernel = 1 / math.sqrt(D), 0, fk
        lib.check(L.vcla_attention(C.byref(a), lib.dtype_code(torch.bfloat16), lib.stream_ptr()))
        torch.cuda.synchronize()
        _cmp(f"attention_vit[B{B}H{H}T{T},fk{fk}]", out, ref, atol=2e-2)
        outs.append(out)
    if B * H >= 128:
        assert torch.equal(outs[0], outs[1])          # the dispatch picked the whole-sequence kernel
    # shapes the kernel does not serve are refused, not mis-computed
    a.Tq = a.Tk = 130
    a.force_kernel = 3
    with pytest.raises((ValueError, lib.VclaError)):
        lib.check(L.vcla_attention(C.byref(a), lib.dtype_code(torch.bfloat16), lib.stream_ptr()))


def test_attention_mfma_mask_and_strides(lib):
    # fused-qkv strides + left padding, as LLaMA prefill with a padded batch would issue it
    B, H, T, D = 2, 4, 70, 128
    g = torch.Generator().manual_seed(21)
    qkv = bf16r(torch.randn(B, T, 3 * H * D, generator=g))
    km = torch.ones(B, T, dtype=torch.int32)
    km[1, :9] = 0
    q, k, v = (qkv[..., i * H * D:(i + 1) * H * D].view(B, T, H, D).transpose(1, 2) for i in range(3))
    ref = _attn_ref(q, k, v, D ** -0.5, True, km)
    qd = qkv.to(DEV, torch.bfloat16)
    qv, kv, vv = (qd[..., i * H * D:(i + 1) * H * D].view(B, T, H, D).transpose(1, 2) for i in range(3))
    got = lib.attention(qv, kv, vv, D ** -0.5, causal=True, key_mask=km.to(DEV), force_kernel=2)
    valid = torch.ones(B, T, dtype=torch.bool)
    valid[1, :9] = False
    assert torch.isfinite(got).all()
    _cmp("attn_mfma_mask_strided", got[valid.to(DEV)], ref[valid], atol=1.5e-2)


def test_attention_mfma_softmax_rescale_branch(lib):
    """online-softmax hazard: a key far above the running max in a LATE tile forces the rescale of O and l."""
    B, H, T, D = 1, 1, 256, 64
    g = torch.Generator().manual_seed(33)
    q, k, v = (bf16r(torch.randn(B, H, T, D, generator=g)) for _ in range(3))
    k[0, 0, 200] = bf16r(q[0, 0, 17] * 4.0)          # spike: q17 . k200 >> everything before it
    ref = _attn_ref(q, k, v, D ** -0.5, False, None)
    got = lib.attention(q.to(DEV, torch.bfloat16), k.to(DEV, torch.bfloat16), v.to(DEV, torch.bfloat16), D ** -0.5, force_kernel=2)
    _cmp("attn_mfma_rescale", got, ref, atol=1.5e-2)


def test_gemv1_f32_out_bias_residual(lib):
    """the tuned M = 1 kernel with every optional operand (lm_head-style fp32 output, bias, residual, fused norm)"""
    g = torch.Generator().manual_seed(77)
    N, K = 1000, 2048
    x, w = bf16r(torch.randn(1, K, generator=g)), bf16r(torch.randn(N, K, generator=g) * 0.03)
    bias, res, gamma = bf16r(torch.randn(N, generator=g)), bf16r(torch.randn(1, N, generator=g)), bf16r(1 + 0.1 * torch.randn(K, generator=g))
    ref = _gemm_ref(O.llama_rmsnorm(x, gamma, 1e-6), w, bias, 0, res)
    for fk in (2, 6):
        got = lib.gemm(x.to(DEV, torch.bfloat16), _pack(w), N, bias=bias.to(DEV), residual=res.to(DEV, torch.bfloat16), out_f32=True,
                       force_kernel=fk, norm_gamma=gamma.to(DEV), norm_eps=1e-6)
        assert got.dtype == torch.float32
        _cmp(f"gemv1_full[k{fk}]", got, ref, atol=2e-4, rtol=1e-5)


# ------------------------------------------------------------------ compile-time-K decode GEMV (gemv_decode.hip)
@pytest.mark.parametrize("N,K,epi,f32out", [(4096, 4096, 0, False), (12288, 4096, 0, False), (22016, 4096, 3, False), (4096, 11008, 0, False),
                                            (49958, 4096, 0, True), (1000, 4096, 0, False), (5120, 5120, 0, False), (27648, 5120, 3, False),
                                            (5120, 13824, 0, False), (331, 11008, 0, True)])
def test_gemv1x_llama_widths(lib, N, K, epi, f32out):
    """M = 1 at the LLaMA-7B / 13B widths (the instances of gemv1x_kernel): fused RMSNorm, bias, residual, SwiGLU, fp32 logits,
    N that is not a multiple of the 8 rows of a workgroup, the ragged last k-step of K = 11008 / 13824"""
    g = torch.Generator().manual_seed(N + K + epi)
    x = bf16r(torch.randn(1, K, generator=g) * 1.3)
    gamma = bf16r(1 + 0.1 * torch.randn(K, generator=g))
    w = bf16r(torch.randn(N, K, generator=g) * 0.03)
    n_out = N // 2 if epi == 3 else N
    bias = bf16r(torch.randn(N, generator=g) * 0.1)
    res = bf16r(torch.randn(1, n_out, generator=g))
    for use_norm in (True, False):
        h = O.llama_rmsnorm(x, gamma, 1e-6) if use_norm else x
        ref = _gemm_ref(h, w, bias, epi, res)
        got = lib.gemm(x.to(DEV, torch.bfloat16), _pack(w), N, bias=bias.to(DEV), residual=res.to(DEV, torch.bfloat16), epilogue=epi,
                       out_f32=f32out, norm_gamma=gamma.to(DEV) if use_norm else None, norm_eps=1e-6)
        _cmp(f"gemv1x[{N}x{K},epi{epi},norm{int(use_norm)}]", got, ref, atol=2e-4 if f32out else 4e-3, rtol=1e-5 if f32out else 8e-3)


@pytest.mark.parametrize("N,K,epi,f32out", [(4096, 4096, 0, False), (22016, 4096, 3, False), (4096, 11008, 0, False), (49958, 4096, 0, True),
                                            (1001, 11008, 0, False), (5120, 13824, 0, False)])
def test_gemv1x_fp8_llama_widths(lib, N, K, epi, f32out):
    """the fp8-weight instances of the same kernel: exactly the function of the dequantised weights, fp32 accumulate"""
    from visualcla.weights import quantize_fp8_rows, dequantize_fp8_rows
    g = torch.Generator().manual_seed(N + K + epi + 1)
    x = bf16r(torch.randn(1, K, generator=g))
    gamma = bf16r(1 + 0.1 * torch.randn(K, generator=g))
    wp = _pack(bf16r(torch.randn(N, K, generator=g) * 0.05))
    q, sc = quantize_fp8_rows(wp)
    wdq = dequantize_fp8_rows(q, sc)[:N].cpu()
    n_out = N // 2 if epi == 3 else N
    bias = bf16r(torch.randn(N, generator=g) * 0.1)
    res = bf16r(torch.randn(1, n_out, generator=g))
    ref = _gemm_ref(O.llama_rmsnorm(x, gamma, 1e-6), wdq, bias, epi, res)
    got = lib.gemm(x.to(DEV, torch.bfloat16), wp, N, bias=bias.to(DEV), residual=res.to(DEV, torch.bfloat16), epilogue=epi, out_f32=f32out,
                   w_q8=q, w_scale=sc, norm_gamma=gamma.to(DEV), norm_eps=1e-6)
    _cmp(f"gemv1x_fp8[{N}x{K},epi{epi}]", got, ref, atol=3e-4 if f32out else 4e-3, rtol=1e-5 if f32out else 8e-3)


# ------------------------------------------------------------------ fp8 (e4m3fn) decode weights
@pytest.mark.parametrize("M,N,K,epi", [(1, 4096, 512, 0), (1, 2048, 1408, 3), (1, 1000, 4096, 0), (2, 512, 4096, 0), (16, 4096, 1408, 3),
                                       (64, 320, 640, 0), (128, 288, 2048, 3)])
def test_gemm_fp8_weights(lib, M, N, K, epi):
    """the kernels must compute exactly the function of the DEQUANTISED weights (fp8 -> bf16 is exact), fp32 accumulate"""
    from visualcla.weights import quantize_fp8_rows, dequantize_fp8_rows, to_fragment_pair_major_fp8
    if epi == 3:
        N = (N + 31) // 32 * 32
    g = torch.Generator().manual_seed(M + N + K)
    a = bf16r(torch.randn(M, K, generator=g))
    w = bf16r(torch.randn(N, K, generator=g) * 0.05)
    gamma = bf16r(1 + 0.1 * torch.randn(K, generator=g))
    wp = _pack(w)
    q, sc = quantize_fp8_rows(wp)
    wdq = dequantize_fp8_rows(q, sc)[:N].cpu()
    assert (wdq - w).abs().max() <= 0.07 * w.abs().max()          # e4m3: 3 mantissa bits
    use_norm = M == 1
    h = O.llama_rmsnorm(a, gamma, 1e-6) if use_norm else a
    res = bf16r(torch.randn(M, N // 2 if epi == 3 else N, generator=g))
    ref = _gemm_ref(h, wdq, None, epi, res)
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
    got = lib.gemm(a.to(DEV, torch.bfloat16), wp, N, residual=res.to(DEV, torch.bfloat16), epilogue=epi, splitk_ws=ws,
                   w_q8=q, w_q8_frag=to_fragment_pair_major_fp8(q), w_scale=sc,
                   norm_gamma=gamma.to(DEV) if use_norm else None, norm_eps=1e-6)
    _cmp(f"gemm_fp8[{M}x{N}x{K},epi{epi}]", got, ref, atol=3e-3, rtol=8e-3)


def test_panel_splitk_handoff_stress(lib):
    """The split-K panel kernel hands fp32 partial tiles to its reduce kernel through a workspace that every GEMM of a decode
    step reuses.  Stress: the same workspace addresses are reused back to back by launches with DIFFERENT data (a stale slab
    read would be off by O(1)), uneven shapes, many repetitions, while a second stream keeps the memory system busy."""
    g = torch.Generator().manual_seed(123)
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
    shapes = [(64, 4096, 4096), (16, 4096, 11008), (128, 12288, 4096), (33, 2048, 1408)]
    cases = []
    for (M, N, K) in shapes:
        for rep in range(2):
            a = bf16r(torch.randn(M, K, generator=g)).to(DEV, torch.bfloat16)
            w = _pack(bf16r(torch.randn(N, K, generator=g) * 0.05))
            ref = lib.gemm(a, w, N, out_f32=True, force_kernel=1 if M > 8 else 2)      # independent kernel, no split-K
            cases.append((a, w, N, ref))
    noise = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    side = torch.cuda.Stream()
    worst = 0.0
    for it in range(40):
        with torch.cuda.stream(side):
            noise.add_(1)                                   # uneven background HBM traffic
        outs = [lib.gemm(a, w, N, out_f32=True, force_kernel=8, splitk_ws=ws) for (a, w, N, _) in cases]
        for (a, w, N, ref), got in zip(cases, outs):
            err = (got - ref).abs().max().item()
            worst = max(worst, err)
            assert err <= 2e-3 * max(1.0, ref.abs().max().item()), (it, N, err)
    torch.cuda.synchronize()
    _report(f"panel split-K stress: worst |err| {worst:.3e} over 40 x {len(cases)} launches")


@pytest.mark.parametrize("M,N,K,epi", [(4 * 257, 1024, 1024, 0), (8 * 257, 4096, 1024, 1), (1100, 512, 256, 0), (2 * 256 + 128, 768, 128, 2)])
def test_gemm_auto_dispatch_ragged_m(lib, M, N, K, epi):
    """auto dispatch peels the M % 256 tail of ragged problems (ViT rows = B*257) into a second launch: results must be
    seamless across the split, including bias / activation / in-place residual"""
    g = torch.Generator().manual_seed(M + N)
    a, w = bf16r(torch.randn(M, K, generator=g)), bf16r(torch.randn(N, K, generator=g) * 0.05)
    bias, x = bf16r(torch.randn(N, generator=g) * 0.1), bf16r(torch.randn(M, N, generator=g))
    ref = _gemm_ref(a, w, bias, epi, x)
    xd = x.to(DEV, torch.bfloat16)
    lib.gemm(a.to(DEV, torch.bfloat16), _pack(w), N, bias=bias.to(DEV), residual=xd, out=xd, epilogue=epi,
             splitk_ws=torch.zeros(32 << 20, dtype=torch.uint8, device=DEV))
    _cmp(f"gemm_auto_ragged[{M}x{N}x{K},epi{epi}]", xd, ref, atol=2e-3, rtol=8e-3)


@pytest.mark.parametrize("M,N,K,frag", [(64, 4096, 4096, True), (64, 4096, 11008, True), (33, 512, 1024, False), (7, 256, 512, True),
                                        (128, 1000, 256, True), (300, 512, 256, False), (1, 4096, 4096, False)])
def test_gemm_post_norm_is_the_rmsnorm_of_its_own_output(lib, M, N, K, frag):
    """post_norm_*: fused into the panel kernel's split-K reduction (M <= 128 with a workspace), a trailing vcla_rmsnorm
    launch otherwise -- either way C must equal the plain GEMM's C and post_norm_out must equal vcla_rmsnorm(C) bit for bit"""
    from visualcla.weights import to_fragment_major
    g = torch.Generator().manual_seed(M * 7 + N)
    a = (torch.randn(M, K, generator=g) * 0.5).to(DEV, torch.bfloat16)
    w = torch.randn(N, K, generator=g) * 0.05
    wp = torch.zeros((N + 127) // 128 * 128, K, dtype=torch.bfloat16, device=DEV)
    wp[:N] = w.to(DEV, torch.bfloat16)
    res = torch.randn(M, N, generator=g).to(DEV, torch.bfloat16)
    gamma = (1 + 0.1 * torch.randn(N, generator=g)).to(DEV)
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
    wf = to_fragment_major(wp) if frag else None
    plain = lib.gemm(a, wp, N, residual=res, splitk_ws=ws, w_frag=wf)
    h = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    fused = lib.gemm(a, wp, N, residual=res, splitk_ws=ws, w_frag=wf, post_norm_gamma=gamma, post_norm_eps=1e-6, post_norm_out=h)
    torch.cuda.synchronize()
    assert torch.equal(fused, plain)
    assert torch.equal(h, lib.rmsnorm(plain, gamma, 1e-6))
    # in place over the residual, as the decoder layer calls it
    x = res.clone()
    lib.gemm(a, wp, N, residual=x, out=x, splitk_ws=ws, w_frag=wf, post_norm_gamma=gamma, post_norm_eps=1e-6, post_norm_out=h)
    torch.cuda.synchronize()
    assert torch.equal(x, plain)
    with pytest.raises(ValueError):
        lib.gemm(a, wp, N, epilogue=1, post_norm_gamma=gamma, post_norm_eps=1e-6, post_norm_out=h)


@pytest.mark.parametrize("M", [2, 33, 64, 100])
@pytest.mark.parametrize("tag,N,K,epi", [("qkv", 12288, 4096, 0), ("o", 4096, 4096, 0), ("gate-up", 22016, 4096, 3), ("down", 4096, 11008, 0)])
@pytest.mark.parametrize("fp8", [False, True])
def test_panel_gemm_on_the_7b_decode_geometries(lib, M, tag, N, K, epi, fp8):
    """the exact launch geometries of batch decode at 7B (8-wave two-K-group form for M <= 64: gate/up without split-K, qkv with
    2 slices, o / down with 8; 4-wave form at M = 100) against an fp32 matmul of the same bf16 (or dequantised fp8) operands"""
    from visualcla.weights import to_fragment_major, quantize_fp8_rows, dequantize_fp8_rows, to_fragment_pair_major_fp8
    g = torch.Generator().manual_seed(M * 31 + N // 64 + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(DEV, torch.bfloat16)
    wp = torch.zeros((N + 127) // 128 * 128, K, dtype=torch.bfloat16, device=DEV)
    wp[:N] = (torch.randn(N, K, generator=g) * 0.03).to(DEV, torch.bfloat16)
    n_out = N // 2 if epi == 3 else N
    res = torch.randn(M, n_out, generator=g).to(DEV, torch.bfloat16)
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
    if fp8:
        q8, sc8 = quantize_fp8_rows(wp)
        w_ref = dequantize_fp8_rows(q8, sc8)[:N]
        got = lib.gemm(a, wp, N, residual=res, epilogue=epi, splitk_ws=ws, w_q8=q8, w_q8_frag=to_fragment_pair_major_fp8(q8), w_scale=sc8)
    else:
        w_ref = wp[:N]
        got = lib.gemm(a, wp, N, residual=res, epilogue=epi, splitk_ws=ws, w_frag=to_fragment_major(wp))
    ref = _gemm_ref(a.float().cpu(), w_ref.float().cpu(), None, epi, res.float().cpu())
    _cmp(f"panel7b[{tag},M{M},fp8={fp8}]", got, ref, atol=2e-2, rtol=1e-2)


# ------------------------------------------------------------------ streaming decode GEMM (kernel 9) and its fragment-major producers
@pytest.mark.parametrize("rows,cols", [(64, 4096), (2, 256), (33, 512), (17, 8192), (48, 1408)])
def test_rmsnorm_pack_is_the_rmsnorm_in_fragment_layout(lib, rows, cols):
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * 1.5).to(DEV, torch.bfloat16)
    gamma = (1 + 0.1 * torch.randn(cols, generator=g)).to(DEV)
    want = lib.rmsnorm(x, gamma, 1e-6)
    got = lib.rmsnorm_pack(x, gamma, 1e-6)
    torch.cuda.synchronize()
    assert torch.equal(lib.from_frag(got, rows), want)                 # bit-identical values, only the layout differs
    assert torch.equal(got, lib.to_frag(want))                         # padding rows stay zero
    assert torch.equal(lib.rmsnorm_pack(x, None, 0.0), lib.to_frag(x))  # gamma = NULL: plain re-layout


DS_SHAPES = [
    # M, N, K
    (64, 4096, 4096), (64, 12288, 4096), (64, 4096, 11008), (2, 512, 256), (33, 1000, 1408), (7, 320, 64), (16, 128, 512),
    (48, 2080, 2048), (64, 49958, 512), (17, 96, 4096), (3, 16, 8192),
]


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("epi", [0, 3])
@pytest.mark.parametrize("M,N,K", DS_SHAPES)
def test_gemm_dstream(lib, M, N, K, epi, fp8):
    """kernel 9: fragment-major A and W, every workgroup an equal share of tiles over the full K, cross-wave reduction in LDS"""
    from visualcla.weights import to_fragment_major, quantize_fp8_rows, dequantize_fp8_rows, to_fragment_pair_major_fp8
    if epi == 3:
        N = (N + 31) // 32 * 32
    g = torch.Generator().manual_seed(M * 1000 + N + K + epi)
    a = bf16r(torch.randn(M, K, generator=g))
    w = bf16r(torch.randn(N, K, generator=g) * 0.05)
    bias = bf16r(torch.randn(N, generator=g) * 0.1)
    n_out = N // 2 if epi == 3 else N
    res = bf16r(torch.randn(M, n_out, generator=g))
    wp = _pack(w)
    kw = dict(w_frag=to_fragment_major(wp))
    wref = w
    if fp8:
        q, sc = quantize_fp8_rows(wp)
        wref = dequantize_fp8_rows(q, sc)[:N].cpu()
        kw = dict(w_q8_frag=to_fragment_pair_major_fp8(q), w_scale=sc)
    ref = _gemm_ref(a, wref, bias, epi, res)
    af = lib.to_frag(a.to(DEV, torch.bfloat16))
    cf = torch.zeros(n_out // 32, (M + 15) // 16, 64, 8, dtype=torch.bfloat16, device=DEV) if n_out % 32 == 0 else None
    got = lib.gemm(None, wp, N, bias=bias.to(DEV), residual=res.to(DEV, torch.bfloat16), epilogue=epi, force_kernel=9,
                   a_frag=af, m=M, c_frag=cf, **kw)
    _cmp(f"gemm_dstream[fp8={fp8},epi{epi},{M}x{N}x{K}]", got, ref, atol=3e-3 if fp8 else 2e-3, rtol=8e-3)
    if cf is not None:
        torch.cuda.synchronize()
        assert torch.equal(lib.from_frag(cf, M), got)                  # the fragment-major copy holds the same bf16 values
    if epi == 0:   # fp32 output (lm_head), no residual / bias
        got32 = lib.gemm(None, wp, N, out_f32=True, force_kernel=9, a_frag=af, m=M, **kw)
        _cmp(f"gemm_dstream_f32[fp8={fp8},{M}x{N}x{K}]", got32, _gemm_ref(a, wref, None, 0, None), atol=1e-3, rtol=2e-3)


def test_gemm_dstream_identity_in_place_and_determinism(lib):
    from visualcla.weights import to_fragment_major
    # A = I with an asymmetric W catches operand / row-column swaps of either fragment layout
    M, K, N = 48, 192, 333
    a = torch.zeros(M, K)
    a[torch.arange(M), (torch.arange(M) * 3) % K] = 1.0
    g = torch.Generator().manual_seed(5)
    w = bf16r(torch.randn(N, K, generator=g))
    wp = _pack(w)
    got = lib.gemm(None, wp, N, out_f32=True, force_kernel=9, a_frag=lib.to_frag(a.to(DEV, torch.bfloat16)), m=M, w_frag=to_fragment_major(wp))
    _cmp("gemm_dstream_identity", got, a @ w.t(), atol=0.0)
    # residual read and written in place (the decode residual stream), twice the same bits
    M, N, K = 64, 4096, 4096
    a, w2, x = bf16r(torch.randn(M, K, generator=g)), bf16r(torch.randn(N, K, generator=g) * 0.05), bf16r(torch.randn(M, N, generator=g))
    wp2 = _pack(w2)
    wf2, af = to_fragment_major(wp2), lib.to_frag(a.to(DEV, torch.bfloat16))
    outs = []
    for _ in range(2):
        xd = x.to(DEV, torch.bfloat16)
        lib.gemm(None, wp2, N, residual=xd, out=xd, force_kernel=9, a_frag=af, m=M, w_frag=wf2)
        outs.append(xd)
    _cmp("gemm_dstream_inplace_residual", outs[0], x + a @ w2.t(), atol=2e-3, rtol=8e-3)
    assert torch.equal(outs[0], outs[1])
    # and the same function as the panel kernel it replaces, to accumulation-order noise
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
    ref8 = lib.gemm(a.to(DEV, torch.bfloat16), wp2, N, out_f32=True, force_kernel=8, splitk_ws=ws, w_frag=wf2)
    got9 = lib.gemm(None, wp2, N, out_f32=True, force_kernel=9, a_frag=af, m=M, w_frag=wf2)
    _cmp("gemm_dstream_vs_panel", got9, ref8, atol=2e-4, rtol=1e-4)


def test_gemm_dstream_error_conventions(lib):
    from visualcla.weights import to_fragment_major
    a = torch.randn(65, 256).to(DEV, torch.bfloat16)
    wp = _pack(torch.randn(128, 256))
    with pytest.raises(ValueError):    # M > 64
        lib.gemm(None, wp, 128, force_kernel=9, a_frag=lib.to_frag(a), m=65, w_frag=to_fragment_major(wp))
    with pytest.raises(ValueError):    # no fragment-major weights
        lib.gemm(None, wp, 128, force_kernel=9, a_frag=lib.to_frag(a[:8]), m=8)
    with pytest.raises(ValueError):    # an activation epilogue the streaming kernel does not implement
        lib.gemm(None, wp, 128, force_kernel=9, a_frag=lib.to_frag(a[:8]), m=8, w_frag=to_fragment_major(wp), epilogue=1)
    with pytest.raises(ValueError):    # C_frag from a kernel that cannot write it
        lib.gemm(a[:8], wp, 128, force_kernel=1, c_frag=torch.zeros(4, 1, 64, 8, dtype=torch.bfloat16, device=DEV))


@pytest.mark.parametrize("B,H,d,pos", [(64, 32, 128, 190), (2, 4, 128, 37), (33, 8, 64, 3), (16, 32, 32, 65)])
def test_attn_decode_fragment_major_output(lib, B, H, d, pos):
    """out_frag = 1 stores the same values as the row-major form, in the layout the streaming o_proj GEMM reads"""
    from visualcla.weights import rope_tables
    ctx = max(64, (pos + 64) // 64 * 64)
    g = torch.Generator().manual_seed(B + H + d + pos)
    qkv = torch.randn(B, 3 * H * d, generator=g).to(DEV, torch.bfloat16)
    kc0 = torch.randn(B, H, ctx, d, generator=g).to(DEV, torch.bfloat16)
    vc0 = torch.randn(B, H, ctx, d, generator=g).to(DEV, torch.bfloat16)
    cos, sin = (t.to(DEV) for t in rope_tables(1024, d, 10000.0))
    outs = []
    for frag in (0, 1):
        kc, vc = kc0.clone(), vc0.clone()
        out = torch.zeros((H * d) // 32, (B + 15) // 16, 64, 8, dtype=torch.bfloat16, device=DEV) if frag else \
            torch.empty(B, H * d, dtype=torch.bfloat16, device=DEV)
        lib.check(lib.load().vcla_attn_decode_fused(qkv.data_ptr(), kc.data_ptr(), vc.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                                    out.data_ptr(), B, H, d, ctx, pos, None, None, 0, 1 / math.sqrt(d),
                                                    lib.dtype_code(torch.bfloat16), frag, lib.stream_ptr()))
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(lib.from_frag(outs[1], B), outs[0])


@pytest.mark.parametrize("M,D,N2,epi2", [(64, 4096, 12288, 0), (64, 4096, 22016, 3), (33, 512, 1408 * 2, 3), (2, 256, 320, 0)])
def test_gemm_dstream_deferred_rmsnorm(lib, M, D, N2, epi2):
    """o_proj / down_proj -> RMSNorm -> next GEMM without a norm launch: the producer stores gamma * x fragment-major plus per-row
    partial sums of squares (c_frag_gamma, c_row_ssq), the consumer scales its accumulators by rstd (a_row_ssq):
    W . bf16(gamma * x) * rstd(x).  One bf16 rounding fewer than HF's gamma * bf16(x * rstd) -- compared against both."""
    from visualcla.weights import to_fragment_major
    g = torch.Generator().manual_seed(M + D + N2)
    a = bf16r(torch.randn(M, D, generator=g))
    w1 = bf16r(torch.randn(D, D, generator=g) * 0.03)
    res = bf16r(torch.randn(M, D, generator=g))
    gamma = bf16r(1 + 0.1 * torch.randn(D, generator=g))
    w2 = bf16r(torch.randn(N2, D, generator=g) * 0.05)
    w1p, w2p = _pack(w1), _pack(w2)
    xd = res.to(DEV, torch.bfloat16)
    cf = torch.zeros(D // 32, (M + 15) // 16, 64, 8, dtype=torch.bfloat16, device=DEV)
    ssq = torch.zeros(M, D // 16, dtype=torch.float32, device=DEV)
    lib.gemm(None, w1p, D, residual=xd, out=xd, force_kernel=9, a_frag=lib.to_frag(a.to(DEV, torch.bfloat16)), m=M,
             w_frag=to_fragment_major(w1p), c_frag=cf, c_frag_gamma=gamma.to(DEV), c_row_ssq=ssq)
    torch.cuda.synchronize()
    x = xd.float().cpu()                                               # the residual stream as stored (bf16)
    _cmp("deferred_norm.x", x, res + a @ w1.t(), atol=2e-3, rtol=8e-3)
    assert torch.equal(lib.from_frag(cf, M).float().cpu(), bf16r(gamma * x))
    _cmp("deferred_norm.ssq", ssq.sum(1), (x * x).sum(1), atol=0.0, rtol=1e-5)
    rstd = torch.rsqrt((x * x).mean(1, keepdim=True) + 1e-6)
    got = lib.gemm(None, w2p, N2, epilogue=epi2, force_kernel=9, a_frag=cf, m=M, w_frag=to_fragment_major(w2p), a_row_ssq=ssq, a_norm_eps=1e-6)
    w2s = w2
    ref = _gemm_ref(bf16r(gamma * x) * rstd, w2s, None, epi2, None)      # what the kernels compute, in fp32
    _cmp(f"deferred_norm.consumer[{M}x{N2}x{D},epi{epi2}]", got, ref, atol=2e-3, rtol=8e-3)
    hf = _gemm_ref(O.llama_rmsnorm(xd.cpu(), gamma.to(torch.bfloat16), 1e-6).float(), w2, None, epi2, None)   # HF's rounding order
    err = (got.float().cpu() - hf).abs()
    _report(f"deferred rmsnorm vs HF order [{M}x{N2}x{D},epi{epi2}]: max {err.max().item():.3e} mean {err.mean().item():.3e} (ref absmax {hf.abs().max().item():.2e})")
    assert err.max().item() <= 2e-2 * max(1.0, hf.abs().max().item())


# ------------------------------------------------------------------ fp8 x fp8 on the fp8 MFMA pipe (kernel 10, BASELINE configs[4])
@pytest.mark.parametrize("rows,cols", [(300, 4096), (7, 128), (129, 11008), (64, 1024)])
def test_quant_fp8_rows_matches_torch(lib, rows, cols):
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * torch.rand(rows, 1, generator=g) * 3).to(torch.bfloat16)
    x[min(3, rows - 1)] = 0                                            # an all-zero row must not divide by zero
    q, sc = lib.quant_fp8_rows(x.to(DEV))
    torch.cuda.synchronize()
    xf = x.float()
    sc_ref = (xf.abs().amax(dim=1) / 448.0).clamp_min(1e-20)
    assert torch.equal(sc.cpu(), sc_ref)
    q_ref = (xf * (1.0 / sc_ref)[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(q.cpu(), q_ref)                                 # same scale, same round-to-nearest-even


FP8_MFMA_SHAPES = [(300, 512, 256, 0), (1000, 1000, 1408, 1), (257, 384, 128, 2), (514, 2048, 1024, 3), (2 * 257, 4096, 1024, 1), (130, 320, 11008, 0),
                   # the four GEMMs of a LLaMA-7B layer at the BASELINE widths (a 256-row prefill; o_proj at 1024 rows = 4 row tiles): the kernel is EXACT on the
                   # operands it is given -- this is where W8A8's kernel error is separated from its format noise (tests/test_gpu_model.py explains why a layer-level
                   # comparison cannot do it)
                   (256, 12288, 4096, 0), (256, 22016, 4096, 3), (300, 4096, 11008, 0), (1024, 4096, 4096, 0)]


@pytest.mark.parametrize("M,N,K,epi", FP8_MFMA_SHAPES)
def test_gemm_fp8_mfma(lib, M, N, K, epi):
    """kernel 10 computes exactly the function of the DEQUANTISED operands (e4m3 x e4m3 products are exact in fp32), fp32
    accumulate, row scales applied in the epilogue; the distance to the unquantised product is reported"""
    from visualcla.weights import quantize_fp8_rows, dequantize_fp8_rows
    if epi == 3:
        N = (N + 31) // 32 * 32
    g = torch.Generator().manual_seed(M + N + K + epi)
    a = bf16r(torch.randn(M, K, generator=g))
    w = bf16r(torch.randn(N, K, generator=g) * 0.05)
    bias = bf16r(torch.randn(N, generator=g) * 0.1)
    n_out = N // 2 if epi == 3 else N
    res = bf16r(torch.randn(M, n_out, generator=g))
    wp = _pack(w)
    wq, ws_ = quantize_fp8_rows(wp)
    aq, as_ = lib.quant_fp8_rows(a.to(DEV, torch.bfloat16))
    a_dq = dequantize_fp8_rows(aq, as_).cpu()
    w_dq = dequantize_fp8_rows(wq, ws_)[:N].cpu()
    ref = _gemm_ref(a_dq, w_dq, bias, epi, res)
    got = lib.gemm(None, wp, N, bias=bias.to(DEV), residual=res.to(DEV, torch.bfloat16), epilogue=epi, force_kernel=10,
                   a_q8=aq, a_scale=as_, w_q8=wq, w_scale=ws_)
    _cmp(f"gemm_fp8_mfma[{M}x{N}x{K},epi{epi}]", got, ref, atol=3e-3, rtol=8e-3)
    full = _gemm_ref(a, w, bias, epi, res)
    err = (got.float().cpu() - full).abs()
    _report(f"gemm_fp8_mfma[{M}x{N}x{K},epi{epi}] vs the unquantised product: max {err.max().item():.3e} mean {err.mean().item():.3e} (ref absmax {full.abs().max().item():.2e})")
    if epi == 0:
        got32 = lib.gemm(None, wp, N, out_f32=True, force_kernel=10, a_q8=aq, a_scale=as_, w_q8=wq, w_scale=ws_)
        _cmp(f"gemm_fp8_mfma_f32[{M}x{N}x{K}]", got32, _gemm_ref(a_dq, w_dq, None, 0, None), atol=1e-3, rtol=2e-3)


def test_gemm_fp8_mfma_identity_and_errors(lib):
    """A = I (exactly representable) against an asymmetric W of exactly representable values: catches any operand-layout slip of
    the 16x16x128 fragment (32 consecutive k per lane)"""
    M, K, N = 300, 384, 333
    a = torch.zeros(M, K)
    a[torch.arange(M), (torch.arange(M) * 5) % K] = 1.0
    g = torch.Generator().manual_seed(5)
    w = (torch.randint(-8, 9, (N, K), generator=g).float() / 8.0)     # multiples of 1/8 up to 1: exact in e4m3
    wp = _pack(w)
    from visualcla.weights import quantize_fp8_rows
    wq = wp.float().to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
    ones_w = torch.ones(wp.shape[0], device=DEV)
    aq = a.to(DEV).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
    ones_a = torch.ones(M, device=DEV)
    got = lib.gemm(None, wp, N, out_f32=True, force_kernel=10, a_q8=aq, a_scale=ones_a, w_q8=wq, w_scale=ones_w)
    _cmp("gemm_fp8_mfma_identity", got, a @ w.t(), atol=0.0)
    with pytest.raises(ValueError):    # K must be a multiple of 128
        lib.gemm(None, _pack(torch.randn(64, 192)), 64, force_kernel=10, a_q8=aq[:, :192].contiguous(), a_scale=ones_a,
                 w_q8=wq[:128, :192].contiguous(), w_scale=ones_w)
    with pytest.raises(ValueError):    # scales are mandatory
        lib.gemm(None, wp, N, force_kernel=10, a_q8=aq, w_q8=wq, w_scale=ones_w)


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("M,N,K,S", [(64, 4096, 4096, 4), (64, 4096, 11008, 4), (33, 1000, 1408, 2), (48, 2080, 2048, 8), (7, 320, 512, 3), (64, 4096, 4096, 16)])
def test_gemm_dstream_splitk(lib, M, N, K, S, fp8):
    """ds_splitk: S K slices per tile group, fp32 partial tiles in the workspace, a second launch sums them in slice order and runs
    the epilogue (bias, residual in place, fragment-major copy x gamma, row sums of squares) -- same function as the unsplit
    kernel, deterministic"""
    from visualcla.weights import to_fragment_major, quantize_fp8_rows, dequantize_fp8_rows, to_fragment_pair_major_fp8
    g = torch.Generator().manual_seed(M * 1000 + N + K + S)
    a = bf16r(torch.randn(M, K, generator=g))
    w = bf16r(torch.randn(N, K, generator=g) * 0.05)
    bias = bf16r(torch.randn(N, generator=g) * 0.1)
    res = bf16r(torch.randn(M, N, generator=g))
    gamma = bf16r(1 + 0.1 * torch.randn(N, generator=g))
    wp = _pack(w)
    kw = dict(w_frag=to_fragment_major(wp))
    wref = w
    if fp8:
        q, sc = quantize_fp8_rows(wp)
        wref = dequantize_fp8_rows(q, sc)[:N].cpu()
        kw = dict(w_q8_frag=to_fragment_pair_major_fp8(q), w_scale=sc)
    ref = _gemm_ref(a, wref, bias, 0, res)
    af = lib.to_frag(a.to(DEV, torch.bfloat16))
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=DEV)
    with_frag = N % 32 == 0
    outs = []
    for _ in range(2):
        xd = res.to(DEV, torch.bfloat16)
        cf = torch.zeros(N // 32, (M + 15) // 16, 64, 8, dtype=torch.bfloat16, device=DEV) if with_frag else None
        ssq = torch.zeros(M, N // 16, dtype=torch.float32, device=DEV) if N % 16 == 0 else None
        lib.gemm(None, wp, N, bias=bias.to(DEV), residual=xd, out=xd, force_kernel=9, a_frag=af, m=M, splitk_ws=ws, ds_splitk=S,
                 c_frag=cf, c_frag_gamma=gamma.to(DEV) if with_frag else None, c_row_ssq=ssq, **kw)
        outs.append((xd, cf, ssq))
    torch.cuda.synchronize()
    xd, cf, ssq = outs[0]
    _cmp(f"gemm_dstream_splitk[fp8={fp8},S{S},{M}x{N}x{K}]", xd, ref, atol=3e-3 if fp8 else 2e-3, rtol=8e-3)
    x = xd.float().cpu()
    if cf is not None:
        assert torch.equal(lib.from_frag(cf, M).float().cpu(), bf16r(gamma * x))
    if ssq is not None:
        # the reduce launch writes one partial per 256 columns when N % 256 == 0 ([M][N/256], packed at the front of the buffer),
        # else one per 16-column tile ([M][N/16])
        parts = N // 256 if N % 256 == 0 else N // 16
        rows = ssq.flatten()[:M * parts].view(M, parts)
        _cmp("gemm_dstream_splitk.ssq", rows.sum(1), (x * x).sum(1), atol=0.0, rtol=1e-5)
        if parts != N // 16:
            assert float(ssq.flatten()[M * parts:].abs().max()) == 0.0
            # ... and the consumer given that layout reproduces W2 . RMSNorm(x): same check as test_gemm_dstream_deferred_rmsnorm
            g2 = torch.Generator().manual_seed(5)
            w2 = bf16r(torch.randn(64, N, generator=g2) * 0.05)
            w2p = _pack(w2)
            got2 = lib.gemm(None, w2p, 64, force_kernel=9, a_frag=cf, m=M, w_frag=to_fragment_major(w2p), a_row_ssq=rows.contiguous(), a_norm_eps=1e-6)
            rstd = torch.rsqrt((x * x).mean(1, keepdim=True) + 1e-6)
            ref2 = _gemm_ref(bf16r(gamma * x) * rstd, w2, None, 0, None)      # what the kernels compute, in fp32
            _cmp("gemm_dstream_splitk.consumer", got2, ref2, atol=2e-3, rtol=8e-3)
    assert torch.equal(outs[0][0], outs[1][0])
