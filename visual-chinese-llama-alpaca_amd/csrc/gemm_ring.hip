// gemm_ring.hip -- kernel 11 of vcla_gemm: the INTAKE-bound MFMA tile kernel for 129 - 256 activation rows (LLaMA decode batches of
// 129 - 256 sequences = the N = 1 leg of north_star's batch-256 claim; also prefills of 129 - 256 prompt rows).
//
// Why a kernel of its own (profiles/r05_l2_intake.txt, tools/l2_intake.hip): at these shapes neither MFMA (46 GF per gate/up launch = 18 us at
// peak) nor HBM (180 MB = 27 us) binds -- what a CU can INGEST does.  Measured on MI355X: an L2-resident panel arrives at 56 B/clk/CU (34.5 TB/s
// chip) with >= 32 KiB in flight per CU, at 17 - 20 B/clk per wave (latency-bound per wave: ~450 cycles per round trip); beside an HBM weight
// stream the same reads slow to 10 - 30 B/clk/CU and starve the stream (every request, hit or miss, queues behind the misses of its CU), so the
// rate of either operand is (its bytes in flight) / (loaded latency ~1 - 2 us).  The round-4 dispatch (128 x 128 register-staged tiles, 8 K
// slices + a reduce launch) kept ~32 KiB in flight per workgroup and re-read the activation panel per slice: gate/up 90 us, o_proj / down_proj
// 2 x 43 us at M = 256.  This kernel keeps 100 - 130 KiB in flight per CU in an LDS ring fed by LDS-DMA from ALL 8 waves:
//   * tile BM x BN over the FULL K (no split-K partials, no reduce launch), one workgroup per CU, tiles chosen per shape so that
//     ceil(M / BM) * ceil(N / BN) <= 256: 256 x 96 (gate/up), 128 x 96 (qkv), 64 x 64 (o_proj / down_proj: each W row block is fetched from
//     HBM once and hits L2 for the other three row tiles, which run on the same XCD);
//   * stage = KS K-slabs of 64: BM x 64 activations + BN x 64 weights, `global_load_lds_dwordx4` pieces of 1 KiB (8 rows x 128 B; fp8 weights:
//     16 rows x 64 B) dealt round-robin to the 8 waves, bank-conflict swizzle applied on the SOURCE address (as gemm_mfma256.hip);
//   * NS stages: NS - 1 in flight while one is consumed; all DMA from inline asm with hand-counted vmcnt, ONE barrier per stage; every wave
//     issues exactly PP DMA instructions per stage (absent pieces and the K tail are 4-byte dummies into a sink) so the count is a literal;
//   * fp8 (e4m3fn) weights (W8A16, BASELINE configs[4]): the 1-byte rows are staged as they are (half the LDS and HBM bytes), fragments are read
//     with ds_read_b64 and widened to bf16 in registers (exact), the per-row scale is applied in the epilogue -- the same function of the
//     dequantised weights as the M <= 128 decode kernels compute.
// Operands are the plain row-major matrices (A [M, lda], W [N_pad, K]); epilogue = gemm_epilogue.h (bias / residual / SwiGLU / fp8 scales).
#include "vcla_common.h"
#include "gemm_epilogue.h"
#include "gemm_tiles.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void gr_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void gr_dma4(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void gr_dma16_nt(const void* gsrc, unsigned lds_dst) {      // non-temporal: a line one CU reads once should not displace the panel in L2
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void gr_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 8 OCP fp8 (e4m3fn) values in two dwords -> one bf16x8 MFMA operand (exact: e4m3 fits in bf16)
__device__ __forceinline__ bf16x8_t gr_fp8x8_to_bf16x8(uint32_t lo, uint32_t hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
    const f32x2_t c = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
    const u32x4_t p = {pack_bf2(a.x, a.y), pack_bf2(b.x, b.y), pack_bf2(c.x, c.y), pack_bf2(d.x, d.y)};
    return __builtin_bit_cast(bf16x8_t, p);
}

// BM x BN output tile, WM x WN = 8 waves (wave (wm, wn) owns BM/WM rows x BN/WN columns), NS ring stages of KS K-slabs, W8 = fp8 weights
// VAR (experiments, tools/bench_kernels.py ring; VCLA_RING_VAR): bit 0 = DMA statements spread over the MFMA groups of the stage instead of issued in one
// burst behind the barrier; bit 1 = nt policy on the weight pieces; bits 4.. = timing ablations with GARBAGE results: 16 = no fragment reads / MFMAs
// (DMA + barriers only), 32 = no weight DMA, 48 = no activation DMA
template <int EPI, typename OutT, int BM, int BN, int WM, int WN, int NS, int KS, bool W8, int VAR = 0>
__global__ __launch_bounds__(512) void gemm_ring_kernel(vcla_gemm_args a, int tiles_m, int tiles_n, int n_pad) {
    constexpr bool SPREAD = (VAR & 1) != 0, WNT = (VAR & 2) != 0;
    constexpr int ABL = VAR >> 4;
    static_assert(WM * WN == 8 && BM % 64 == 0 && BM % (16 * WM) == 0 && BN % (16 * WN) == 0 && NS >= 3, "tile / wave grid");
    static_assert(EPI != VCLA_EPI_SWIGLU || (BN / WN) % 32 == 0, "SwiGLU pairs (gate, up) tiles inside a wave");
    extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
    constexpr int A_BYTES = BM * 128;                          // one K slab of the activations: BM rows x 64 bf16
    constexpr int W_BYTES = W8 ? BN * 64 : BN * 128;           // ... of the weights
    constexpr int SLAB = A_BYTES + W_BYTES, STAGE = KS * SLAB;
    constexpr int IA = BM / 64;                                // A pieces per wave and slab (BM / 8 pieces of 1 KiB over 8 waves)
    constexpr int PW = W_BYTES / 1024;                         // W pieces per slab
    constexpr int IW = (PW + 7) / 8;                           // ... per wave (the last one may be absent: dummy)
    constexpr int PP = KS * (IA + IW);                         // DMA instructions per wave and stage, exactly
    constexpr int MI = BM / WM / 16, NJ = BN / WN / 16;
    static_assert(W_BYTES % 1024 == 0, "whole pieces");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int tm, tn;
    tile_assign(blockIdx.x, tiles_m, tiles_n, tiles_m, tm, tn);      // consecutive blocks of an XCD: the row tiles of ONE column tile (they share its W rows)
    const int m0 = tm * BM, n0 = tn * BN;
    const unsigned lds_u = (unsigned)(uintptr_t)(lds_ptr_t)ring;
    const unsigned sink = lds_u + NS * STAGE;                   // 256 B nobody reads: destination of the dummies

    // ---- sources of this wave's pieces (slab 0), LDS offsets inside a slab.  Row-major operands: rows lda / K elements apart, K slabs 128 B
    // (fp8: 64 B) apart; slab-major operands ([K/64][rows][64], vcla_gemm_args.A_slab / W_slab / W_q8_slab): rows 128 B apart, slabs rows * 128 B
    // apart -- a piece's 8 (16) rows are then ONE contiguous 1 KiB, which is what the DMA path moves at full rate
    const char* Ab = (const char*)(a.A_slab ? a.A_slab : a.A);
    const int64_t a_rs = a.A_slab ? 128 : a.lda * 2, a_ss = a.A_slab ? a.a_slab_rows * 128 : 128;
    constexpr int W_ROW = W8 ? 64 : 128;                      // bytes of one weight row inside a K slab
    const bool wslab = W8 ? a.W_q8_slab != nullptr : a.W_slab != nullptr;
    const char* Wb = (const char*)(W8 ? (wslab ? a.W_q8_slab : a.W_q8) : (wslab ? a.W_slab : a.W));
    const int64_t w_rs = wslab ? W_ROW : (int64_t)a.K * (W8 ? 1 : 2), w_ss = wslab ? (int64_t)n_pad * W_ROW : W_ROW;
    const char* asrc[IA];
    const char* wsrc[IW];
    bool wreal[IW];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int piece = wave + 8 * i, row = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);          // source-side swizzle (involution shared with lds_off)
        int am = m0 + row;
        am = am < a.M ? am : a.M - 1;
        asrc[i] = Ab + (int64_t)am * a_rs + chunk * 16;
    }
#pragma unroll
    for (int i = 0; i < IW; ++i) {
        const int q = wave + 8 * i;
        wreal[i] = q < PW;
        const int qq = wreal[i] ? q : 0;
        if constexpr (W8) {
            const int row = qq * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((row >> 2) & 3);          // 16-byte chunk of the 64-byte slab row this lane fetches
            int wr = n0 + row;
            wr = wr < n_pad ? wr : n_pad - 1;
            wsrc[i] = Wb + (int64_t)wr * w_rs + c * 16;
        } else {
            const int row = qq * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            int wr = n0 + row;
            wr = wr < n_pad ? wr : n_pad - 1;
            wsrc[i] = Wb + (int64_t)wr * w_rs + chunk * 16;
        }
    }
    const int nslab = a.K / GM_BK;
    // DMA slot `idx` (0 .. PP - 1; compile-time after unrolling) of stage `st` into ring buffer `buf`: slab k = idx / (IA + IW), then IA activation
    // pieces and IW weight pieces.  Every slot issues exactly ONE instruction (absent pieces / the K tail: a 4-byte dummy into the sink).
    auto issue_slot = [&](int idx, int st, int buf) {
        const int k = idx / (IA + IW), r = idx % (IA + IW);
        const int slab = st * KS + k;
        const bool live = slab < nslab;
        const unsigned sb = lds_u + buf * STAGE + k * SLAB;
        if (r < IA) {
            if (live && ABL != 3) gr_dma16(asrc[r] + (int64_t)slab * a_ss, sb + (unsigned)(wave + 8 * r) * 1024u);
            else gr_dma4(asrc[r], sink);
        } else {
            const int i = r - IA;
            if (live && wreal[i] && ABL != 2) {
                if constexpr (WNT) gr_dma16_nt(wsrc[i] + (int64_t)slab * w_ss, sb + A_BYTES + (unsigned)(wave + 8 * i) * 1024u);
                else gr_dma16(wsrc[i] + (int64_t)slab * w_ss, sb + A_BYTES + (unsigned)(wave + 8 * i) * 1024u);
            } else gr_dma4(wsrc[i], sink);
        }
    };
    auto issue = [&](int st, int buf) {
#pragma unroll
        for (int idx = 0; idx < PP; ++idx) issue_slot(idx, st, buf);
    };

    f32x4_t acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fch = lane >> 4;
    const int nst = (nslab + KS - 1) / KS;

#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s, s);
    int buf_c = 0, buf_i = NS - 1;
    for (int st = 0; st < nst; ++st) {
        gr_vmcnt<(NS - 2) * PP>();                              // this wave's pieces of stage st have landed (NS - 2 younger stages may not have)
        __builtin_amdgcn_s_barrier();                          // ... everyone's; and the buffer of stage st - 1 is no longer read
        asm volatile("" ::: "memory");
        if constexpr (!SPREAD) issue(st + NS - 1, buf_i);
        const unsigned char* Sb = ring + buf_c * STAGE;
        constexpr int G = KS * 2 * MI;                          // MFMA groups (NJ MFMAs each) of a stage: the SPREAD form issues PP DMA statements between them
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const bool have = !(KS > 1 && st * KS + k >= nslab);  // K tail of a multi-slab stage (wave-uniform); the DMA slots are issued regardless
            const unsigned char* As = Sb + k * SLAB;
            const unsigned char* Ws = As + A_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8_t wf[NJ], af[MI];
                if (have && ABL != 1) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int row = wn * (BN / WN) + j * 16 + frow;
                        if constexpr (W8) {
                            const int cg = kk * 2 + (fch >> 1);
                            const uint2 raw = *reinterpret_cast<const uint2*>(Ws + row * 64 + ((cg ^ ((row >> 2) & 3)) << 4) + (fch & 1) * 8);
                            wf[j] = gr_fp8x8_to_bf16x8(raw.x, raw.y);
                        } else {
                            wf[j] = *reinterpret_cast<const bf16x8_t*>(Ws + lds_off(row, kk * 4 + fch));
                        }
                    }
#pragma unroll
                    for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(As + lds_off(wm * (BM / WM) + i * 16 + frow, kk * 4 + fch));
                }
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    if (have && ABL != 1) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
                    }
                    if constexpr (SPREAD) {                      // this group's share of the stage's PP DMA statements
                        constexpr int dummy = 0; (void)dummy;
                        const int g = (k * 2 + kk) * MI + i;
#pragma unroll
                        for (int idx = 0; idx < PP; ++idx)
                            if (idx * G / PP == g) issue_slot(idx, st + NS - 1, buf_i);
                    }
                }
            }
        }
        asm volatile("" ::: "memory");                          // the fragment reads stay on this side of the next barrier
        buf_c = buf_c + 1 == NS ? 0 : buf_c + 1;
        buf_i = buf_i + 1 == NS ? 0 : buf_i + 1;
    }
    gr_vmcnt<0>();                                              // no DMA (the tail's dummies) may land in LDS after the workgroup has given it up
    gemm_epilogue<EPI, OutT, MI, NJ, W8>(a, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);
}

// ------------------------------------------------------------------ host side
template <int EPI, typename OutT, int BM, int BN, int WM, int WN, int NS, int KS, bool W8, int VAR = 0>
static int launch_ring_cfg(const vcla_gemm_args* a, hipStream_t s) {
    constexpr size_t lds = (size_t)NS * KS * (BM * 128 + (W8 ? BN * 64 : BN * 128)) + 256;
    static_assert(lds <= 160 * 1024, "ring exceeds the 160 KiB of a CU");
#ifdef VCLA_RING_EXPERIMENTS
    if constexpr (VAR == 0 && sizeof(OutT) == 2 && !W8) {       // experiment builds: VCLA_RING_VAR selects a variant of the bf16 instances at run time
        const char* e = getenv("VCLA_RING_VAR");
        switch (e ? atoi(e) : 0) {
            case 1: return launch_ring_cfg<EPI, OutT, BM, BN, WM, WN, NS, KS, W8, 1>(a, s);
            case 2: return launch_ring_cfg<EPI, OutT, BM, BN, WM, WN, NS, KS, W8, 2>(a, s);
            case 3: return launch_ring_cfg<EPI, OutT, BM, BN, WM, WN, NS, KS, W8, 3>(a, s);
            case 16: return launch_ring_cfg<EPI, OutT, BM, BN, WM, WN, NS, KS, W8, 16>(a, s);
            case 17: return launch_ring_cfg<EPI, OutT, BM, BN, WM, WN, NS, KS, W8, 17>(a, s);
            case 32: return launch_ring_cfg<EPI, OutT, BM, BN, WM, WN, NS, KS, W8, 32>(a, s);
            case 48: return launch_ring_cfg<EPI, OutT, BM, BN, WM, WN, NS, KS, W8, 48>(a, s);
            default: break;
        }
    }
#endif
    auto kern = gemm_ring_kernel<EPI, OutT, BM, BN, WM, WN, NS, KS, W8, VAR>;
    static bool attr_set[VCLA_MAX_DEVICES] = {};   // per instantiation and device
    { const int rc_ = vcla_raise_dyn_lds((const void*)kern, lds, attr_set); if (rc_) return rc_; }
    const int tiles_m = (a->M + BM - 1) / BM, tiles_n = (a->N + BN - 1) / BN;
    const int n_pad = (a->N + 127) / 128 * 128;
    kern<<<tiles_m * tiles_n, 512, lds, s>>>(*a, tiles_m, tiles_n, n_pad);
    VCLA_CHECK_LAUNCH("gemm_ring_kernel");
    return VCLA_OK;
}

// tile choice: the candidate with the fewest bytes per CU, (BM + BN) * rounds, among {256 x 96, 128 x 96, 64 x 64}; force_kernel 12 / 13 / 14 force one
static int ring_cfg(const vcla_gemm_args* a) {
    const int forced = a->force_kernel - 11;                              // 0 = choose
    if (forced >= 1 && forced <= 3 && !(forced > 1 && a->epilogue == VCLA_EPI_SWIGLU)) return forced;
    static const int bm[3] = {256, 128, 64}, bn[3] = {96, 96, 64};
    int best = 0;
    long best_cost = 0;
    for (int c = 0; c < 3; ++c) {
        if (c > 0 && a->epilogue == VCLA_EPI_SWIGLU) continue;            // SwiGLU needs (gate, up) tile PAIRS inside a wave: only the 96-columns-per-wave tile has them
        const long tiles = (long)((a->M + bm[c] - 1) / bm[c]) * ((a->N + bn[c] - 1) / bn[c]);
        const long cost = ((tiles + 255) / 256) * (bm[c] + bn[c]);
        if (!best || cost < best_cost) { best = c + 1; best_cost = cost; }
    }
    return best;
}

template <int EPI, typename OutT, bool W8>
static int launch_ring(const vcla_gemm_args* a, hipStream_t s) {
    // NS: as many stages as 160 KiB hold (fp8 weights: smaller slabs -> one more stage)
    const int cfg = ring_cfg(a);
    if (cfg == 1) return launch_ring_cfg<EPI, OutT, 256, 96, 8, 1, W8 ? 4 : 3, 1, W8>(a, s);
    if constexpr (EPI == VCLA_EPI_SWIGLU) return vcla_fail(VCLA_ERR_BAD_ARG, "gemm: the SwiGLU ring tile is 256 x 96");
    else {
        if (cfg == 2) return launch_ring_cfg<EPI, OutT, 128, 96, 4, 2, W8 ? 6 : 5, 1, W8>(a, s);
        return launch_ring_cfg<EPI, OutT, 64, 64, 4, 2, W8 ? 5 : 4, 2, W8>(a, s);
    }
}

// entry point for gemm.hip's dispatch (kernel 11); arguments validated there
int vcla_gemm_ring_launch(const vcla_gemm_args* a, hipStream_t s) {
    const bool w8 = a->W_q8 != nullptr || a->W_q8_slab != nullptr;
    if (a->epilogue == VCLA_EPI_SWIGLU && !a->out_f32) return w8 ? launch_ring<VCLA_EPI_SWIGLU, bf16_t, true>(a, s) : launch_ring<VCLA_EPI_SWIGLU, bf16_t, false>(a, s);
    if (a->epilogue == VCLA_EPI_NONE) {
        if (a->out_f32) return w8 ? launch_ring<VCLA_EPI_NONE, float, true>(a, s) : launch_ring<VCLA_EPI_NONE, float, false>(a, s);     // lm_head: fp32 logits
        return w8 ? launch_ring<VCLA_EPI_NONE, bf16_t, true>(a, s) : launch_ring<VCLA_EPI_NONE, bf16_t, false>(a, s);
    }
    return vcla_fail(VCLA_ERR_BAD_ARG, "gemm: the ring kernel implements epilogues NONE (bf16 / fp32 output) and SWIGLU (bf16 output), got %d", a->epilogue);
}
