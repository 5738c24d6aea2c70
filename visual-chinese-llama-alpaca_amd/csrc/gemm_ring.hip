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
//     issues exactly PP DMA instructions per stage, branch-free (absent pieces and the K tail land in a 1-KiB sink), so the count is a literal;
//   * fp8 (e4m3fn) weights (W8A16, BASELINE configs[4]): the 1-byte rows travel as they are (half the HBM and DMA bytes), are widened to bf16 ONCE per
//     stage by the workgroup (exact) into the image the fragment reads use, and the per-row scale is applied in the epilogue -- the same function of
//     the dequantised weights as the M <= 128 decode kernels compute.
// Operands are the plain row-major matrices (A [M, lda], W [N_pad, K]) or their slab-major twins; epilogue = gemm_epilogue.h (bias / residual / SwiGLU / fp8 scales).
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
template <int N> __device__ __forceinline__ void gr_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 8 OCP fp8 (e4m3fn) values in two dwords -> one bf16x8 MFMA operand (exact: e4m3 fits in bf16)
__device__ __forceinline__ bf16x8_t gr_fp8x8_to_bf16x8(uint32_t lo, uint32_t hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
    const f32x2_t c = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
    const u32x4_t p = {pack_bf2(a.x, a.y), pack_bf2(b.x, b.y), pack_bf2(c.x, c.y), pack_bf2(d.x, d.y)};
    return __builtin_bit_cast(bf16x8_t, p);
}

// BM x BN output tile, WM x WN = 8 waves (wave (wm, wn) owns BM/WM rows x BN/WN columns), NS ring stages of KS K-slabs, W8 = fp8 weights.
// LDS: [A ring: NS stages][W ring: NS + WLEAD stages][W8 only: two bf16 images of a stage's weights][sink].
// W8: the e4m3 rows travel as they are (half the bytes) one stage AHEAD of the activations (WLEAD = 1); behind the barrier of stage st every wave
// widens 1/8 of stage st + 1's weights to bf16 ONCE (exact), into the image the fragment reads of stage st + 1 will use -- the first form of this
// kernel widened every fragment in every wave that read it (8 x the cvt work, on the critical path of each MFMA group): gate/up 78 us at M = 256
// against 62 us for the bf16 weights (profiles/r05_ring_microbench.txt).
// Measured and dropped (experiment build, profiles/r05_ring_variants.txt): the DMA statements spread over the MFMA groups instead of one burst behind the
// barrier (-0 .. 7 %, inside the noise in the model), nt policy on the weight pieces (equal); with the fragment reads and MFMAs REMOVED the kernel
// is 3 % faster, without the weight DMA 12 %, without the activation DMA 0 %: the stage cadence is set by the round trip of NS - 1 stages in flight
// (~90 - 130 KiB per CU against ~2 us under load), not by issue, compute or either operand alone.
// WF (bf16 weights): the weight pieces come from the FRAGMENT-major twin (vcla_gemm_args.W_frag, [N_pad/16][K/32][64 lanes][8]: what the M <= 128 decode kernels
// stream; the LLaMA matrices carry it already) -- piece = one (16 rows x 32 k) MFMA fragment = 1 KiB CONTIGUOUS in global memory, landing in LDS in lane order, so
// the fragment read is lane * 16 (conflict-free, no swizzle).  Half of a stage's DMA instructions are then contiguous KiBs instead of 8-row gathers.
template <int EPI, typename OutT, int BM, int BN, int WM, int WN, int NS, int KS, bool W8, bool WF = false>
__global__ __launch_bounds__(512) void gemm_ring_kernel(vcla_gemm_args a, int tiles_m, int tiles_n, int n_pad) {
    static_assert(!(W8 && WF), "fragment-major pieces are the bf16 twin");
    static_assert(WM * WN == 8 && BM % 64 == 0 && BM % (16 * WM) == 0 && BN % (16 * WN) == 0 && NS >= 3, "tile / wave grid");
    static_assert(EPI != VCLA_EPI_SWIGLU || (BN / WN) % 32 == 0, "SwiGLU pairs (gate, up) tiles inside a wave");
    extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
    constexpr int WLEAD = W8 ? 1 : 0;                          // stages the weight stream runs ahead of the activations
    constexpr int A_BYTES = BM * 128;                          // one K slab of the activations: BM rows x 64 bf16
    constexpr int W_BYTES = W8 ? BN * 64 : BN * 128;           // ... of the weights as they travel
    constexpr int WB_BYTES = BN * 128;                         // ... of the weights as bf16
    constexpr int A_STAGE = KS * A_BYTES, W_STAGE = KS * W_BYTES;
    constexpr int W_RING = NS * A_STAGE;                       // LDS offsets
    constexpr int W_BF = W_RING + (NS + WLEAD) * W_STAGE;
    constexpr int SINK = W_BF + (W8 ? 2 * KS * WB_BYTES : 0);
    constexpr int IA = BM / 64;                                // A pieces per wave and slab (BM / 8 pieces of 1 KiB over 8 waves)
    constexpr int PW = W_BYTES / 1024;                         // W pieces per slab
    constexpr int IW = (PW + 7) / 8;                           // ... per wave (the last one may be absent: dummy)
    constexpr int PP = KS * (IA + IW);                         // DMA instructions per wave and stage, exactly
    constexpr int MI = BM / WM / 16, NJ = BN / WN / 16;
    static_assert(W_BYTES % 1024 == 0, "whole pieces");
    static_assert((NS - 1) * PP <= 63, "vmcnt range");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int tm, tn;
    tile_assign(blockIdx.x, tiles_m, tiles_n, tiles_m, tm, tn);      // consecutive blocks of an XCD: the row tiles of ONE column tile (they share its W rows)
    const int m0 = tm * BM, n0 = tn * BN;
    const unsigned lds_u = (unsigned)(uintptr_t)(lds_ptr_t)ring;
    const unsigned sink = lds_u + SINK;                         // 1 KiB nobody reads: destination of the absent / out-of-range pieces

    // ---- sources of this wave's pieces (slab 0), LDS offsets inside a slab.  Row-major operands: rows lda / K elements apart, K slabs 128 B
    // (fp8: 64 B) apart; slab-major operands ([K/64][rows][64], vcla_gemm_args.A_slab / W_slab / W_q8_slab): rows 128 B apart, slabs rows * 128 B
    // apart -- a piece's 8 (16) rows are then ONE contiguous 1 KiB (measured: +2 - 8 % here, profiles/r05_ring_microbench.txt)
    const char* Ab = (const char*)(a.A_slab ? a.A_slab : a.A);
    const int64_t a_rs = a.A_slab ? 128 : a.lda * 2, a_ss = a.A_slab ? a.a_slab_rows * 128 : 128;
    constexpr int W_ROW = W8 ? 64 : 128;                      // bytes of one weight row inside a K slab
    const bool wslab = W8 ? a.W_q8_slab != nullptr : a.W_slab != nullptr;
    const char* Wb = (const char*)(WF ? a.W_frag : (W8 ? (wslab ? a.W_q8_slab : a.W_q8) : (wslab ? a.W_slab : a.W)));
    const int64_t w_rs = wslab ? W_ROW : (int64_t)a.K * (W8 ? 1 : 2), w_ss = WF ? 2048 : (wslab ? (int64_t)n_pad * W_ROW : W_ROW);
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
        if constexpr (WF) {
            int t16 = n0 / 16 + (qq >> 1);                        // piece qq = fragment (16-row tile qq / 2, k-step qq % 2) of the slab
            t16 = t16 < n_pad / 16 ? t16 : n_pad / 16 - 1;
            wsrc[i] = Wb + ((int64_t)t16 * (a.K / 32) + (qq & 1)) * 1024 + lane * 16;
        } else if constexpr (W8) {
            const int row = qq * 16 + (lane >> 2);                // 16 rows x 64 B per piece, stored LINEAR (the widening pass reads 16 B per lane in order)
            int wr = n0 + row;
            wr = wr < n_pad ? wr : n_pad - 1;
            wsrc[i] = Wb + (int64_t)wr * w_rs + (lane & 3) * 16;
        } else {
            const int row = qq * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            int wr = n0 + row;
            wr = wr < n_pad ? wr : n_pad - 1;
            wsrc[i] = Wb + (int64_t)wr * w_rs + chunk * 16;
        }
    }
    const int nslab = a.K / GM_BK;
    // issue group j = {activations of stage j, weights of stage j + WLEAD}: PP instructions, always, and BRANCH-FREE: a piece that is absent (waves past the
    // last weight piece) or lies outside [0, nslab) is still one 1-KiB DMA -- of a valid address (slab 0 of the wave's own piece) into the SINK.  (The first
    // form branched around 4-byte dummies: ~170 scalar instructions per K slab in the 64 x 64 tile against ~75 now.  Measured EQUAL -- o_proj 23.7 vs 24.3 us,
    // down_proj 55.8 vs 58.1 -- as were 9 single-slab stages against 4 double-slab ones (128 vs 96 KiB in flight) and slab-major operands: the 64 x 64 tile
    // sits at ~20 B/clk/CU whichever of issue overhead, ring depth or access shape is relieved ALONE, profiles/r05_ring_microbench.txt.)  buf_a / buf_w: the
    // ring buffers of the two stages.
    auto issue = [&](int j, int buf_a, int buf_w) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int slab_a = j * KS + k, slab_w = (j + WLEAD) * KS + k;
            const bool live_a = j >= 0 && slab_a < nslab, live_w = slab_w < nslab;
            const int64_t oa = live_a ? (int64_t)slab_a * a_ss : 0, ow = live_w ? (int64_t)slab_w * w_ss : 0;
            const unsigned ab = lds_u + buf_a * A_STAGE + k * A_BYTES, wb = lds_u + W_RING + buf_w * W_STAGE + k * W_BYTES;
#pragma unroll
            for (int i = 0; i < IA; ++i) gr_dma16(asrc[i] + oa, live_a ? ab + (unsigned)(wave + 8 * i) * 1024u : sink);
#pragma unroll
            for (int i = 0; i < IW; ++i) gr_dma16(wsrc[i] + ow, (live_w && wreal[i]) ? wb + (unsigned)(wave + 8 * i) * 1024u : sink);
        }
    };
    // W8: widen this wave's share of the e4m3 weights in W-ring buffer `buf_w` into bf16 image `img` (unit u = 16 bytes = 16 weights of one row:
    // (slab, row, 16-k chunk c) -> the two 16-byte chunks 2c, 2c + 1 of that row in the [rows][128 B] image the fragment reads use)
    auto widen = [&](int buf_w, int img) {
        if constexpr (W8) {
            constexpr int UNITS = KS * BN * 4;
            const unsigned char* src = ring + W_RING + buf_w * W_STAGE;
            unsigned char* dst = ring + W_BF + img * (KS * WB_BYTES);
#pragma unroll
            for (int p = 0; p < (UNITS + 511) / 512; ++p) {
                const int u = p * 512 + wave * 64 + lane;
                if (u < UNITS) {
                    const int k = u / (BN * 4), r = (u / 4) % BN, c = u & 3;
                    const u32x4_t raw = *reinterpret_cast<const u32x4_t*>(src + u * 16);
                    unsigned char* row = dst + k * WB_BYTES;
                    *reinterpret_cast<bf16x8_t*>(row + lds_off(r, 2 * c)) = gr_fp8x8_to_bf16x8(raw.x, raw.y);
                    *reinterpret_cast<bf16x8_t*>(row + lds_off(r, 2 * c + 1)) = gr_fp8x8_to_bf16x8(raw.z, raw.w);
                }
            }
        }
    };

    f32x4_t acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fch = lane >> 4;
    const int nst = (nslab + KS - 1) / KS;

    // ---- prologue: groups -WLEAD .. NS - 2 (W8: group -1 carries the weights of stage 0 alone, which are widened before the loop)
    int buf_a = 0, buf_w = 0;                                   // ring buffers of the NEXT group to issue
    auto step = [&](int& b, int n) { b = b + 1 == n ? 0 : b + 1; };
    if constexpr (W8) { issue(-1, 0, buf_w); step(buf_w, NS + WLEAD); }
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) { issue(s, buf_a, buf_w); step(buf_a, NS); step(buf_w, NS + WLEAD); }
    if constexpr (W8) {
        gr_vmcnt<(NS - 1) * PP>();                              // group -1 has landed (NS - 1 younger groups may not have)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        widen(0, 0);
    }
    int cur_a = 0, cur_w = WLEAD;                               // A ring buffer of stage st; W ring buffer of stage st + WLEAD
    for (int st = 0; st < nst; ++st) {
        gr_vmcnt<(NS - 2) * PP>();                              // this wave's pieces of group st have landed (NS - 2 younger groups may not have)
        __builtin_amdgcn_s_barrier();                          // ... everyone's; the buffers of stage st - 1 are no longer read; (W8) the bf16 image of stage st is complete
        asm volatile("" ::: "memory");
        issue(st + NS - 1, buf_a, buf_w);
        step(buf_a, NS); step(buf_w, NS + WLEAD);
        if constexpr (W8) widen(cur_w, (st + 1) & 1);           // weights of stage st + 1 -> the other image (read after the NEXT barrier)
        const unsigned char* Ab_s = ring + cur_a * A_STAGE;
        const unsigned char* Wb_s = W8 ? ring + W_BF + (st & 1) * (KS * WB_BYTES) : ring + W_RING + cur_w * W_STAGE;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            if (KS > 1 && st * KS + k >= nslab) break;          // K tail of a multi-slab stage (wave-uniform)
            const unsigned char* As = Ab_s + k * A_BYTES;
            const unsigned char* Ws = Wb_s + k * WB_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8_t wf[NJ], af[MI];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if constexpr (WF) wf[j] = *reinterpret_cast<const bf16x8_t*>(Ws + ((wn * (BN / WN) / 16 + j) * 2 + kk) * 1024 + lane * 16);
                    else wf[j] = *reinterpret_cast<const bf16x8_t*>(Ws + lds_off(wn * (BN / WN) + j * 16 + frow, kk * 4 + fch));
                }
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(As + lds_off(wm * (BM / WM) + i * 16 + frow, kk * 4 + fch));
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("" ::: "memory");                          // the fragment reads / image writes stay on this side of the next barrier
        step(cur_a, NS); step(cur_w, NS + WLEAD);
    }
    gr_vmcnt<0>();                                              // no DMA (the tail's dummies) may land in LDS after the workgroup has given it up
    gemm_epilogue<EPI, OutT, MI, NJ, W8>(a, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);
}

// ------------------------------------------------------------------ host side
template <int EPI, typename OutT, int BM, int BN, int WM, int WN, int NS, int KS, bool W8, bool WF = false>
static int launch_ring_cfg(const vcla_gemm_args* a, hipStream_t s) {
    if constexpr (!W8 && !WF && sizeof(OutT) == 2) {          // bf16 output, bf16 weights with a fragment-major twin: its pieces are contiguous KiBs (VCLA_RING_WF=0: row-major pieces)
        static const int wf_env = getenv("VCLA_RING_WF") ? atoi(getenv("VCLA_RING_WF")) : 1;
        if (wf_env && a->W_frag && !a->W_slab) return launch_ring_cfg<EPI, OutT, BM, BN, WM, WN, NS, KS, false, true>(a, s);
    }
    constexpr size_t lds = (size_t)NS * KS * BM * 128 + (size_t)(NS + (W8 ? 1 : 0)) * KS * (W8 ? BN * 64 : BN * 128) + (W8 ? (size_t)2 * KS * BN * 128 : 0) + 1024;
    static_assert(lds <= 160 * 1024, "ring exceeds the 160 KiB of a CU");
    auto kern = gemm_ring_kernel<EPI, OutT, BM, BN, WM, WN, NS, KS, W8, WF>;
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
    // NS: as many stages as 160 KiB hold (fp8 weights: their ring is one stage deeper and two bf16 images of a stage sit beside it)
    const int cfg = ring_cfg(a);
    if (cfg == 1) return launch_ring_cfg<EPI, OutT, 256, 96, 8, 1, 3, 1, W8>(a, s);
    if constexpr (EPI == VCLA_EPI_SWIGLU) return vcla_fail(VCLA_ERR_BAD_ARG, "gemm: the SwiGLU ring tile is 256 x 96");
    else {
        if (cfg == 2) return launch_ring_cfg<EPI, OutT, 128, 96, 4, 2, 5, 1, W8>(a, s);
        if constexpr (!W8) {      // bf16 weights: 9 single-slab stages (128 KiB in flight) instead of 4 double-slab ones (96 KiB): measured equal (+2 %); VCLA_RING_C3=0: the first form
            static const int c3_env = getenv("VCLA_RING_C3") ? atoi(getenv("VCLA_RING_C3")) : 1;
            if (c3_env) return launch_ring_cfg<EPI, OutT, 64, 64, 4, 2, 9, 1, W8>(a, s);
        }
        return launch_ring_cfg<EPI, OutT, 64, 64, 4, 2, 4, 2, W8>(a, s);
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
    // one image through the ViT / the resampler (257 or 65 rows, K = 1024): bias + activation in the tile's own epilogue instead of K slices + a reduce launch
    if (!w8 && !a->out_f32 && a->epilogue == VCLA_EPI_QUICK_GELU) return launch_ring<VCLA_EPI_QUICK_GELU, bf16_t, false>(a, s);
    if (!w8 && !a->out_f32 && a->epilogue == VCLA_EPI_GELU_ERF) return launch_ring<VCLA_EPI_GELU_ERF, bf16_t, false>(a, s);
    return vcla_fail(VCLA_ERR_BAD_ARG, "gemm: the ring kernel implements epilogues NONE (bf16 / fp32 output), SWIGLU (bf16 output) and the two GELUs (bf16 weights and output), got %d", a->epilogue);
}
