// gemm.hip -- C = epilogue(A . W^T + bias) (+ residual) for every Linear on the VisualCLA path.
//
//   A [M, K]  activations (bf16 or fp32), row stride lda
//   W [N_pad, K] bf16 weights, K contiguous (both MFMA operands are read as 8 contiguous k = one 16-byte LDS read)
//
// Three kernels:
//   gemm_mfma_kernel   bf16 x bf16 -> fp32 on v_mfma_f32_16x16x32_bf16; 128x128x64 workgroup tile, 4 waves
//                      (2x2, 64x64 per wave = 4x4 MFMA tiles), LDS double buffer with an XOR swizzle that makes
//                      the ds_read_b128 fragment reads conflict-free, global->register prefetch of tile k+1
//                      under the MFMAs of tile k, XCD-aware tile order.  MFMA operands are swapped (W rows feed
//                      the A port) so each lane ends up with 4 CONSECUTIVE output columns of one row -> 8/16-byte
//                      epilogue stores, and the SwiGLU gate/up pair (16 packed rows apart) sits in the same lane.
//   gemv_kernel        M <= 8 (decode): weight rows streamed once straight into registers with 16-byte
//                      non-temporal loads, 4 rows per wave, wave-shuffle reduction.  HBM-bound by design.
//   gemm_f32_kernel    fp32 activations (parity mode): classic 64x64x16 LDS-tiled FMA kernel.
#include "vcla_common.h"
#include "gemm_epilogue.h"
#include <type_traits>
#include <stdlib.h>

int vcla_gemm_dstream_launch(const vcla_gemm_args* a, hipStream_t s);   // gemm_stream.hip (kernel 9)

#include "gemm_tiles.h"
int vcla_gemm_mfma256_launch(const vcla_gemm_args* a, bool sgb, hipStream_t s);     // gemm_mfma256.hip (kernels 4 / 5)
int vcla_gemm_mfma256_fp8_launch(const vcla_gemm_args* a, hipStream_t s);          // gemm_mfma256.hip (kernel 10)
int vcla_gemm_ring_launch(const vcla_gemm_args* a, hipStream_t s);                 // gemm_ring.hip (kernel 11)
bool vcla_gemm_tile257_ok(const vcla_gemm_args* a);                                // 257-row tiles apply (M = B * 257)

// =================================================================== MFMA kernel

// SPLIT: blockIdx.y = K slice; the raw fp32 accumulators of the slice go to partial[ks][m][n_pad] and gemm_panel_reduce_kernel sums
// the slices in order and applies the epilogue.  For problems of a few dozen tiles (the ViT / resampler GEMMs of a single image:
// M = 257 -> 24 - 96 tiles for 256 CUs, 16 - 64 serial K steps each) -- see launch_mfma.
template <int EPI, typename OutT, bool SPLIT = false>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(vcla_gemm_args a, int tiles_m, int tiles_n, int splitk = 1, int n_pad = 0,
                                                        float* __restrict__ partial = nullptr) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][GM_BM * GM_BK * 2];  // [buf][A|W][16 KiB]

    int tm, tn;
    tile_assign(blockIdx.x, tiles_m, tiles_n, 8, tm, tn);
    const int m0 = tm * GM_BM, n0 = tn * GM_BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- global -> register staging: 4 x 16 B per operand per thread
    const int nk_all = a.K / GM_BK;
    const int ks = SPLIT ? (int)blockIdx.y : 0;
    const int k_beg = SPLIT ? (int)((int64_t)ks * nk_all / splitk) : 0, k_end = SPLIT ? (int)((int64_t)(ks + 1) * nk_all / splitk) : nk_all;
    const bf16_t* Ag = (const bf16_t*)a.A + (int64_t)k_beg * GM_BK;
    const bf16_t* Wg = (const bf16_t*)a.W + (int64_t)k_beg * GM_BK;
    const bf16_t* aptr[4];
    const bf16_t* wptr[4];
    int soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = i * 256 + tid;
        const int row = id >> 3, ch = id & 7;
        int am = m0 + row;
        if (am >= a.M) am = a.M - 1;  // clamp: rows past M are computed on valid memory and never stored
        aptr[i] = Ag + (int64_t)am * a.lda + ch * 8;
        wptr[i] = Wg + (int64_t)(n0 + row) * a.K + ch * 8;  // W is padded to a multiple of 128 rows
        soff[i] = lds_off(row, ch);
    }
    uint4 ra[4], rw[4];
    const int nk = k_end - k_beg;                    // >= 1 (the launcher keeps splitk <= K / 64)

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const uint4*>(aptr[i]);
        rw[i] = *reinterpret_cast<const uint4*>(wptr[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<uint4*>(&lds[0][0][soff[i]]) = ra[i];
        *reinterpret_cast<uint4*>(&lds[0][1][soff[i]]) = rw[i];
    }
    __syncthreads();

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: lane l reads row (l & 15) of a 16-row sub-tile, k-chunk (l >> 4) (+4 for the 2nd K=32 step)
    const int frow = lane & 15, fch = lane >> 4;

    auto compute_tile = [&](int cur) {
        const unsigned char* As = lds[cur][0];
        const unsigned char* Ws = lds[cur][1];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ar = wm * 64 + i * 16 + frow;
                af[i] = *reinterpret_cast<const bf16x8_t*>(As + lds_off(ar, kk * 4 + fch));
                const int wr = wn * 64 + i * 16 + frow;
                wf[i] = *reinterpret_cast<const bf16x8_t*>(Ws + lds_off(wr, kk * 4 + fch));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    // steady state: prefetch tile kt+1 into registers, MFMA tile kt from LDS, then park the prefetch in the
    // other LDS buffer; one barrier per K tile.  The last tile is peeled so the loop body is branch-free.
    for (int kt = 0; kt < nk - 1; ++kt) {
        const int cur = kt & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const uint4*>(aptr[i] + (int64_t)(kt + 1) * GM_BK);
            rw[i] = *reinterpret_cast<const uint4*>(wptr[i] + (int64_t)(kt + 1) * GM_BK);
        }
        compute_tile(cur);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<uint4*>(&lds[cur ^ 1][0][soff[i]]) = ra[i];
            *reinterpret_cast<uint4*>(&lds[cur ^ 1][1][soff[i]]) = rw[i];
        }
        __syncthreads();
    }
    compute_tile((nk - 1) & 1);

    if constexpr (SPLIT) {
        // acc[i][j][r] = C[m][n], m = mw + i*16 + (lane & 15), n = nw + j*16 + (lane >> 4)*4 + r (operands swapped, see gemm_epilogue.h)
        const int mw = m0 + wm * 64, nw = n0 + wn * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mw + i * 16 + (lane & 15);
            if (m >= a.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = nw + j * 16 + (lane >> 4) * 4;      // < n_pad: W is padded to 128 rows
                *reinterpret_cast<float4*>(partial + ((int64_t)ks * a.M + m) * n_pad + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
    } else {
        gemm_epilogue<EPI, OutT, 4>(a, acc, m0 + wm * 64, n0 + wn * 64, lane);
    }
}

// 16-byte non-temporal weight load: streamed-once data should not displace the L2-resident activations
__device__ __forceinline__ uint4 ldg_nt(const bf16_t* p) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
}

// =================================================================== skinny MFMA kernel (2 <= M <= 128): batch decode, short prefill
// HBM-bound by construction: W is streamed exactly once, straight from global memory into MFMA operand registers (no LDS
// round trip: a weight element is used by one wave only).  Workgroup = WPB waves, output tile = all M rows x 32 columns;
// the waves split K between them (intra-workgroup split-K), each wave accumulates MT x 2 MFMA tiles over its K slice, the
// partial tiles are reduced through LDS in a fixed tree order (deterministic), wave 0 runs the shared epilogue.
// A (activations, M x K, L2-resident) is read as MFMA fragments directly from global/L2.
#define SK_BN 32
template <int EPI, typename OutT, int MT, int WPB>
__global__ __launch_bounds__(WPB * 64) void gemm_skinny_kernel(vcla_gemm_args a) {
    __shared__ __attribute__((aligned(16))) float4 part[(WPB / 2) * MT * 2 * 64];  // [wave][tile][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * SK_BN;
    const int frow = lane & 15, g = lane >> 4;
    const int Kw = a.K / WPB;          // K slice of this wave (multiple of 32, checked on the host)
    const int kbeg = wave * Kw;
    const bf16_t* Ag = (const bf16_t*)a.A;
    const bf16_t* Wg = (const bf16_t*)a.W;
    const bf16_t* wp[2];
    const bf16_t* ap[MT];
#pragma unroll
    for (int j = 0; j < 2; ++j) wp[j] = Wg + (int64_t)(n0 + j * 16 + frow) * a.K + kbeg + g * 8;  // rows < N_pad (128-row padding)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int m = i * 16 + frow;
        if (m >= a.M) m = a.M - 1;     // clamp: garbage rows are never stored
        ap[i] = Ag + (int64_t)m * a.lda + kbeg + g * 8;
    }
    f32x4_t acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    constexpr int UN = 4;              // k-steps of 32 issued together (8 weight loads of 16 B in flight per lane)
    int k = 0;
    for (; k + 32 * UN <= Kw; k += 32 * UN) {
        uint4 wv[UN][2];
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int j = 0; j < 2; ++j) wv[u][j] = ldg_nt(wp[j] + k + u * 32);
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            bf16x8_t af[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(ap[i] + k + u * 32);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wv[u][j]), af[i], acc[i][j], 0, 0, 0);
        }
    }
    for (; k < Kw; k += 32) {
        uint4 wv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) wv[j] = ldg_nt(wp[j] + k);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(ap[i] + k);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wv[j]), af, acc[i][j], 0, 0, 0);
        }
    }
    // ---- fixed-order tree reduction over the waves: the upper half stores, the lower half adds
#pragma unroll
    for (int half = WPB / 2; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    part[((wave - half) * MT * 2 + i * 2 + j) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
        __syncthreads();
        if (wave < half) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 p = part[(wave * MT * 2 + i * 2 + j) * 64 + lane];
                    acc[i][j][0] += p.x; acc[i][j][1] += p.y; acc[i][j][2] += p.z; acc[i][j][3] += p.w;
                }
        }
        __syncthreads();
    }
    if (wave == 0) gemm_epilogue<EPI, OutT, MT, 2>(a, acc, 0, n0, lane);
}

template <int EPI, typename OutT, int MT>
static int launch_skinny_mt(const vcla_gemm_args* a, hipStream_t s) {
    const int blocks = (a->N + SK_BN - 1) / SK_BN;
    // LDS for the partial tiles: (WPB/2) * MT * 2 KiB; MT = 8 fits 64 KiB only with 8 waves -> always legal
    if (a->K % (32 * 8) == 0) gemm_skinny_kernel<EPI, OutT, MT, 8><<<blocks, 512, 0, s>>>(*a);
    else gemm_skinny_kernel<EPI, OutT, MT, 2><<<blocks, 128, 0, s>>>(*a);   // K % 64 == 0 always holds
    VCLA_CHECK_LAUNCH("gemm_skinny_kernel");
    return VCLA_OK;
}

template <int EPI, typename OutT>
static int launch_skinny(const vcla_gemm_args* a, hipStream_t s) {
    if (a->M <= 16) return launch_skinny_mt<EPI, OutT, 1>(a, s);
    if (a->M <= 32) return launch_skinny_mt<EPI, OutT, 2>(a, s);
    if (a->M <= 64) return launch_skinny_mt<EPI, OutT, 4>(a, s);
    if (a->M <= 128) return launch_skinny_mt<EPI, OutT, 8>(a, s);
    return vcla_fail(VCLA_ERR_BAD_SHAPE, "gemm: skinny kernel needs M <= 128 (got %d)", a->M);
}

// =================================================================== panel MFMA kernel (M <= 128, batch decode), split-K over workgroups
// Workgroup = 4 waves = all M rows x 128 columns of one K slice; wave w owns columns 32w..32w+31 (2 MFMA n-tiles) and streams
// ITS weight rows straight from HBM into MFMA operand registers (a weight element is needed by one wave only), 4 K-tiles
// ahead.  The activation panel A[M, 64] of each K tile is shared by the 4 waves: coalesced full-line loads (4 tiles ahead,
// register ring) -> swizzled LDS double buffer -> ds_read_b128 fragments.  Every load has the same 4-tile distance, so the
// in-order vmcnt queue never forces an early wait.  grid = (N/128, S): S K-slices keep >= ~300 workgroups in flight; S > 1
// writes fp32 partial tiles to a workspace and gemm_panel_reduce_kernel applies the epilogue (fixed summation order).
// (Measured alternative, round 1: reducing inside the launch -- last-arriver ticket, agent-scope release fence per
// workgroup -- made the kernel 20 us slower (23.9 -> 44.7 us at M=64): the per-workgroup L2 write-back costs far more
// than the ~5 us second launch.  Kept as two launches.
// Also measured and dropped (M = 64, the four LLaMA decode GEMMs of a layer, 128 us total with this kernel): an 8-deep
// register ring (154 us: 342 VGPRs -> 1 workgroup per CU), 512 / 768 workgroups via more K slices (156 / 158 us: the fp32
// partial traffic grows faster than the latency hiding), and a stream-K launch that deals exactly 2 equal runs of K tiles to
// every CU (142 us).  PMC (profiles/r01_pmc_panel_m64.txt): no LDS bank conflicts, HBM reads = algorithmic bytes, MFMA
// busy 10 % -- what is left is per-launch ramp (~7 us on 12-44 us kernels) and the partial round trip.)
#define PN_BN 128
#define VCLA_POST_NORM_DONE (-12345)   // internal: the launcher already produced post_norm_out
#define PN_RING 4
// 8 OCP fp8 (e4m3fn) values in two dwords -> one bf16x8 MFMA operand (exact: e4m3 fits in bf16)
__device__ __forceinline__ bf16x8_t fp8x8_to_bf16x8(uint32_t lo, uint32_t hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
    const f32x2_t c = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
    const u32x4_t p = {pack_bf2(a.x, a.y), pack_bf2(b.x, b.y), pack_bf2(c.x, c.y), pack_bf2(d.x, d.y)};
    return __builtin_bit_cast(bf16x8_t, p);
}

// WMODE: 0 = W row-major bf16, 1 = fragment-major bf16 (W_frag), 2 = fragment-pair-major fp8 (W_q8_frag + w_scale)
// KG = K groups per workgroup (1 or 2).  KG = 2: 8 waves; waves 4..7 repeat the column assignment of waves 0..3 on the second
// half of the workgroup's K slice (own A buffers), the two accumulator sets meet in LDS and group 0 runs the epilogue.  Twice
// the waves (= loads in flight) per CU at the same grid, half the split-K partials for the same number of K streams: the
// launch can then stay at <= 256 workgroups (one per CU, no second round) and the big-N GEMMs need no partials at all.
template <int EPI, typename OutT, int MT, int WMODE, int KG = 1>
__global__ __launch_bounds__(256 * KG, (MT == 8 && KG == 1) ? 2 : 1) void gemm_panel_kernel(vcla_gemm_args a, int splitk, int n_pad, float* __restrict__ partial) {
    constexpr int NA = (MT * 128 + 255) / 256;  // 16-byte A chunks per thread per K tile
    __shared__ __attribute__((aligned(16))) unsigned char As_all[KG][2][MT * 16 * 128];
    const int kg = KG == 1 ? 0 : (int)(threadIdx.x >> 8);
    auto& As = As_all[kg];
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * PN_BN, ks = blockIdx.y;
    const int nk = a.K / GM_BK;
    const int s_beg = (int)((int64_t)ks * nk / splitk), s_end = (int)((int64_t)(ks + 1) * nk / splitk);
    // this K group's part of the slice; the loop trip count follows group 0 (the longer one), group 1 pads with zero tiles
    const int len0 = KG == 1 ? s_end - s_beg : (s_end - s_beg + 1) / 2;
    const int t_beg = kg == 0 ? s_beg : s_beg + len0;
    const int nkc = kg == 0 ? len0 : (s_end - s_beg) - len0;     // K tiles of this group (group 1 may have 0)
    const int nkc_loop = len0;
    const bf16_t* Ag = (const bf16_t*)a.A + (int64_t)t_beg * GM_BK;
    const bf16_t* Wg = (const bf16_t*)a.W + (int64_t)t_beg * GM_BK;

    // A staging map: chunk id -> (row, 16-byte chunk); rows past M are clamped (never stored).  Every thread always loads
    // (ids past the tile wrap around) so the register ring stays branch-free; only the LDS store is predicated.
    auto a_src = [&](int i) {
        const int id = i * 256 + tid;
        const int row = (id >> 3) % (MT * 16), ch = id & 7;
        const int am = row < a.M ? row : a.M - 1;
        return Ag + (int64_t)am * a.lda + ch * 8;
    };
    auto a_dst = [&](int i) {
        const int id = i * 256 + tid;
        return lds_off((id >> 3) % (MT * 16), id & 7);
    };
    const bf16_t *asrc0 = a_src(0), *asrc1 = a_src(1), *asrc2 = a_src(2), *asrc3 = a_src(3);
    const int adst0 = a_dst(0), adst1 = a_dst(1), adst2 = a_dst(2), adst3 = a_dst(3);
    // W fragment rows of this wave: 2 n-tiles; rows past N stay inside the 128-row padding, past n_pad are clamped
    // FRAG: W_frag[n/16][k/32][lane][8] -- the fragment of (16-row tile, k-step) is 1 KiB contiguous, k-steps are adjacent:
    //       this wave streams two fully contiguous regions; per-lane address = tile base + kstep * 512 + lane * 8 elements
    const bf16_t* wsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (WMODE == 1) {
            int nt = (n0 + wave * 32 + j * 16) >> 4;
            if (nt >= (n_pad >> 4)) nt = (n_pad >> 4) - 1;
            wsrc[j] = (const bf16_t*)a.W_frag + ((int64_t)nt * (a.K / 32) + (int64_t)t_beg * 2) * 512 + lane * 8;
        } else if (WMODE == 2) {
            // fp8: one 1 KiB block per (16-row tile, 64 k); pointer arithmetic in bf16_t units (2 bytes)
            int nt = (n0 + wave * 32 + j * 16) >> 4;
            if (nt >= (n_pad >> 4)) nt = (n_pad >> 4) - 1;
            wsrc[j] = (const bf16_t*)a.W_q8_frag + ((int64_t)nt * (a.K / 64) + (int64_t)t_beg) * 512 + lane * 8;
        } else {
            int wr = n0 + wave * 32 + j * 16 + frow;
            if (wr >= n_pad) wr = n_pad - 1;
            wsrc[j] = Wg + (int64_t)wr * a.K + g * 8;
        }
    }
    constexpr int WTILE = WMODE == 1 ? 1024 : (WMODE == 2 ? 512 : GM_BK);  // bf16_t units between consecutive K tiles
    constexpr int WSTEP = WMODE == 1 ? 512 : 32;                           // ... between the two k-steps of a tile
    constexpr int NWL = WMODE == 2 ? 2 : 4;                                // 16-byte weight loads per lane per K tile
    // 4-slot register ring, one named array per slot: slot indices must be literals for the compiler to keep the ring in
    // VGPRs (a ring indexed through a lambda parameter is demoted to scratch).
    struct AReg { u32x4_t c0, c1, c2, c3; };  // up to 4 chunks per thread; unused members are never touched (NA < 4)
    AReg ra0, ra1, ra2, ra3;
    u32x4_t rw0[4], rw1[4], rw2[4], rw3[4];  // [kk*2 + j]
    // tiles past the end of the slice fetch nothing and get ZEROED operands (the MFMAs then add 0): the slice length need
    // not be a multiple of 4.  (Re-loading the last tile instead cost ~5 %: 135 -> 128 us
    // for the four M = 64 decode GEMMs of a layer.)
#define PN_LOAD(tile_, RA, RW)                                                                      \
    {                                                                                               \
        if ((tile_) < nkc) { /* wave-uniform: past the end nothing is fetched, the slot just holds zero weights */ \
            const int64_t ko_ = (int64_t)(tile_) * GM_BK;                                           \
            RA.c0 = *reinterpret_cast<const u32x4_t*>(asrc0 + ko_);                                   \
            if (NA >= 2) RA.c1 = *reinterpret_cast<const u32x4_t*>(asrc1 + ko_);                      \
            if (NA >= 4) { RA.c2 = *reinterpret_cast<const u32x4_t*>(asrc2 + ko_); RA.c3 = *reinterpret_cast<const u32x4_t*>(asrc3 + ko_); } \
            _Pragma("unroll") for (int q = 0; q < NWL; ++q)                                         \
                RW[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wsrc[q & 1] + (int64_t)(tile_) * WTILE + (q >> 1) * WSTEP)); \
        } else { /* zero BOTH operands: an uninitialised A register may hold NaN bits, and 0 * NaN = NaN */ \
            RA.c0 = u32x4_t{0u, 0u, 0u, 0u};                                                        \
            if (NA >= 2) RA.c1 = u32x4_t{0u, 0u, 0u, 0u};                                           \
            if (NA >= 4) { RA.c2 = u32x4_t{0u, 0u, 0u, 0u}; RA.c3 = u32x4_t{0u, 0u, 0u, 0u}; }      \
            _Pragma("unroll") for (int q = 0; q < NWL; ++q) RW[q] = u32x4_t{0u, 0u, 0u, 0u};        \
        }                                                                                           \
    }
#define PN_STORE(RA, buf_)                                                                          \
    {                                                                                               \
        if (MT * 128 >= 256 || tid < MT * 128) *reinterpret_cast<u32x4_t*>(&As[buf_][adst0]) = RA.c0; \
        if (NA >= 2) *reinterpret_cast<u32x4_t*>(&As[buf_][adst1]) = RA.c1;                           \
        if (NA >= 4) { *reinterpret_cast<u32x4_t*>(&As[buf_][adst2]) = RA.c2; *reinterpret_cast<u32x4_t*>(&As[buf_][adst3]) = RA.c3; } \
    }
#define PN_COMPUTE(RW, cur_)                                                                        \
    {                                                                                               \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                          \
            bf16x8_t w0_, w1_;                                                                      \
            if (WMODE == 2) { /* RW[j] = 16 fp8 of n-tile j: .xy -> k-step 0, .zw -> k-step 1 */    \
                w0_ = kk == 0 ? fp8x8_to_bf16x8(RW[0].x, RW[0].y) : fp8x8_to_bf16x8(RW[0].z, RW[0].w); \
                w1_ = kk == 0 ? fp8x8_to_bf16x8(RW[1].x, RW[1].y) : fp8x8_to_bf16x8(RW[1].z, RW[1].w); \
            } else {                                                                                \
                w0_ = __builtin_bit_cast(bf16x8_t, RW[kk * 2 + 0]);                                 \
                w1_ = __builtin_bit_cast(bf16x8_t, RW[kk * 2 + 1]);                                 \
            }                                                                                       \
            _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                        \
                const bf16x8_t af_ = *reinterpret_cast<const bf16x8_t*>(&As[cur_][lds_off(i * 16 + frow, kk * 4 + g)]); \
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0_, af_, acc[i][0], 0, 0, 0);  \
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1_, af_, acc[i][1], 0, 0, 0);  \
            }                                                                                       \
        }                                                                                           \
    }
    f32x4_t acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    PN_LOAD(0, ra0, rw0) PN_LOAD(1, ra1, rw1) PN_LOAD(2, ra2, rw2) PN_LOAD(3, ra3, rw3)
    PN_STORE(ra0, 0)
    __syncthreads();
    const int nk4 = (nkc_loop + PN_RING - 1) / PN_RING * PN_RING;
    for (int kt = 0; kt < nk4; kt += PN_RING) {
        // tile kt+t: MFMAs from LDS buffer t&1 and ring slot t; park tile kt+t+1 (loaded 3 steps ago) in the other LDS
        // buffer; refill slot t with tile kt+t+4; one barrier per tile
        PN_COMPUTE(rw0, 0) PN_STORE(ra1, 1) PN_LOAD(kt + 4, ra0, rw0) __syncthreads();
        PN_COMPUTE(rw1, 1) PN_STORE(ra2, 0) PN_LOAD(kt + 5, ra1, rw1) __syncthreads();
        PN_COMPUTE(rw2, 0) PN_STORE(ra3, 1) PN_LOAD(kt + 6, ra2, rw2) __syncthreads();
        PN_COMPUTE(rw3, 1) PN_STORE(ra0, 0) PN_LOAD(kt + 7, ra3, rw3) __syncthreads();
    }
#undef PN_LOAD
#undef PN_STORE
#undef PN_COMPUTE
    if constexpr (KG == 2) {
        // group 1 hands its accumulators over through LDS (the A buffers are free after the last barrier): [i][j][thread]
        static_assert(KG == 1 || sizeof(As_all) >= (size_t)MT * 2 * 256 * 16, "accumulator hand-over must fit the A buffers");
        f32x4_t* xch = reinterpret_cast<f32x4_t*>(&As_all[0][0][0]);
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) xch[(i * 2 + j) * 256 + tid] = acc[i][j];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4_t o = xch[(i * 2 + j) * 256 + tid];
                acc[i][j][0] += o[0]; acc[i][j][1] += o[1]; acc[i][j][2] += o[2]; acc[i][j][3] += o[3];
            }
    }
    if (splitk == 1) {
        gemm_epilogue<EPI, OutT, MT, 2>(a, acc, 0, n0 + wave * 32, lane);
    } else {
        // fp32 partial tile: partial[ks][m][n], 4 consecutive columns per lane (16-byte stores)
        float* pp = partial + (int64_t)ks * a.M * n_pad;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = i * 16 + frow;
            if (m >= a.M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wave * 32 + j * 16 + g * 4;
                if (n < n_pad)
                    *reinterpret_cast<float4*>(pp + (int64_t)m * n_pad + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
    }
}

// sum the S partial tiles in slice order, then bias / activation / SwiGLU / residual / store (4 columns per thread)
template <int EPI, typename OutT>
__global__ __launch_bounds__(256) void gemm_panel_reduce_kernel(vcla_gemm_args a, int splitk, int n_pad, const float* __restrict__ partial) {
    const int n_out = (EPI == VCLA_EPI_SWIGLU) ? a.N / 2 : a.N;
    const int groups = (n_out + 3) / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)a.M * groups) return;
    const int m = (int)(idx / groups), n = (int)(idx % groups) * 4;
    float v[4];
    if constexpr (EPI == VCLA_EPI_SWIGLU) {
        const int np_ = (n >> 4) * 32 + (n & 15);  // packed gate column; up = +16
        float gt[4] = {0, 0, 0, 0}, up[4] = {0, 0, 0, 0};
        for (int s = 0; s < splitk; ++s) {
            const float* pp = partial + ((int64_t)s * a.M + m) * n_pad + np_;
            const float4 x = *reinterpret_cast<const float4*>(pp), y = *reinterpret_cast<const float4*>(pp + 16);
            gt[0] += x.x; gt[1] += x.y; gt[2] += x.z; gt[3] += x.w;
            up[0] += y.x; up[1] += y.y; up[2] += y.z; up[3] += y.w;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (a.w_scale) { gt[r] *= a.w_scale[np_ + r]; up[r] *= a.w_scale[np_ + 16 + r]; }
            if (a.bias) { gt[r] += a.bias[np_ + r]; up[r] += a.bias[np_ + 16 + r]; }
            v[r] = act_silu(gt[r]) * up[r];
        }
    } else {
        float sacc[4] = {0, 0, 0, 0};
        for (int s = 0; s < splitk; ++s) {
            const float4 x = *reinterpret_cast<const float4*>(partial + ((int64_t)s * a.M + m) * n_pad + n);
            sacc[0] += x.x; sacc[1] += x.y; sacc[2] += x.z; sacc[3] += x.w;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = sacc[r];
            if (a.w_scale) x *= a.w_scale[n + r];
            if (a.bias && n + r < a.N) x += a.bias[n + r];
            v[r] = epi_act<EPI>(x);
        }
    }
    const int64_t crow = remap_row(a, m);
    OutT* cp = (OutT*)a.C + crow * a.ldc + n;
    const bf16_t* rp = a.residual ? (const bf16_t*)a.residual + (int64_t)m * a.ldr + n : nullptr;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (n + r < n_out) {
            float x = v[r];
            if (rp) x += bf2f(rp[r]);
            Act<OutT>::st(cp + r, x);
        }
}

// split-K reduction of one output row per workgroup + residual + store, then the NEXT layer norm of that row in the same
// launch (post_norm_*): the row is staged in LDS exactly as rmsnorm_kernel stages it, and the statistics / rounding follow
// that kernel instruction for instruction, so fused and unfused paths are bit-identical.
#define PN_NORM_MAX 8192
#define PN_MAX_SPLITK 8
template <typename OutT>
__global__ __launch_bounds__(1024) void gemm_panel_reduce_norm_kernel(vcla_gemm_args a, int splitk, int n_pad, const float* __restrict__ partial) {
    __shared__ float row[PN_NORM_MAX];
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const bf16_t* rp = a.residual ? (const bf16_t*)a.residual + (int64_t)m * a.ldr : nullptr;
    OutT* cp = (OutT*)a.C + (int64_t)m * a.ldc;
    for (int n = tid * 4; n < a.N; n += 1024 * 4) {
        float4 p[PN_MAX_SPLITK];   // every slice's load in flight before the first add
#pragma unroll
        for (int s = 0; s < PN_MAX_SPLITK; ++s)
            p[s] = s < splitk ? *reinterpret_cast<const float4*>(partial + ((int64_t)s * a.M + m) * n_pad + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        float x[4] = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < PN_MAX_SPLITK; ++s) { x[0] += p[s].x; x[1] += p[s].y; x[2] += p[s].z; x[3] += p[s].w; }   // slice order, as the plain reduce
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (n + r >= a.N) break;
            float v = x[r];
            if (a.w_scale) v *= a.w_scale[n + r];
            if (a.bias) v += a.bias[n + r];
            if (rp) v += bf2f(rp[n + r]);
            v = Act<OutT>::rnd(v);
            Act<OutT>::st(cp + n + r, v);
            row[n + r] = v;
        }
    }
    __syncthreads();
    // statistics exactly as rmsnorm_kernel: 256 strided partial sums (threads 0..255), wave sums, 4-term total
    float q = 0.f;
    if (tid < 256)
        for (int c = tid; c < a.N; c += 256) q += row[c] * row[c];
    q = wave_sum(q);
    if ((tid & 63) == 0 && tid < 256) red[tid >> 6] = q;
    __syncthreads();
    const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)a.N + a.post_norm_eps);
    OutT* yr = (OutT*)a.post_norm_out + (int64_t)m * a.post_norm_ld;
    for (int c = tid; c < a.N; c += 1024) Act<OutT>::st(yr + c, a.post_norm_gamma[c] * Act<OutT>::rnd(row[c] * rstd));
}

static int panel_splitk(const vcla_gemm_args* a, int n_pad) {
    const int tiles_n = (a->N + PN_BN - 1) / PN_BN, nk = a->K / GM_BK;
    int s = (320 + tiles_n / 2) / tiles_n;
    // M > 64 (8 MFMA row tiles): the kernel is held to 246 registers (launch bounds) = two workgroups per CU; the grid
    // stays within that one round of 512 (measured: 198 -> 158 us per layer for the four M = 128 GEMMs)
    if (a->M > 64) s = 512 / tiles_n;
    if (s < 1) s = 1;
    if (s > 8) s = 8;
    if (s > nk) s = nk;
    while (s > 1 && (size_t)s * a->M * n_pad * 4 > a->splitk_ws_bytes) --s;  // no / small workspace -> fewer slices
    if (!a->splitk_ws) s = 1;
    return s;
}

template <int EPI, typename OutT, int MT>
static int launch_panel_mt(const vcla_gemm_args* a, hipStream_t s) {
    const int n_pad = (a->N + 127) / 128 * 128;
    int splitk = panel_splitk(a, n_pad);
    const int tiles_n = (a->N + PN_BN - 1) / PN_BN;
    // 8-wave form (two K groups per workgroup): fragment-major weights (bf16 or fp8), M <= 64, at most one workgroup per CU
    static const int kg_env = getenv("VCLA_PANEL_KG") ? atoi(getenv("VCLA_PANEL_KG")) : 2;
    bool kg2 = false;
    if constexpr (MT <= 4) {
        if (kg_env == 2 && (a->W_frag || a->W_q8_frag) && tiles_n <= 256) {
            int s2 = 256 / tiles_n;                    // <= 256 workgroups: every CU gets at most one, no second round
            const int nk = a->K / GM_BK;
            static const int s2max = getenv("VCLA_PANEL_S2MAX") ? atoi(getenv("VCLA_PANEL_S2MAX")) : 8;
            if (s2 > s2max) s2 = s2max;
            if (s2 > nk / 2) s2 = nk / 2 > 0 ? nk / 2 : 1;   // each K group wants at least one tile
            while (s2 > 1 && (size_t)s2 * a->M * n_pad * 4 > a->splitk_ws_bytes) --s2;
            if (!a->splitk_ws) s2 = 1;
            if (tiles_n * s2 >= 128) { kg2 = true; splitk = s2; }   // small problems keep the 4-wave form (more workgroups)
        }
    }
    dim3 grid(tiles_n, splitk);
    if (kg2) {
        if constexpr (MT <= 4) {
            if (a->W_q8_frag) gemm_panel_kernel<EPI, OutT, MT, 2, 2><<<grid, 512, 0, s>>>(*a, splitk, n_pad, (float*)a->splitk_ws);
            else gemm_panel_kernel<EPI, OutT, MT, 1, 2><<<grid, 512, 0, s>>>(*a, splitk, n_pad, (float*)a->splitk_ws);
        }
    } else
    if (a->W_q8_frag) gemm_panel_kernel<EPI, OutT, MT, 2><<<grid, 256, 0, s>>>(*a, splitk, n_pad, (float*)a->splitk_ws);
    else if (a->W_frag) gemm_panel_kernel<EPI, OutT, MT, 1><<<grid, 256, 0, s>>>(*a, splitk, n_pad, (float*)a->splitk_ws);
    else gemm_panel_kernel<EPI, OutT, MT, 0><<<grid, 256, 0, s>>>(*a, splitk, n_pad, (float*)a->splitk_ws);
    VCLA_CHECK_LAUNCH("gemm_panel_kernel");
    if constexpr (EPI == VCLA_EPI_NONE) {
        if (splitk > 1 && splitk <= PN_MAX_SPLITK && a->post_norm_gamma) {   // vcla_gemm checked the preconditions and skips its own rmsnorm launch
            gemm_panel_reduce_norm_kernel<OutT><<<a->M, 1024, 0, s>>>(*a, splitk, n_pad, (const float*)a->splitk_ws);
            VCLA_CHECK_LAUNCH("gemm_panel_reduce_norm_kernel");
            return VCLA_POST_NORM_DONE;
        }
    }
    if (splitk > 1) {
        const int n_out = (EPI == VCLA_EPI_SWIGLU) ? a->N / 2 : a->N;
        const int64_t work = (int64_t)a->M * ((n_out + 3) / 4);
        gemm_panel_reduce_kernel<EPI, OutT><<<(unsigned)((work + 255) / 256), 256, 0, s>>>(*a, splitk, n_pad, (const float*)a->splitk_ws);
        VCLA_CHECK_LAUNCH("gemm_panel_reduce_kernel");
    }
    return VCLA_OK;
}

template <int EPI, typename OutT>
static int launch_panel(const vcla_gemm_args* a, hipStream_t s) {
    if (a->M <= 16) return launch_panel_mt<EPI, OutT, 1>(a, s);
    if (a->M <= 32) return launch_panel_mt<EPI, OutT, 2>(a, s);
    if (a->M <= 64) return launch_panel_mt<EPI, OutT, 4>(a, s);
    if (a->M <= 128) return launch_panel_mt<EPI, OutT, 8>(a, s);
    return vcla_fail(VCLA_ERR_BAD_SHAPE, "gemm: panel kernel needs M <= 128 (got %d)", a->M);
}

// =================================================================== GEMV kernel (M <= 8)
// x [MB, K] act dtype; each wave owns 4 weight rows (SWIGLU: 2 gate rows + their 2 up rows), lanes stride K by 8.
template <typename T> __device__ __forceinline__ void load8(const T* p, float* v);
template <> __device__ __forceinline__ void load8<float>(const float* p, float* v) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float* v) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    bf8_to_f32(t, v);
}


// R = weight rows per wave (SWIGLU: R/2 gate rows + their R/2 up rows).  Optional fused RMSNorm prologue
// (norm_gamma != NULL): y = W . (gamma * x * rstd(x)) computed as rstd * sum_k w_k (gamma_k x_k), with sum x^2
// accumulated in the same pass -- the decode path needs no separate norm launch and no normalised copy of x.
template <typename T, typename OutT, int MB, int EPI, int R>
__global__ __launch_bounds__(256) void gemv_kernel(vcla_gemm_args a) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);  // global wave index
    int rows[R];
    int nvalid;  // outputs this wave really owns
    if constexpr (EPI == VCLA_EPI_SWIGLU) {
        constexpr int P = R / 2;
        const int j0 = gw * P;  // output columns j0 .. j0+P-1
        nvalid = (a.N / 2 - j0) < P ? (a.N / 2 - j0) : P;
#pragma unroll
        for (int r = 0; r < P; ++r) {
            const int j = j0 + r;
            rows[r] = (j >> 4) * 32 + (j & 15);  // gate row in the 16-interleaved packing
            rows[r + P] = rows[r] + 16;          // matching up row
        }
    } else {
        nvalid = (a.N - gw * R) < R ? (a.N - gw * R) : R;
#pragma unroll
        for (int r = 0; r < R; ++r) rows[r] = gw * R + r;
    }
    if (nvalid <= 0) return;  // wave-uniform
    const bf16_t* Wg = (const bf16_t*)a.W;
    const T* X = (const T*)a.A;
    const bf16_t* wp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wp[r] = Wg + (int64_t)rows[r] * a.K;  // rows beyond N stay inside the 128-row padding
                                                                      // (SWIGLU waves are never ragged: N/2 % 16 == 0)
    const T* xp[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) xp[m] = X + (int64_t)(m < a.M ? m : a.M - 1) * a.lda;

    float acc[R][MB];
    float ssq[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        ssq[m] = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r][m] = 0.f;
    }
    const bool fused_norm = a.norm_gamma != nullptr;

#pragma unroll 2
    for (int k = lane * 8; k < a.K; k += 512) {
        uint4 w[R];
#pragma unroll
        for (int r = 0; r < R; ++r) w[r] = ldg_nt(wp[r] + k);
        float xv[MB][8];
#pragma unroll
        for (int m = 0; m < MB; ++m) load8<T>(xp[m] + k, xv[m]);
        if (fused_norm) {
            const float4 g0 = *reinterpret_cast<const float4*>(a.norm_gamma + k);
            const float4 g1 = *reinterpret_cast<const float4*>(a.norm_gamma + k + 4);
            const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ssq[m] += xv[m][e] * xv[m][e];
                    xv[m][e] *= gm[e];
                }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float wf[8];
            bf8_to_f32(w[r], wf);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[r][m] += wf[e] * xv[m][e];
        }
    }
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        float sc = 1.f;
        if (fused_norm) sc = rsqrtf(wave_sum(ssq[m]) / (float)a.K + a.norm_eps);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r][m] = wave_sum(acc[r][m]) * sc;
    }

    // lane (m * R + r) writes output (m, r)
    OutT* Cg = (OutT*)a.C;
    if constexpr (EPI == VCLA_EPI_SWIGLU) {
        constexpr int P = R / 2;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < P; ++r) {
                if (lane == m * R + r && r < nvalid && m < a.M) {
                    float gt = acc[r][m], up = acc[r + P][m];
                    if (a.bias) { gt += a.bias[rows[r]]; up += a.bias[rows[r + P]]; }
                    float v = act_silu(gt) * up;
                    const int n = gw * P + r;
                    if (a.residual) v += Act<T>::ld((const T*)a.residual + (int64_t)m * a.ldr + n);
                    Act<OutT>::st(Cg + remap_row(a, m) * a.ldc + n, v);
                }
            }
    } else {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (lane == m * R + r && r < nvalid && m < a.M) {
                    float v = acc[r][m];
                    const int n = rows[r];
                    if (a.bias) v += a.bias[n];
                    v = epi_act<EPI>(v);
                    if (a.residual) v += Act<T>::ld((const T*)a.residual + (int64_t)m * a.ldr + n);
                    Act<OutT>::st(Cg + remap_row(a, m) * a.ldc + n, v);
                }
            }
    }
}

// =================================================================== M = 1 bf16 GEMV (the decode hot kernel)
// Configuration picked with tools/bench_kernels.py `tune` on MI355X (profiles/r01_kernel_microbench_run3.txt):
// x (pre-multiplied by the RMSNorm gain) is staged ONCE per workgroup in LDS as fp32, so the vector-memory queue carries
// nothing but weight rows; 8 waves per workgroup, 1 row per wave (2 for SwiGLU pairs / very tall matrices), 4 k-steps
// (4 x 16 B per lane per row) in flight, non-temporal weight loads.  +10..40 % over the generic gemv_kernel.
template <int R, int U, int WPB, bool SWIGLU, typename OutT>
__global__ __launch_bounds__(WPB * 64) void gemv1_kernel(vcla_gemm_args a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [K]
    __shared__ float red[WPB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * WPB + wave;
    const bf16_t* X = (const bf16_t*)a.A;
    const bool fused_norm = a.norm_gamma != nullptr;
    float rstd = 1.f;
    {
        float ss = 0.f;
        for (int k = threadIdx.x * 8; k < a.K; k += WPB * 64 * 8) {
            float xv[8];
            bf8_to_f32(*reinterpret_cast<const uint4*>(X + k), xv);
            if (fused_norm) {
                const float4 g0 = *reinterpret_cast<const float4*>(a.norm_gamma + k);
                const float4 g1 = *reinterpret_cast<const float4*>(a.norm_gamma + k + 4);
                const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) { ss += xv[e] * xv[e]; xv[e] *= gm[e]; }
            }
            *reinterpret_cast<float4*>(xs + k) = make_float4(xv[0], xv[1], xv[2], xv[3]);
            *reinterpret_cast<float4*>(xs + k + 4) = make_float4(xv[4], xv[5], xv[6], xv[7]);
        }
        if (fused_norm) {
            ss = wave_sum(ss);
            if (lane == 0) red[wave] = ss;
        }
        __syncthreads();
        if (fused_norm) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < WPB; ++w) tot += red[w];
            rstd = rsqrtf(tot / (float)a.K + a.norm_eps);
        }
    }
    constexpr int P = SWIGLU ? R / 2 : R;  // outputs per wave
    const int n_out = SWIGLU ? a.N / 2 : a.N;
    if (gw * P >= n_out) return;
    int rows[R];
    if (SWIGLU) {
#pragma unroll
        for (int r = 0; r < P; ++r) {
            const int j = gw * P + r;
            rows[r] = (j >> 4) * 32 + (j & 15);
            rows[r + P] = rows[r] + 16;
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) rows[r] = gw * R + r;  // rows past N stay inside the 128-row padding of W
    }
    const bf16_t* Wg = (const bf16_t*)a.W;
    const bf16_t* wp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wp[r] = Wg + (int64_t)rows[r] * a.K;
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;

    for (int k0 = lane * 8; k0 < a.K; k0 += 512 * U) {
        uint4 w[U][R];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * 512;
            if (k < a.K) {
#pragma unroll
                for (int r = 0; r < R; ++r) w[u][r] = ldg_nt(wp[r] + k);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * 512;
            if (k < a.K) {
                const float4 x0 = *reinterpret_cast<const float4*>(xs + k), x1 = *reinterpret_cast<const float4*>(xs + k + 4);
                const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float wf[8];
                    bf8_to_f32(w[u][r], wf);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[r] += wf[e] * xv[e];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]) * rstd;
    OutT* Cg = (OutT*)a.C + remap_row(a, 0) * a.ldc;
    if (SWIGLU) {
#pragma unroll
        for (int r = 0; r < P; ++r)
            if (lane == r) {
                float gt = acc[r], up = acc[r + P];
                if (a.bias) { gt += a.bias[rows[r]]; up += a.bias[rows[r + P]]; }
                float v = act_silu(gt) * up;
                const int n = gw * P + r;
                if (a.residual) v += bf2f(((const bf16_t*)a.residual)[n]);
                Act<OutT>::st(Cg + n, v);
            }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (lane == r && rows[r] < a.N) {
                float v = acc[r];
                if (a.bias) v += a.bias[rows[r]];
                if (a.residual) v += bf2f(((const bf16_t*)a.residual)[rows[r]]);
                Act<OutT>::st(Cg + rows[r], v);
            }
    }
}

// fp8 (e4m3fn) weights, per-row scale: the same streaming structure with HALF the bytes per weight element.
// lane loads 16 B = 16 weights; 8 v_cvt_pk_f32_fp8 per load; the row scale is applied once after the reduction.
template <int R, int U, int WPB, bool SWIGLU, typename OutT>
__global__ __launch_bounds__(WPB * 64) void gemv1_fp8_kernel(vcla_gemm_args a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [K]
    __shared__ float red[WPB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * WPB + wave;
    const bf16_t* X = (const bf16_t*)a.A;
    const bool fused_norm = a.norm_gamma != nullptr;
    float rstd = 1.f;
    {
        float ss = 0.f;
        for (int k = threadIdx.x * 8; k < a.K; k += WPB * 64 * 8) {
            float xv[8];
            bf8_to_f32(*reinterpret_cast<const uint4*>(X + k), xv);
            if (fused_norm) {
                const float4 g0 = *reinterpret_cast<const float4*>(a.norm_gamma + k);
                const float4 g1 = *reinterpret_cast<const float4*>(a.norm_gamma + k + 4);
                const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) { ss += xv[e] * xv[e]; xv[e] *= gm[e]; }
            }
            *reinterpret_cast<float4*>(xs + k) = make_float4(xv[0], xv[1], xv[2], xv[3]);
            *reinterpret_cast<float4*>(xs + k + 4) = make_float4(xv[4], xv[5], xv[6], xv[7]);
        }
        if (fused_norm) {
            ss = wave_sum(ss);
            if (lane == 0) red[wave] = ss;
        }
        __syncthreads();
        if (fused_norm) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < WPB; ++w) tot += red[w];
            rstd = rsqrtf(tot / (float)a.K + a.norm_eps);
        }
    }
    constexpr int P = SWIGLU ? R / 2 : R;
    const int n_out = SWIGLU ? a.N / 2 : a.N;
    if (gw * P >= n_out) return;
    int rows[R];
    if (SWIGLU) {
#pragma unroll
        for (int r = 0; r < P; ++r) {
            const int j = gw * P + r;
            rows[r] = (j >> 4) * 32 + (j & 15);
            rows[r + P] = rows[r] + 16;
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) rows[r] = gw * R + r;
    }
    const unsigned char* Wq = (const unsigned char*)a.W_q8;
    const unsigned char* wp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wp[r] = Wq + (int64_t)rows[r] * a.K;
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    for (int k0 = lane * 16; k0 < a.K; k0 += 1024 * U) {
        u32x4_t w[U][R];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * 1024;
            if (k < a.K) {
#pragma unroll
                for (int r = 0; r < R; ++r) w[u][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wp[r] + k));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * 1024;
            if (k < a.K) {
                float xv[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 t = *reinterpret_cast<const float4*>(xs + k + q * 4);
                    xv[q * 4] = t.x; xv[q * 4 + 1] = t.y; xv[q * 4 + 2] = t.z; xv[q * 4 + 3] = t.w;
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t d[4] = {w[u][r].x, w[u][r].y, w[u][r].z, w[u][r].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(d[q], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(d[q], true);
                        acc[r] += lo.x * xv[q * 4] + lo.y * xv[q * 4 + 1] + hi.x * xv[q * 4 + 2] + hi.y * xv[q * 4 + 3];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]) * rstd * a.w_scale[rows[r]];
    OutT* Cg = (OutT*)a.C + remap_row(a, 0) * a.ldc;
    if (SWIGLU) {
#pragma unroll
        for (int r = 0; r < P; ++r)
            if (lane == r) {
                float gt = acc[r], up = acc[r + P];
                if (a.bias) { gt += a.bias[rows[r]]; up += a.bias[rows[r + P]]; }
                float v = act_silu(gt) * up;
                const int n = gw * P + r;
                if (a.residual) v += bf2f(((const bf16_t*)a.residual)[n]);
                Act<OutT>::st(Cg + n, v);
            }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (lane == r && rows[r] < a.N) {
                float v = acc[r];
                if (a.bias) v += a.bias[rows[r]];
                if (a.residual) v += bf2f(((const bf16_t*)a.residual)[rows[r]]);
                Act<OutT>::st(Cg + rows[r], v);
            }
    }
}

template <int R, bool SWIGLU, typename OutT>
static int launch_gemv1_fp8(const vcla_gemm_args* a, hipStream_t s) {
    constexpr int WPB = 8, U = 4;
    constexpr int P = SWIGLU ? R / 2 : R;
    const int n_out = SWIGLU ? a->N / 2 : a->N;
    const int waves = (n_out + P - 1) / P;
    gemv1_fp8_kernel<R, U, WPB, SWIGLU, OutT><<<(waves + WPB - 1) / WPB, WPB * 64, (size_t)a->K * 4, s>>>(*a);
    VCLA_CHECK_LAUNCH("gemv1_fp8_kernel");
    return VCLA_OK;
}

template <int R, bool SWIGLU, typename OutT>
static int launch_gemv1(const vcla_gemm_args* a, hipStream_t s) {
    constexpr int WPB = 8, U = 4;
    constexpr int P = SWIGLU ? R / 2 : R;
    const int n_out = SWIGLU ? a->N / 2 : a->N;
    const int waves = (n_out + P - 1) / P;
    const int blocks = (waves + WPB - 1) / WPB;
    gemv1_kernel<R, U, WPB, SWIGLU, OutT><<<blocks, WPB * 64, (size_t)a->K * 4, s>>>(*a);
    VCLA_CHECK_LAUNCH("gemv1_kernel");
    return VCLA_OK;
}

// bf16, M == 1, no activation other than SwiGLU, K*4 bytes of LDS available
static bool gemv1_applicable(const vcla_gemm_args* a, int dtype) {
    return dtype == VCLA_BF16 && a->M == 1 && a->K <= 15360 && (a->epilogue == VCLA_EPI_NONE || a->epilogue == VCLA_EPI_SWIGLU);
}
int vcla_gemv1x_launch(const vcla_gemm_args* a, hipStream_t s);   // gemv_decode.hip
static int launch_gemv1_auto(const vcla_gemm_args* a, hipStream_t s) {
    {   // compile-time-K persistent form for the LLaMA widths (bf16 and fp8 weights); -1 = no instance for this K
        const int rc = vcla_gemv1x_launch(a, s);
        if (rc != -1) return rc;
    }
    if (a->W_q8 && a->w_scale) {   // fp8 weight copy present: half the HBM bytes
        if (a->epilogue == VCLA_EPI_SWIGLU)
            return a->out_f32 ? launch_gemv1_fp8<2, true, float>(a, s) : launch_gemv1_fp8<2, true, bf16_t>(a, s);
        return a->out_f32 ? launch_gemv1_fp8<2, false, float>(a, s) : launch_gemv1_fp8<2, false, bf16_t>(a, s);
    }
    if (a->epilogue == VCLA_EPI_SWIGLU)
        return a->out_f32 ? launch_gemv1<2, true, float>(a, s) : launch_gemv1<2, true, bf16_t>(a, s);
    if (a->N >= 16384)  // very tall (lm_head): 2 rows per wave halves the per-workgroup x staging
        return a->out_f32 ? launch_gemv1<2, false, float>(a, s) : launch_gemv1<2, false, bf16_t>(a, s);
    return a->out_f32 ? launch_gemv1<1, false, float>(a, s) : launch_gemv1<1, false, bf16_t>(a, s);
}

// =================================================================== fp32-activation tile kernel (parity mode)
// 64x64x16 tile, 256 threads, thread (ty, tx) computes rows ty+16i, columns tx+16j (so SWIGLU pairs are thread-local)
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(vcla_gemm_args a) {
    __shared__ float As[16][64 + 4];
    __shared__ float Ws[16][64 + 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const float* Ag = (const float*)a.A;
    const bf16_t* Wg = (const bf16_t*)a.W;
    const int lrow = tid >> 2, lk = (tid & 3) * 4;
    int am = m0 + lrow;
    if (am >= a.M) am = a.M - 1;
    const float* ap = Ag + (int64_t)am * a.lda + lk;
    const bf16_t* wp = Wg + (int64_t)(n0 + lrow) * a.K + lk;  // padded rows exist up to a multiple of 128
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < a.K; k0 += 16) {
        const float4 av = *reinterpret_cast<const float4*>(ap + k0);
        float wv[4];
        Act<bf16_t>::ld4(wp + k0, wv);
        As[lk][lrow] = av.x; As[lk + 1][lrow] = av.y; As[lk + 2][lrow] = av.z; As[lk + 3][lrow] = av.w;
        Ws[lk][lrow] = wv[0]; Ws[lk + 1][lrow] = wv[1]; Ws[lk + 2][lrow] = wv[2]; Ws[lk + 3][lrow] = wv[3];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float av4[4], wv4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { av4[i] = As[k][ty + 16 * i]; wv4[i] = Ws[k][tx + 16 * i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av4[i], wv4[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* Cg = (float*)a.C;
    const float* Rg = (const float*)a.residual;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty + 16 * i;
        if (m >= a.M) continue;
        const int64_t crow = remap_row(a, m);
        if constexpr (EPI == VCLA_EPI_SWIGLU) {
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const int pg = n0 + tx + 32 * jp;  // packed gate column, up = +16
                const int n = (n0 / 2) + tx + 16 * jp;
                if (n >= a.N / 2) continue;
                float gt = acc[i][2 * jp], up = acc[i][2 * jp + 1];
                if (a.bias) { gt += a.bias[pg]; up += a.bias[pg + 16]; }
                float v = act_silu(gt) * up;
                if (Rg) v += Rg[(int64_t)m * a.ldr + n];
                Cg[crow * a.ldc + n] = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + tx + 16 * j;
                if (n >= a.N) continue;
                float v = acc[i][j];
                if (a.bias) v += a.bias[n];
                v = epi_act<EPI>(v);
                if (Rg) v += Rg[(int64_t)m * a.ldr + n];
                Cg[crow * a.ldc + n] = v;
            }
        }
    }
}

// =================================================================== host dispatch
template <int EPI, typename OutT>
static int launch_mfma(const vcla_gemm_args* a, hipStream_t s) {
    const int tiles_m = (a->M + GM_BM - 1) / GM_BM, tiles_n = (a->N + GM_BN - 1) / GM_BN;
    const int tiles = tiles_m * tiles_n, nk = a->K / GM_BK;
    // Few tiles, long K (one image through the ViT: 24 - 96 tiles on 256 CUs, 16 - 64 serial K steps): split K so that ~256
    // workgroups run >= 4 K steps each; fp32 partial tiles + the panel kernel's reduce launch.  VCLA_MFMA128_SPLITK=0: off.
    static const int sk_env = getenv("VCLA_MFMA128_SPLITK") ? atoi(getenv("VCLA_MFMA128_SPLITK")) : 1;
    // Round 4: also for 129 - 255 tiles (a 129 - 256-row decode batch: qkv = 192 tiles -> 2 slices) and with a post-norm request (o_proj /
    // down_proj of those batches: 64 tiles ran 64 serial K steps on 64 CUs = 153 us; the wrapper's rmsnorm launch follows the reduce).
    static const int s_force = getenv("VCLA_MFMA128_S") ? atoi(getenv("VCLA_MFMA128_S")) : 0;     // experiments: force the slice count
    int S = 1;
    if (sk_env && a->splitk_ws && tiles < 256 && nk >= 8) {
        const int n_pad = (a->N + 127) / 128 * 128;
        S = (256 + tiles - 1) / tiles;
        if (tiles <= 64 && a->M > 128) S *= 2;       // one 128-row pair of tiles per 128 columns: fill both workgroup slots of every CU
        if (s_force > 0) S = s_force;
        if (S > nk / 4) S = nk / 4;
        if (S > 8) S = 8;
        while (S > 1 && (size_t)S * a->M * n_pad * 4 > a->splitk_ws_bytes) --S;
    }
    if (S > 1) {
        const int n_pad = (a->N + 127) / 128 * 128;
        gemm_mfma_kernel<EPI, OutT, true><<<dim3(tiles, S), 256, 0, s>>>(*a, tiles_m, tiles_n, S, n_pad, (float*)a->splitk_ws);
        VCLA_CHECK_LAUNCH("gemm_mfma_kernel");
        if constexpr (EPI == VCLA_EPI_NONE && sizeof(OutT) == 2) {
            if (S <= PN_MAX_SPLITK && a->post_norm_gamma && a->N <= PN_NORM_MAX) {   // reduce + residual + the NEXT RMSNorm of the row in one launch (as the panel kernel)
                gemm_panel_reduce_norm_kernel<OutT><<<a->M, 1024, 0, s>>>(*a, S, n_pad, (const float*)a->splitk_ws);
                VCLA_CHECK_LAUNCH("gemm_panel_reduce_norm_kernel");
                return VCLA_POST_NORM_DONE;
            }
        }
        const int n_out = (EPI == VCLA_EPI_SWIGLU) ? a->N / 2 : a->N;
        const int64_t work = (int64_t)a->M * ((n_out + 3) / 4);
        gemm_panel_reduce_kernel<EPI, OutT><<<(unsigned)((work + 255) / 256), 256, 0, s>>>(*a, S, n_pad, (const float*)a->splitk_ws);
        VCLA_CHECK_LAUNCH("gemm_panel_reduce_kernel");
        return VCLA_OK;
    }
    gemm_mfma_kernel<EPI, OutT><<<tiles, 256, 0, s>>>(*a, tiles_m, tiles_n);
    VCLA_CHECK_LAUNCH("gemm_mfma_kernel");
    return VCLA_OK;
}

// Which MFMA tile?  256x256 (1 workgroup / CU) runs ~1.35x the 128x128 kernel (2 / CU) per flop but quantises worse:
// cost = rounds(ceil) / rounds(exact) / relative speed.
static bool prefer_256(const vcla_gemm_args* a) {
    if (a->M < 256 || a->N < 256) return false;
    const double b256 = (double)((a->M + 255) / 256) * ((a->N + 255) / 256) / 256.0;
    const double b128 = (double)((a->M + 127) / 128) * ((a->N + 127) / 128) / 512.0;
    const double c256 = ceil(b256) / b256 / 1.35, c128 = ceil(b128) / b128;
    return c256 < c128;
}

template <typename T, typename OutT, int EPI, int R>
static int launch_gemv_r(const vcla_gemm_args* a, hipStream_t s) {
    constexpr int per_wave = (EPI == VCLA_EPI_SWIGLU) ? R / 2 : R;  // outputs per wave
    const int n_out = (EPI == VCLA_EPI_SWIGLU) ? a->N / 2 : a->N;
    const int waves = (n_out + per_wave - 1) / per_wave;
    const int blocks = (waves + 3) / 4;
    if (a->M == 1) gemv_kernel<T, OutT, 1, EPI, R><<<blocks, 256, 0, s>>>(*a);
    else if (a->M == 2) gemv_kernel<T, OutT, 2, EPI, R><<<blocks, 256, 0, s>>>(*a);
    else if (a->M <= 4) gemv_kernel<T, OutT, 4, EPI, R><<<blocks, 256, 0, s>>>(*a);
    else if (a->M <= 8) gemv_kernel<T, OutT, 8, EPI, R><<<blocks, 256, 0, s>>>(*a);
    else return vcla_fail(VCLA_ERR_BAD_SHAPE, "gemv: M=%d > 8", a->M);
    VCLA_CHECK_LAUNCH("gemv_kernel");
    return VCLA_OK;
}

template <typename T, typename OutT, int EPI>
static int launch_gemv(const vcla_gemm_args* a, hipStream_t s) {
    // enough waves to keep >= 8 per CU in flight: 4 rows per wave only when the matrix is tall
    const int n_out = (EPI == VCLA_EPI_SWIGLU) ? a->N / 2 : a->N;
    if (EPI == VCLA_EPI_SWIGLU) {
        if (n_out >= 4096) return launch_gemv_r<T, OutT, EPI, 4>(a, s);
        return launch_gemv_r<T, OutT, EPI, 2>(a, s);
    }
    if (n_out >= 8192) return launch_gemv_r<T, OutT, EPI, 4>(a, s);
    return launch_gemv_r<T, OutT, EPI, 2>(a, s);
}

template <int EPI>
static int dispatch_epi(const vcla_gemm_args* a, int dtype, int kernel, hipStream_t s) {
    if (kernel == 1) {
        return a->out_f32 ? launch_mfma<EPI, float>(a, s) : launch_mfma<EPI, bf16_t>(a, s);
    } else if (kernel == 4) {
        return vcla_gemm_mfma256_launch(a, true, s);
    } else if (kernel == 5) {  // same kernel, compiler-chosen ds_read / MFMA interleave, plain form (A/B reference for kernel 4)
        return vcla_gemm_mfma256_launch(a, false, s);
    } else if (kernel == 7) {
        return a->out_f32 ? launch_skinny<EPI, float>(a, s) : launch_skinny<EPI, bf16_t>(a, s);
    } else if (kernel == 8) {
        return a->out_f32 ? launch_panel<EPI, float>(a, s) : launch_panel<EPI, bf16_t>(a, s);
    } else if (kernel == 9) {
        return vcla_gemm_dstream_launch(a, s);
    } else if (kernel == 10) {
        return vcla_gemm_mfma256_fp8_launch(a, s);
    } else if (kernel >= 11 && kernel <= 14) {
        return vcla_gemm_ring_launch(a, s);
    } else if (kernel == 2 || kernel == 6) {
        if (kernel == 2 && gemv1_applicable(a, dtype)) return launch_gemv1_auto(a, s);
        if (dtype == VCLA_F32) return launch_gemv<float, float, EPI>(a, s);
        return a->out_f32 ? launch_gemv<bf16_t, float, EPI>(a, s) : launch_gemv<bf16_t, bf16_t, EPI>(a, s);
    } else {
        dim3 grid((a->N + 63) / 64, (a->M + 63) / 64);
        gemm_f32_kernel<EPI><<<grid, 256, 0, s>>>(*a);
        VCLA_CHECK_LAUNCH("gemm_f32_kernel");
        return VCLA_OK;
    }
}

static int gemm_impl(const vcla_gemm_args* a, int dtype, void* stream);

extern "C" int vcla_gemm(const vcla_gemm_args* a, int dtype, void* stream) {
    VCLA_REQUIRE(a, VCLA_ERR_BAD_ARG, "gemm: null args");
    if (a->post_norm_gamma) {
        VCLA_REQUIRE(a->post_norm_out && a->epilogue == VCLA_EPI_NONE && !a->out_f32 && a->c_group_rows <= 0 && a->N <= PN_NORM_MAX &&
                         a->post_norm_ld >= a->N, VCLA_ERR_BAD_ARG,
                     "gemm: post_norm needs post_norm_out, epilogue NONE, activation-dtype C, no row regrouping, N <= %d", PN_NORM_MAX);
    }
    const int rc = gemm_impl(a, dtype, stream);
    if (rc == VCLA_POST_NORM_DONE) return VCLA_OK;
    if (rc || !a->post_norm_gamma || a->M == 0) return rc;
    return vcla_rmsnorm(a->C, a->ldc, a->post_norm_gamma, a->post_norm_out, a->post_norm_ld, a->M, a->N, a->post_norm_eps, dtype, stream);
}

static int gemm_impl(const vcla_gemm_args* a, int dtype, void* stream) {
    VCLA_REQUIRE(dtype == VCLA_F32 || dtype == VCLA_BF16, VCLA_ERR_BAD_DTYPE, "gemm: bad dtype %d", dtype);
    VCLA_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0 && a->K % GM_BK == 0, VCLA_ERR_BAD_SHAPE,
                 "gemm: M=%d N=%d K=%d (K must be a positive multiple of %d)", a->M, a->N, a->K, GM_BK);
    VCLA_REQUIRE((a->A || a->A_frag || a->A_q8 || a->A_slab) && a->W && (a->C || a->C_frag), VCLA_ERR_BAD_ARG, "gemm: null pointer");
    VCLA_REQUIRE(!(a->A_slab || a->W_slab || a->W_q8_slab) || (a->force_kernel >= 11 && a->force_kernel <= 14) || a->force_kernel == 4, VCLA_ERR_BAD_ARG,
                 "gemm: slab-major operands belong to the ring kernel (force_kernel 11 - 14) and the 256 x 256 kernel (4)");
    VCLA_REQUIRE(!a->A_slab || (vcla_aligned(a->A_slab, 16) && a->a_slab_rows >= a->M), VCLA_ERR_BAD_ARG, "gemm: A_slab must be 16-byte aligned with a_slab_rows >= M");
    VCLA_REQUIRE(a->epilogue >= VCLA_EPI_NONE && a->epilogue <= VCLA_EPI_SWIGLU, VCLA_ERR_BAD_ARG, "gemm: bad epilogue %d",
                 a->epilogue);
    VCLA_REQUIRE(a->epilogue != VCLA_EPI_SWIGLU || a->N % 32 == 0, VCLA_ERR_BAD_SHAPE,
                 "gemm: SWIGLU needs N %% 32 == 0 (got %d)", a->N);
    const int64_t aa = dtype == VCLA_F32 ? 4 : 8;
    VCLA_REQUIRE(!a->A || (a->lda % aa == 0 && a->lda >= a->K && vcla_aligned(a->A, 16)), VCLA_ERR_BAD_SHAPE,
                 "gemm: A must be 16-byte aligned with lda %% %lld == 0 (lda=%lld)", (long long)aa, (long long)a->lda);
    VCLA_REQUIRE(vcla_aligned(a->W, 16), VCLA_ERR_BAD_SHAPE, "gemm: W must be 16-byte aligned");
    if (a->M == 0) return VCLA_OK;
    int kernel = a->force_kernel;
    if (kernel == 0 && a->A_frag) kernel = 9;   // fragment-major activations exist only for the streaming decode GEMM
    if (kernel == 0 && a->A_q8) kernel = 10;    // fp8 activations exist only for the fp8 MFMA kernel
    if (kernel == 0) {
        if (dtype == VCLA_F32) kernel = a->M <= 8 ? 2 : 3;
        else if (a->M == 1 || (a->norm_gamma && a->M <= 8)) kernel = 2;   // GEMV (fused-norm capable)
        else if (a->M <= 128) {
            // W streamed once.  Measured on MI355X (profiles/r01_kernel_microbench_run5.txt): the split-K panel kernel wins
            // for M >= 64, for short-N / long-K shapes (down-proj) and for N <= 4096 once M >= 32; else the skinny kernel
            const bool panel = a->W_frag || a->W_q8_frag || (a->splitk_ws && (a->M >= 64 || (a->N <= 4096 && (a->K >= 8192 || a->M >= 32))));
            kernel = panel ? 8 : 7;
            // the resampler's 64 latent rows (K = 1024: q / kv / out / fc1): one launch of 64 x 64 ring tiles instead of K slices + a reduce launch
            // (graph-replayed, rotating weights, M = 64: 15.8 - 18.0 -> 13.0 - 13.3 us; K = 4096 (fc2) stays: 19.2 vs 26.0).  The decode-side
            // twins (W_frag / W_q8_frag: LLaMA rows) keep their kernels.
            static const int ring_env = getenv("VCLA_RING") ? atoi(getenv("VCLA_RING")) : 1;
            const char* rv_ = getenv("VCLA_RING_VIT"); const int ring_vit_env = rv_ ? atoi(rv_) : 1;      // read per call: tools/bench_kernels.py vit1 flips it
            if (ring_env && ring_vit_env && a->M > 16 && a->K <= 2048 && a->N >= 512 && !a->W_frag && !a->W_q8_frag && !a->W_q8 && !a->norm_gamma && !a->out_f32 &&
                a->epilogue != VCLA_EPI_SWIGLU && a->c_group_rows <= 0 && a->A) kernel = 11;
        }
        else {
            // Ragged M (ViT: M = B*257): peel the M % 256 <= 128 tail rows into their own small launch so the 256x256 kernel
            // runs WHOLE rounds (e.g. 65 x 4 tiles = 1.02 rounds -> 64 x 4 = exactly one round + a 64-row panel GEMM).
            const int rem = a->M % 256;
            if (rem > 0 && rem <= 128 && a->M > 256 && a->N >= 256 && a->c_group_rows <= 0) {
                vcla_gemm_args head = *a, tail = *a;
                head.M = a->M - rem;
                if (prefer_256(&head)) {
                    static const int pf_env = getenv("VCLA_GEMM_PF") ? atoi(getenv("VCLA_GEMM_PF")) : 1;
                    static const int xr_env = getenv("VCLA_GEMM_XR") ? atoi(getenv("VCLA_GEMM_XR")) : 1;
                    if (pf_env && xr_env && vcla_gemm_tile257_ok(a) && a->K >= 3 * GM_BK) {
                        // M = B * 257: ONE launch of 257-row tiles instead of whole 256-row rounds + a tail launch
                        head = *a;
                        head.force_kernel = 4;
                        head.post_norm_gamma = nullptr;                          // the wrapper normalises all of C afterwards
                        return gemm_impl(&head, dtype, stream);
                    }
                    const size_t es = 2, cs = a->out_f32 ? 4 : 2;
                    tail.M = rem;
                    tail.A = (const char*)a->A + (size_t)head.M * a->lda * es;
                    tail.C = (char*)a->C + (size_t)head.M * a->ldc * cs;
                    if (a->residual) tail.residual = (const char*)a->residual + (size_t)head.M * a->ldr * es;
                    head.force_kernel = 4;
                    // the tail: one launch of the skinny kernel (intra-workgroup split-K, epilogue in the kernel) for short K, the split-K
                    // panel kernel + its reduce launch for long K.  Measured at 64 rows (tools/bench_kernels.py vittail): K = 1024:
                    // 9.7 - 10.8 us vs 11.6 - 13.6 us; K = 4096: 28.1 vs 15.6 us.  VCLA_TAIL_KERNEL = 7 / 8 forces one form.
                    static const int tail_env = getenv("VCLA_TAIL_KERNEL") ? atoi(getenv("VCLA_TAIL_KERNEL")) : 0;
                    if (!a->post_norm_gamma && (tail_env == 7 || (tail_env == 0 && a->K <= 2048))) tail.force_kernel = 7;
                    head.post_norm_gamma = tail.post_norm_gamma = nullptr;   // the wrapper normalises all of C afterwards
                    int rc = gemm_impl(&head, dtype, stream);
                    return rc ? rc : gemm_impl(&tail, dtype, stream);
                }
            }
            kernel = prefer_256(a) ? 4 : 1;
            // 129 - 256 rows (a LLaMA decode batch of that many sequences, a prefill of that many prompt rows): the intake-bound ring kernel
            // (gemm_ring.hip) -- full K per tile, no split-K partials.  VCLA_RING=0: the round-4 dispatch (128 x 128 tiles + K slices).
            static const int ring_env = getenv("VCLA_RING") ? atoi(getenv("VCLA_RING")) : 1;
            if (ring_env && a->M <= 256 && (a->epilogue == VCLA_EPI_NONE || (a->epilogue == VCLA_EPI_SWIGLU && !a->out_f32)) && a->c_group_rows <= 0) kernel = 11;
            // ONE image through the ViT (257 rows, K = 1024: qkv / out / fc1): 72 - 96 tiles of 128 x 128 needed K slices + a reduce launch to fill the chip; 64 x 64
            // (128 x 96) ring tiles over the full K fill it in one launch with the bias / GELU / residual in the tile's epilogue.  Graph-replayed, rotating weights,
            // M = 257: qkv 25.4 -> 14.0 us, out 17.8 -> 13.5, fc1 28.0 -> 18.0; fc2 (K = 4096) stays on the K slices (24.2 vs 26.2).  VCLA_RING_VIT=0: off.
            const char* rv_ = getenv("VCLA_RING_VIT"); const int ring_vit_env = rv_ ? atoi(rv_) : 1;      // read per call: tools/bench_kernels.py vit1 flips it
            if (ring_env && ring_vit_env && a->M <= 320 && a->K <= 2048 && !a->out_f32 && a->epilogue != VCLA_EPI_SWIGLU && a->c_group_rows <= 0 && a->A && !a->W_q8) kernel = 11;
        }
    }
    VCLA_REQUIRE(kernel >= 1 && kernel <= 14, VCLA_ERR_BAD_ARG, "gemm: bad force_kernel %d", a->force_kernel);
    VCLA_REQUIRE(!(kernel > 12 && kernel <= 14 && a->epilogue == VCLA_EPI_SWIGLU), VCLA_ERR_BAD_ARG, "gemm: the ring kernel's SwiGLU tile is 256 x 96 (force_kernel 11 / 12)");
    if (kernel == 10) {
        VCLA_REQUIRE(dtype == VCLA_BF16 && a->A_q8 && a->a_scale && a->W_q8 && a->w_scale && vcla_aligned(a->A_q8, 16) && vcla_aligned(a->W_q8, 16) &&
                         a->K % 128 == 0, VCLA_ERR_BAD_ARG, "gemm: the fp8 MFMA kernel needs A_q8 + a_scale, W_q8 + w_scale (16-byte aligned) and K %% 128 == 0 (K=%d)", a->K);
        VCLA_REQUIRE(a->C && !a->C_frag && !a->A_frag && !a->norm_gamma && !a->c_row_ssq && !a->a_row_ssq, VCLA_ERR_BAD_ARG,
                     "gemm: the fp8 MFMA kernel writes a row-major C and takes no fused norms");
    } else if (kernel >= 11) {
        VCLA_REQUIRE(dtype == VCLA_BF16 && (a->A || a->A_slab) && a->C && (a->epilogue == VCLA_EPI_NONE || !a->out_f32) && !a->C_frag &&
                         !a->A_frag && !a->A_q8 && !a->a_scale && !a->norm_gamma && !a->c_row_ssq && !a->a_row_ssq && !a->c_frag_gamma, VCLA_ERR_BAD_ARG,
                     "gemm: the ring kernel takes a row-major bf16 A, epilogue NONE (bf16 / fp32 C) or SWIGLU / a GELU (bf16 C), no fused norms");
        VCLA_REQUIRE(!(a->W_q8 || a->W_q8_slab) || (a->w_scale && vcla_aligned(a->W_q8, 16) && vcla_aligned(a->W_q8_slab, 16)), VCLA_ERR_BAD_ARG,
                     "gemm: the ring kernel's fp8 weights need W_q8 or W_q8_slab (16-byte aligned) + w_scale");
        VCLA_REQUIRE(!a->W_slab || vcla_aligned(a->W_slab, 16), VCLA_ERR_BAD_ARG, "gemm: W_slab must be 16-byte aligned");
    } else if (kernel == 9) {
        VCLA_REQUIRE(dtype == VCLA_BF16 && a->A_frag && vcla_aligned(a->A_frag, 16) && a->M <= 64 && (a->W_frag || a->W_q8_frag), VCLA_ERR_BAD_ARG,
                     "gemm: the streaming kernel needs bf16, A_frag, M <= 64 (got %d) and W_frag or W_q8_frag", a->M);
        VCLA_REQUIRE((a->epilogue == VCLA_EPI_NONE || (a->epilogue == VCLA_EPI_SWIGLU && !a->out_f32)) && a->c_group_rows <= 0 && !a->norm_gamma &&
                         !a->post_norm_gamma, VCLA_ERR_BAD_ARG, "gemm: the streaming kernel implements epilogues NONE / SWIGLU, no row regrouping, no fused norms");
        VCLA_REQUIRE(!a->C_frag || (!a->out_f32 && ((a->epilogue == VCLA_EPI_SWIGLU ? a->N / 2 : a->N) % 32 == 0) && vcla_aligned(a->C_frag, 16)),
                     VCLA_ERR_BAD_ARG, "gemm: C_frag needs a bf16 output whose width is a multiple of 32");
        VCLA_REQUIRE(!a->c_frag_gamma || a->C_frag, VCLA_ERR_BAD_ARG, "gemm: c_frag_gamma without C_frag");
        VCLA_REQUIRE(!a->c_row_ssq || (a->epilogue == VCLA_EPI_NONE && !a->out_f32 && a->N % 16 == 0), VCLA_ERR_BAD_ARG,
                     "gemm: c_row_ssq needs epilogue NONE, a bf16 output and N %% 16 == 0");
        VCLA_REQUIRE(!a->a_row_ssq || a->a_row_ssq_parts > 0, VCLA_ERR_BAD_ARG, "gemm: a_row_ssq needs a_row_ssq_parts > 0");
        VCLA_REQUIRE(!(a->a_row_ssq && a->ds_splitk > 1 && !a->ds_raw_partials), VCLA_ERR_BAD_ARG,
                     "gemm: a_row_ssq (deferred RMSNorm of the A operand) cannot be combined with ds_splitk > 1: the split-K reduce launch does not apply rstd");
        VCLA_REQUIRE(a->ds_splitk <= 1 || (a->epilogue == VCLA_EPI_NONE && a->ds_splitk <= 16 && a->N % 4 == 0 && (a->ldc % 4 == 0 || !a->C) && a->splitk_ws &&
                                           a->splitk_ws_bytes >= (size_t)a->ds_splitk * a->M * a->N * 4 && (!a->residual || a->ldr % 4 == 0) &&
                                           (!a->c_row_ssq || a->N % 16 == 0) && a->K / (a->W_q8_frag ? 64 : 32) >= a->ds_splitk),
                     VCLA_ERR_BAD_ARG, "gemm: ds_splitk needs epilogue NONE, N %% 4 == 0, 4-element aligned rows and a workspace of ds_splitk * M * N * 4 bytes");
    } else {
        VCLA_REQUIRE((a->A || (kernel == 4 && a->A_slab)) && a->C && !a->C_frag && !a->c_frag_gamma && !a->c_row_ssq && !a->a_row_ssq && !a->A_q8 && !a->a_scale, VCLA_ERR_BAD_ARG,
                     "gemm: A_frag / C_frag / deferred-norm fields belong to the streaming kernel (9), A_q8 / a_scale to the fp8 MFMA kernel (10)");
        VCLA_REQUIRE(!(kernel == 4 && (a->A_slab || a->W_slab)) || (a->K >= 3 * GM_BK && !a->W_q8_slab && (!a->W_slab || vcla_aligned(a->W_slab, 16))), VCLA_ERR_BAD_ARG,
                     "gemm: the 256 x 256 kernel reads slab-major bf16 operands in its direct-to-LDS form only (K >= 192)");
    }
    VCLA_REQUIRE(!((kernel == 1 || kernel == 4 || kernel == 5 || kernel == 7 || kernel == 8 || kernel >= 11) && dtype != VCLA_BF16), VCLA_ERR_BAD_DTYPE, "gemm: MFMA kernels need bf16 activations");
    VCLA_REQUIRE(!(kernel == 3 && dtype != VCLA_F32), VCLA_ERR_BAD_DTYPE, "gemm: fp32 tile kernel needs fp32 activations");
    VCLA_REQUIRE(!((kernel == 2 || kernel == 6) && a->M > 8), VCLA_ERR_BAD_SHAPE, "gemm: GEMV kernel needs M <= 8 (got %d)", a->M);
    VCLA_REQUIRE(!a->norm_gamma || kernel == 2 || kernel == 6, VCLA_ERR_BAD_ARG, "gemm: the fused RMSNorm prologue exists only in the GEMV kernel (M <= 8)");
    VCLA_REQUIRE(!a->norm_gamma || vcla_aligned(a->norm_gamma, 16), VCLA_ERR_BAD_ARG, "gemm: norm_gamma must be 16-byte aligned");
    VCLA_REQUIRE(!((kernel == 7 || kernel == 8) && a->M > 128), VCLA_ERR_BAD_SHAPE, "gemm: skinny / panel kernels need M <= 128 (got %d)", a->M);
    VCLA_REQUIRE(!a->W_frag || (vcla_aligned(a->W_frag, 16) && a->K % 32 == 0), VCLA_ERR_BAD_ARG, "gemm: W_frag must be 16-byte aligned");
    VCLA_REQUIRE((!a->W_q8 && !a->W_q8_frag && !a->W_q8_slab) || (a->w_scale && dtype == VCLA_BF16), VCLA_ERR_BAD_ARG,
                 "gemm: fp8 weights need w_scale and bf16 activations");
    VCLA_REQUIRE(!a->w_scale || a->W_q8 || a->W_q8_frag || a->W_q8_slab, VCLA_ERR_BAD_ARG, "gemm: w_scale without fp8 weights");
    VCLA_REQUIRE(!(a->W_q8 || a->W_q8_frag) || kernel == 2 || kernel == 8 || kernel == 9 || kernel == 10 || kernel >= 11, VCLA_ERR_BAD_ARG,
                 "gemm: fp8 weights are implemented for the M = 1 GEMV (needs W_q8), the M <= 128 panel kernel (needs W_q8_frag) and the ring kernel (W_q8)");
    VCLA_REQUIRE(!(kernel == 2 && a->w_scale) || (a->W_q8 && gemv1_applicable(a, dtype)), VCLA_ERR_BAD_ARG,
                 "gemm: fp8 GEMV needs W_q8, M = 1, bf16, epilogue NONE/SWIGLU");
    VCLA_REQUIRE(!(kernel == 8 && a->w_scale) || a->W_q8_frag, VCLA_ERR_BAD_ARG, "gemm: fp8 panel kernel needs W_q8_frag");
    VCLA_REQUIRE(!a->splitk_ws || vcla_aligned(a->splitk_ws, 16), VCLA_ERR_BAD_ARG, "gemm: splitk_ws must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    switch (a->epilogue) {
        case VCLA_EPI_NONE: return dispatch_epi<VCLA_EPI_NONE>(a, dtype, kernel, s);
        case VCLA_EPI_QUICK_GELU: return dispatch_epi<VCLA_EPI_QUICK_GELU>(a, dtype, kernel, s);
        case VCLA_EPI_GELU_ERF: return dispatch_epi<VCLA_EPI_GELU_ERF>(a, dtype, kernel, s);
        default: return dispatch_epi<VCLA_EPI_SWIGLU>(a, dtype, kernel, s);
    }
}
