// gemm_stream.hip -- the batch-decode GEMM (2 <= M <= 64 rows): C = epilogue(A . W^T) with W streamed from HBM exactly once.
//
// Why a separate kernel.  At M <= 64 every LLaMA projection is a pure weight stream (13.4 GB per decode step shared by the whole
// batch) and one CU can pull only ~11 B/clk from HBM (MI355X_MICROARCH.md: ~10 B/cyc/CU), so the stream reaches the chip's
// ~6 TB/s only when ALL 256 CUs pull an EQUAL share for the WHOLE launch.  The split-K panel kernel (gemm.hip, kernel 8) ran the
// gate/up GEMM on 172 workgroups (22016 / 128 column tiles: a third of the chip idle, 4.4 TB/s) and paid an fp32 partial round
// trip + a second launch for o_proj / down_proj (profiles/r01_bench_b64_kernel_stats.csv: 8 % of the step).  Here:
//   * grid = 256 workgroups (one per CU), workgroup g owns the 16-column weight tiles [g*T/256, (g+1)*T/256) over the FULL K:
//     equal bytes per CU (within one tile), no split-K partials, bias / SwiGLU / residual applied in the launch;
//   * BOTH operands arrive in MFMA-fragment-major order (W_frag / W_q8_frag packed at load; A_frag written by the producer:
//     vcla_rmsnorm_pack, vcla_attn_decode_fused(out_frag), or the C_frag epilogue of the previous GEMM), so every operand fetch
//     is ONE contiguous 1 KiB wave load straight into MFMA registers: no LDS, no barrier in the K loop.  The fetches are buffer
//     loads (uniform descriptor + one per-lane offset register + a scalar byte offset per load): the address arithmetic of the
//     whole K loop runs on the scalar unit and the vector registers hold nothing but the operand ring and the accumulators;
//   * the 8 waves of a workgroup interleave the K stages (wave w takes stages w, w+8, ...), each with its own register ring of
//     D stages in flight; the hot loop is branch-free (the tile count is a template parameter, the K tail is peeled) so the
//     compiler emits counted vmcnt waits; the partial tiles meet once, through LDS, after the stream (fixed summation order).
// The activations (M x K bf16 <= 1.4 MB) are re-read by every workgroup from L2; that traffic (134 MB for K = 4096) rides under
// the HBM stream (L2 ~34 TB/s).  (Measured and dropped: starting every workgroup at a different K offset so that the 32 CUs
// of an XCD do not ask the L2 for the same activation line at the same time -- no change; the ~20 B/clk a CU can ingest is the limit.)  Algorithmic bytes per launch = N_pad * K * 2 (bf16) or N_pad * K (fp8).
#include "vcla_common.h"
#include "gemm_epilogue.h"
#include <stdlib.h>

#define DS_WAVES 8
#define DS_ROUND 8   // epilogue units reduced per LDS round (one per wave)

// 8 OCP fp8 (e4m3fn) values in two dwords -> one bf16x8 MFMA operand (exact: e4m3 fits in bf16)
__device__ __forceinline__ bf16x8_t ds_fp8x8_to_bf16x8(uint32_t lo, uint32_t hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
    const f32x2_t c = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
    const u32x4_t p = {pack_bf2(a.x, a.y), pack_bf2(b.x, b.y), pack_bf2(c.x, c.y), pack_bf2(d.x, d.y)};
    return __builtin_bit_cast(bf16x8_t, p);
}

// ring depth: as many stages in flight as fit a ~200-register budget next to the accumulators (acc = NT*MT*4 registers,
// a stage = (KS*MT + NT)*4), a power of two where possible (K / 256 stages per wave is a power of two for K = 4096: the peeled
// tail then issues no loads), at most 4
constexpr int ds_depth(int MT, int NT, bool FP8, int budget = 0) {
    const int acc = NT * MT * 4, st = ((FP8 ? 2 : 1) * MT + NT) * 4;
    if (FP8 && MT == 3 && NT == 6) return 1;        // (fp8 gate/up at 33 - 48 rows: a second stage spills two registers to scratch -- no product kernel uses scratch)
    const int d = ((budget ? budget : (FP8 ? 176 : 200)) - acc) / st;   // fp8: the in-register conversion needs temporaries
    return d >= 4 ? 4 : (d >= 3 ? 3 : (d >= 2 ? 2 : 1));   // 4 stages x 8 waves already keep > 100 KiB per CU in flight
}

struct DsCtx {
    __amdgpu_buffer_rsrc_t rA, rW;
    unsigned voff;            // lane * 16
    unsigned a_stage_bytes;   // A fragments of one stage: KS k-steps x MT tiles x 1 KiB
    unsigned w_tile_bytes;    // one 16-row weight tile over the full K
    int wave, lane, nst, mt_c;
    int s_beg;                // first K stage of this workgroup's slice (split-K; 0 without)
    int ks;                   // slice index, splitk slices in total
    int splitk;
    float* partial;           // [splitk][M][N] fp32 partial tiles (split-K only)
    int M, N;
};

// deferred RMSNorm, consumer side: rstd of every activation row from the producer's per-tile sums of squares.  Called once per
// workgroup after the K loop of its first chunk: wave w owns rows 8w .. 8w+7, 8 lanes per row, every lane's loads issued back to
// back (one L2 round trip, hidden behind the slab writes of the cross-wave reduction); fixed summation order (lane-strided
// partial sums, xor-shuffle tree).  The result lands in LDS; the reduction's barrier publishes it to the other waves.
__device__ __forceinline__ void ds_row_rstd(const vcla_gemm_args& a, const DsCtx& c, float* rstd_s) {
    const int parts = a.a_row_ssq_parts;
    const int r = c.wave * 8 + (c.lane >> 3), seg = c.lane & 7;
    float q = 0.f;
    if (r < a.M) {
        const float* src = a.a_row_ssq + (int64_t)r * parts;
        if ((parts & 31) == 0) {
            const int per = parts >> 3;          // contiguous run of this lane, a multiple of 4
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = k * 4 < per ? *reinterpret_cast<const float4*>(src + seg * per + k * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 8; ++k) q += (v[k].x + v[k].y) + (v[k].z + v[k].w);
            for (int k = 32; k < per; k += 4) { const float4 t = *reinterpret_cast<const float4*>(src + seg * per + k); q += (t.x + t.y) + (t.z + t.w); }
        } else {
            for (int p = seg; p < parts; p += 8) q += src[p];
        }
    }
    q += __shfl_xor(q, 1, 64);
    q += __shfl_xor(q, 2, 64);
    q += __shfl_xor(q, 4, 64);
    if (seg == 0 && r < 64) rstd_s[r] = r < a.M ? rsqrtf(q / (float)a.K + a.a_norm_eps) : 0.f;
}

// One chunk of NT weight tiles [c0, c0 + NT) x all rows, K stages wave, wave + 8, ... of this wave; then the cross-wave reduction
// and the epilogue.  MT = 16-row tiles of A; FP8: W_q8_frag (two k-steps per 16-byte lane load) instead of W_frag.
template <int EPI, typename OutT, int MT, int NT, bool FP8, int BUD = 0>
__device__ __forceinline__ void ds_chunk(const DsCtx& c, int c0, f32x4_t* slab, float* rstd_s, bool first) {
    constexpr int TPU = EPI == VCLA_EPI_SWIGLU ? 2 : 1;       // tiles per epilogue unit (SwiGLU: gate tile + up tile)
    constexpr int KS = FP8 ? 2 : 1;
    constexpr int D = ds_depth(MT, NT, FP8, BUD);
    static_assert(NT % TPU == 0, "SwiGLU chunks hold whole gate/up pairs");
    f32x4_t acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    u32x4_t ra[D][KS][MT], rw[D][NT];
    const unsigned w_chunk_off = (unsigned)c0 * c.w_tile_bytes;
    const int T = c.nst;

    // stage t of this wave = global stage ks = wave + 8 t
#define DS_LOAD(s_, t_)                                                                                         \
    {                                                                                                           \
        const unsigned ks_ = (unsigned)(c.s_beg + c.wave + DS_WAVES * (t_));                                    \
        const unsigned ao_ = ks_ * c.a_stage_bytes, wo_ = w_chunk_off + (ks_ << 10);                            \
        _Pragma("unroll") for (int q = 0; q < KS; ++q)                                                          \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                      \
                ra[s_][q][i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(c.rA, c.voff, ao_ + ((q * c.mt_c + i) << 10), 0)); \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                          \
            rw[s_][j] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(c.rW, c.voff, wo_ + j * c.w_tile_bytes, 2 /* nt */)); \
    }
#define DS_COMPUTE(s_)                                                                                          \
    {                                                                                                           \
        _Pragma("unroll") for (int q = 0; q < KS; ++q)                                                          \
            _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                    \
                const bf16x8_t wf_ = FP8 ? (q == 0 ? ds_fp8x8_to_bf16x8(rw[s_][j].x, rw[s_][j].y)                \
                                                   : ds_fp8x8_to_bf16x8(rw[s_][j].z, rw[s_][j].w))              \
                                         : __builtin_bit_cast(bf16x8_t, rw[s_][j]);                             \
                _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                  \
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf_, __builtin_bit_cast(bf16x8_t, ra[s_][q][i]), acc[j][i], 0, 0, 0); \
            }                                                                                                   \
    }
    if (T >= D) {
        // prologue: D stages in flight before the first MFMA
#pragma unroll
        for (int s = 0; s < D; ++s) DS_LOAD(s, s)
        // steady state: every stage of the ring is computed and refilled; branch-free -> counted vmcnt waits
        int t0 = 0;
        for (; t0 + 2 * D <= T; t0 += D) {
#pragma unroll
            for (int s = 0; s < D; ++s) {
                DS_COMPUTE(s)
                DS_LOAD(s, t0 + s + D)
            }
        }
        // peeled tail: T - t0 in [D, 2D) stages left, the first D of them already in the ring
#pragma unroll
        for (int s = 0; s < D; ++s) {
            DS_COMPUTE(s)
            if (t0 + s + D < T) DS_LOAD(s, t0 + s + D)
        }
#pragma unroll
        for (int s = 0; s < D; ++s)
            if (t0 + D + s < T) DS_COMPUTE(s)
    } else {
        // short K (< 256 * D; no LLaMA shape): one stage at a time
        for (int t = 0; t < T; ++t) {
            DS_LOAD(0, t)
            DS_COMPUTE(0)
        }
    }
#undef DS_LOAD
#undef DS_COMPUTE

    // ---- the 8 K-interleaved partial tiles meet in LDS; wave w reduces and finishes epilogue unit (round base + w).
    // unit u of the chunk = (tile group jj = u / MT, row tile i = u % MT); SwiGLU units carry the gate and the up tile.
    // Everything below reads the argument block through the kernarg segment pointer, laundered so that the optimiser cannot
    // hoist the loads: the ~40 scalar registers of epilogue-only arguments are loaded HERE instead of being held -- and
    // spilled -- across the K loop (gemm_dstream_kernel's first argument IS the vcla_gemm_args block, offset 0).
    typedef const __attribute__((address_space(4))) vcla_gemm_args* kernarg_p;   // the argument block is the first kernel argument
    kernarg_p ap_ = (kernarg_p)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ap_));
    vcla_gemm_args a;
    __builtin_memcpy(&a, (const void*)ap_, sizeof(a));
    if (first && a.a_row_ssq) ds_row_rstd(a, c, rstd_s);
    constexpr int NU = (NT / TPU) * MT;
#pragma unroll
    for (int r0 = 0; r0 < NU; r0 += DS_ROUND) {
#pragma unroll
        for (int uu = 0; uu < DS_ROUND; ++uu) {
            if (r0 + uu < NU) {   // compile-time after unrolling: the accumulator indices below are literals
                const int u = r0 + uu;
#pragma unroll
                for (int tt = 0; tt < TPU; ++tt) slab[((c.wave * DS_ROUND + uu) * TPU + tt) * 64 + c.lane] = acc[(u / MT) * TPU + tt][u % MT];
            }
        }
        __syncthreads();
        const int u = r0 + c.wave;
        if (u < NU) {
            f32x4_t sum[1][TPU];
#pragma unroll
            for (int tt = 0; tt < TPU; ++tt) sum[0][tt] = slab[((0 * DS_ROUND + c.wave) * TPU + tt) * 64 + c.lane];
#pragma unroll
            for (int w2 = 1; w2 < DS_WAVES; ++w2)
#pragma unroll
                for (int tt = 0; tt < TPU; ++tt) {
                    const f32x4_t p = slab[((w2 * DS_ROUND + c.wave) * TPU + tt) * 64 + c.lane];
                    sum[0][tt][0] += p[0]; sum[0][tt][1] += p[1]; sum[0][tt][2] += p[2]; sum[0][tt][3] += p[3];
                }
            const int jj = u / MT, i = u - jj * MT;
            if (c.splitk > 1) {
                // split-K: the raw fp32 sums of this K slice go to the workspace; ds_reduce_kernel finishes the tile
                if constexpr (TPU == 1) {
                    const int m = i * 16 + (c.lane & 15), n = (c0 + jj) * 16 + (c.lane >> 4) * 4;
                    if (m < c.M && n < c.N)      // N % 4 == 0 is checked on the host
                        *reinterpret_cast<f32x4_t*>(c.partial + ((int64_t)c.ks * c.M + m) * c.N + n) = sum[0][0];
                }
            } else {
                if (a.a_row_ssq) {   // deferred RMSNorm, consumer side: the lane's 4 values belong to row i*16 + (lane & 15)
                    const float rs = rstd_s[i * 16 + (c.lane & 15)];
#pragma unroll
                    for (int tt = 0; tt < TPU; ++tt) { sum[0][tt][0] *= rs; sum[0][tt][1] *= rs; sum[0][tt][2] *= rs; sum[0][tt][3] *= rs; }
                }
                // the lane id is laundered: everything the epilogue derives from it (column quads, fragment offsets) is recomputed HERE instead
                // of being computed at kernel entry and carried -- spilled to scratch, in the M = 64 SwiGLU instance -- across the K loop
                int ln = c.lane;
                asm volatile("" : "+v"(ln));
                gemm_epilogue<EPI, OutT, 1, TPU>(a, sum, i * 16, (c0 + jj * TPU) * 16, ln);
            }
        }
        __syncthreads();
    }
}

// A workgroup walks its share of tiles in chunks of at most 4 tiles (6 = three gate/up pairs for SwiGLU); every chunk size has
// its own branch-free instantiation of the loop.
// WIDE (epilogue NONE, split-K with raw partials: the qkv GEMM whose consumer -- the decode attention -- sums the slices itself): chunks
// of up to 6 tiles, so that a group of 2 workgroups covers its 6 tiles of the 768 in ONE pass over its half of K and each CU reads half
// the activation panel once (the point of the exercise: a CU's intake, not HBM, bounds these GEMMs).  A separate instantiation: the
// default kernel's code (and register allocation) stays what it was.
template <int EPI, typename OutT, int MT, bool FP8, bool WIDE = false>
__global__ __launch_bounds__(DS_WAVES * 64) void gemm_dstream_kernel(vcla_gemm_args a, int units_total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ds_smem[];
    f32x4_t* slab = reinterpret_cast<f32x4_t*>(ds_smem);      // [wave][unit in round][tile of unit][lane]
    constexpr int TPU = EPI == VCLA_EPI_SWIGLU ? 2 : 1;
    float* rstd_s = reinterpret_cast<float*>(ds_smem + (size_t)DS_WAVES * DS_ROUND * TPU * 64 * sizeof(f32x4_t));   // [64] behind the slabs
    constexpr int NTW = (EPI == VCLA_EPI_SWIGLU || WIDE) ? 6 : 4;
    constexpr int KS = FP8 ? 2 : 1;
    static_assert(!WIDE || EPI == VCLA_EPI_NONE, "wide chunks: epilogue NONE");
    DsCtx c;
    c.lane = threadIdx.x & 63;
    c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = a.ds_splitk > 1 ? a.ds_splitk : 1;
    const int G = gridDim.x / S, g = blockIdx.x / S;           // G groups of S workgroups: one K slice each, the same tiles
    c.splitk = S; c.ks = blockIdx.x - g * S; c.partial = (float*)a.splitk_ws; c.M = a.M; c.N = a.N;
    const int t_beg = (int)((int64_t)g * units_total / G) * TPU, t_end = (int)((int64_t)(g + 1) * units_total / G) * TPU;
    const int KST = a.K / (32 * KS);                           // stages along K
    c.s_beg = (int)((int64_t)c.ks * KST / S);
    const int s_len = (int)((int64_t)(c.ks + 1) * KST / S) - c.s_beg;
    c.nst = (s_len - c.wave + DS_WAVES - 1) / DS_WAVES;        // stages of this wave inside the slice: s_beg + wave + 8 t
    c.mt_c = (a.M + 15) >> 4;                                  // == MT (the launcher instantiates MT = ceil(M/16))
    const int n_pad = (a.N + 127) / 128 * 128;
    c.w_tile_bytes = (unsigned)a.K * (FP8 ? 16u : 32u);
    c.rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.A_frag), 0, (int)((int64_t)c.mt_c * 16 * a.K * 2), 0x00020000);
    c.rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(FP8 ? a.W_q8_frag : a.W_frag), 0, (int)((int64_t)(n_pad / 16) * c.w_tile_bytes), 0x00020000);
    c.voff = c.lane * 16;
    c.a_stage_bytes = (unsigned)(KS * c.mt_c) << 10;

    for (int c0 = t_beg; c0 < t_end; c0 += NTW) {
        const int nt = (t_end - c0) < NTW ? (t_end - c0) : NTW;     // tiles of this chunk (workgroup-uniform)
        if constexpr (TPU == 2) {
            if (nt == 6) ds_chunk<EPI, OutT, MT, 6, FP8>(c, c0, slab, rstd_s, c0 == t_beg);
            else if (nt == 4) ds_chunk<EPI, OutT, MT, 4, FP8>(c, c0, slab, rstd_s, c0 == t_beg);
            else ds_chunk<EPI, OutT, MT, 2, FP8>(c, c0, slab, rstd_s, c0 == t_beg);
        } else {
            if constexpr (WIDE) {      // the launcher guarantees whole chunks of 6 (one instantiation: no spill-prone dispatch over six variants)
                ds_chunk<EPI, OutT, MT, 6, FP8>(c, c0, slab, rstd_s, c0 == t_beg);   // ring depth 2 (a third stage next to 96 accumulators spills 21 registers)
                continue;
            }
            if constexpr (sizeof(OutT) == 4) {
                // fp32 output = the lm_head (49958 columns: 12 - 13 tiles per workgroup).  Every chunk is one more pass over the activation panel
                // (512 KB per CU at M = 64, ~9 us by the intake model): chunks of 6 tiles while 6 remain -> 2 - 3 passes instead of 4
                if (t_end - c0 >= 6) {
                    ds_chunk<EPI, OutT, MT, 6, FP8>(c, c0, slab, rstd_s, c0 == t_beg);
                    c0 += 6 - NTW;
                    continue;
                }
            }
            if (nt == 4) ds_chunk<EPI, OutT, MT, 4, FP8>(c, c0, slab, rstd_s, c0 == t_beg);
            else if (nt == 3) ds_chunk<EPI, OutT, MT, 3, FP8>(c, c0, slab, rstd_s, c0 == t_beg);
            else if (nt == 2) ds_chunk<EPI, OutT, MT, 2, FP8>(c, c0, slab, rstd_s, c0 == t_beg);
            else ds_chunk<EPI, OutT, MT, 1, FP8>(c, c0, slab, rstd_s, c0 == t_beg);
        }
    }
}


// Second launch of a split-K streaming GEMM: out[m, n] = sum over the K slices (in slice order) of the fp32 partial tiles, then the
// epilogue of the unsplit kernel: bias, residual, rounded store to C, optional fragment-major copy (x gamma) for the next
// streaming GEMM and the per-row / per-16-column sums of squares of the deferred RMSNorm.  One thread per 4 consecutive columns
// (16-byte partial loads, fully parallel over M x N / 4 threads).
template <typename OutT>
__global__ __launch_bounds__(256) void ds_reduce_kernel(vcla_gemm_args a, int splitk) {
    const int n4 = a.N >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = idx < (int64_t)a.M * n4;
    const int m = live ? (int)(idx / n4) : 0, n = live ? (int)(idx - (int64_t)m * n4) * 4 : 0;
    const float* pp = (const float*)a.splitk_ws + (int64_t)m * a.N + n;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        for (int s = 0; s < splitk; ++s) {
            const float4 p = *reinterpret_cast<const float4*>(pp + (int64_t)s * a.M * a.N);
            v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
        }
        if (a.w_scale) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= a.w_scale[n + r];
        }
        if (a.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += a.bias[n + r];
        }
        if (a.residual) {
            float rv[4];
            Act<bf16_t>::ld4((const bf16_t*)a.residual + (int64_t)m * a.ldr + n, rv);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += rv[r];
        }
    }
    if (a.c_row_ssq) {   // the 4 threads of a 16-column tile are adjacent lanes (N % 16 == 0 -> a tile never straddles rows)
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float x = Act<OutT>::rnd(v[r]); q += live ? x * x : 0.f; }
        q += __shfl_xor(q, 1, 64);
        q += __shfl_xor(q, 2, 64);
        if ((a.N & 255) == 0) {
            // a wave = 256 consecutive columns of ONE row: reduce all the way, one partial per 256 columns (the consumer sums
            // N / 256 values per row instead of N / 16: 4 KB instead of 64 KB per workgroup at M = 64, N = 4096)
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) q += __shfl_xor(q, o, 64);
            if (live && (threadIdx.x & 63) == 0) a.c_row_ssq[(int64_t)m * (a.N >> 8) + (n >> 8)] = q;
        } else if (live && (threadIdx.x & 3) == 0) {
            a.c_row_ssq[(int64_t)m * (a.N >> 4) + (n >> 4)] = q;
        }
    }
    if (!live) return;
    if (a.C_frag) {
        const int mt_c = (a.M + 15) >> 4;
        bf16_t* fp = (bf16_t*)a.C_frag + ((((int64_t)(n >> 5) * mt_c + (m >> 4)) * 64 + ((n & 31) >> 3) * 16 + (m & 15)) << 3) + (n & 7);
        float f[4] = {v[0], v[1], v[2], v[3]};
        if (a.c_frag_gamma) {
#pragma unroll
            for (int r = 0; r < 4; ++r) f[r] = a.c_frag_gamma[n + r] * Act<OutT>::rnd(v[r]);
        }
        *reinterpret_cast<uint2*>(fp) = make_uint2(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]));
    }
    if (a.C) Act<OutT>::st4((OutT*)a.C + (int64_t)m * a.ldc + n, v);
}

template <int EPI, typename OutT, int MT, bool FP8>
static int ds_launch(const vcla_gemm_args* a, int units, int grid, hipStream_t s) {
    constexpr int TPU = EPI == VCLA_EPI_SWIGLU ? 2 : 1;
    const size_t lds = (size_t)DS_WAVES * DS_ROUND * TPU * 64 * sizeof(f32x4_t) + 64 * sizeof(float);   // 64 KiB (128 KiB for SwiGLU) + rstd[64]
    if constexpr (EPI == VCLA_EPI_NONE && MT >= 3 && sizeof(OutT) == 2) {
        const int groups_ = grid / (a->ds_splitk > 1 ? a->ds_splitk : 1);
        if (a->ds_splitk > 1 && a->ds_raw_partials && units % groups_ == 0 && (units / groups_) % 6 == 0) {
            // raw fp32 slices for a consumer that sums them (vcla_attn_decode_fused_parts), every tile group a whole number of 6-tile chunks: the wide kernel
            auto kw = gemm_dstream_kernel<EPI, OutT, MT, FP8, true>;
            static bool attr_w[VCLA_MAX_DEVICES] = {};
            { const int rc_ = vcla_raise_dyn_lds((const void*)kw, lds, attr_w); if (rc_) return rc_; }
            kw<<<grid, DS_WAVES * 64, lds, s>>>(*a, units);
            VCLA_CHECK_LAUNCH("gemm_dstream_kernel<wide>");
            return VCLA_OK;
        }
    }
    auto kern = gemm_dstream_kernel<EPI, OutT, MT, FP8>;
    static bool attr_set[VCLA_MAX_DEVICES] = {};   // per instantiation and device
    { const int rc_ = vcla_raise_dyn_lds((const void*)kern, lds, attr_set); if (rc_) return rc_; }
    kern<<<grid, DS_WAVES * 64, lds, s>>>(*a, units);
    VCLA_CHECK_LAUNCH("gemm_dstream_kernel");
    if constexpr (EPI == VCLA_EPI_NONE) {
        if (a->ds_splitk > 1 && !a->ds_raw_partials) {
            const int64_t work = (int64_t)a->M * (a->N / 4);
            ds_reduce_kernel<OutT><<<(unsigned)((work + 255) / 256), 256, 0, s>>>(*a, a->ds_splitk);
            VCLA_CHECK_LAUNCH("ds_reduce_kernel");
        }
    }
    return VCLA_OK;
}

template <int EPI, typename OutT, bool FP8>
static int ds_pick_mt(const vcla_gemm_args* a, int units, int grid, hipStream_t s) {
    const int mt = (a->M + 15) / 16;
    if (mt <= 1) return ds_launch<EPI, OutT, 1, FP8>(a, units, grid, s);
    if (mt == 2) return ds_launch<EPI, OutT, 2, FP8>(a, units, grid, s);
    if (mt == 3) return ds_launch<EPI, OutT, 3, FP8>(a, units, grid, s);
    return ds_launch<EPI, OutT, 4, FP8>(a, units, grid, s);
}

// called by vcla_gemm (gemm.hip) for kernel 9; arguments were validated there
int vcla_gemm_dstream_launch(const vcla_gemm_args* a, hipStream_t s) {
    const bool fp8 = a->W_q8_frag != nullptr;
    const bool swiglu = a->epilogue == VCLA_EPI_SWIGLU;
    const int tiles = (a->N + 15) / 16;                        // W_frag rows exist up to N_pad (multiple of 128) >= tiles * 16
    const int units = swiglu ? tiles / 2 : tiles;              // SwiGLU: N % 32 == 0
    static const int grid_env = getenv("VCLA_DS_GRID") ? atoi(getenv("VCLA_DS_GRID")) : 256;   // one workgroup per CU
    int grid = units < grid_env ? units : grid_env;
    if (a->ds_splitk > 1) {   // groups of ds_splitk workgroups share a tile range: keep the launch at one workgroup per CU
        int groups = grid_env / a->ds_splitk;
        if (groups < 1) groups = 1;
        if (groups > units) groups = units;
        grid = groups * a->ds_splitk;
    }
#define DS_GO(EPI_, OUT_) return fp8 ? ds_pick_mt<EPI_, OUT_, true>(a, units, grid, s) : ds_pick_mt<EPI_, OUT_, false>(a, units, grid, s)
    if (swiglu) { DS_GO(VCLA_EPI_SWIGLU, bf16_t); }
    if (a->out_f32) { DS_GO(VCLA_EPI_NONE, float); }
    DS_GO(VCLA_EPI_NONE, bf16_t);
#undef DS_GO
}
