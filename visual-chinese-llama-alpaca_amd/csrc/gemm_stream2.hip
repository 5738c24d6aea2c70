// gemm_stream2.hip -- batch-decode GEMM (2 <= M <= 64, bf16 weights), second form: loader waves + deep weight ring (round 3).
//
// What gemm_stream.hip's kernel leaves on the table.  There every wave fetches BOTH operands of its K stages into registers:
// MT activation fragments (L2 hits, ~0.5 us away) and NT weight fragments (HBM, 2-3 us away under load) per stage, ring depth <= 4.
// A wave's loads return in order (one vmcnt counter), so the weight stream can only run as far ahead as the activation stream does,
// and the activations -- 4 KiB per stage at M = 64 -- take the register budget: o_proj / down_proj (one 16-column tile per
// workgroup, NT = 1) keep 4 KiB of weights in flight per wave, 32 KiB per CU, against the ~60-70 KiB per CU that 6 TB/s x ~2.5 us
// needs.  Measured (profiles/r02_dstream_microbench.txt): 2.2 - 2.8 TB/s on those two, 4.3 - 5.2 TB/s where NT = 3 - 6.
//
// Here the two streams are decoupled by wave role (MI355X guide: "x through LDS in full lines (glds), W any way you like"):
//   * 2 LOADER waves per workgroup move the fragment-major activations into an LDS ring with LDS-DMA (global_load_lds_dwordx4,
//     1 KiB per wave instruction, issued from inline asm so that hipcc neither drains them at the next LDS access nor counts them):
//     up to 8 stages in flight per loader, published to the consumer through a per-wave progress counter in LDS after a COUNTED
//     s_waitcnt vmcnt;
//   * 6 COMPUTE waves interleave the K stages (wave w: stages w, w + 6, ...), each with its private ring slots (one producer,
//     one consumer per slot: plain LDS counters, no workgroup barrier in the K loop) and a private register ring of weight
//     fragments 24 KiB deep (Dw = 24 / NT stages of NT tiles): 144 KiB of HBM requests in flight per CU for every shape.
//     Weight loads past the end of the K slice are masked by the buffer bounds check (never a branch around a load), so every
//     wait in the loop is a counted vmcnt.
// o_proj needs no split-K any more (no fp32 partial round trip, no reduce launch); down_proj keeps K slices (its 1.4 MB activation
// panel per CU is the L2-ingest bound) at the caller's choice.  Epilogue, deferred RMSNorm, fragment-major outputs: as in
// gemm_stream.hip (shared gemm_epilogue.h); the cross-wave reduction slab reuses the ring.
// Every spin is bounded: a loader / consumer that gives up sets nothing and moves on (wrong numbers, caught by the parity tests,
// instead of a hung GPU).
#include "vcla_common.h"
#include "gemm_epilogue.h"
#include <stdlib.h>

#define D2_NC 6      // compute waves
#define D2_NL 2      // loader waves (loader p feeds the compute waves of parity p)
#define D2_WAVES (D2_NC + D2_NL)
#define D2_ROUND 8   // epilogue units per LDS round: one per wave, loaders included
#define D2_SPIN_LIMIT (1 << 21)

typedef __attribute__((address_space(3))) unsigned char* d2_lds_ptr_t;
typedef volatile __attribute__((address_space(3))) unsigned* d2_ctr_ptr_t;
typedef const __attribute__((address_space(3))) bf16x8_t* d2_frag_ptr_t;

constexpr int d2_wdepth(int NT) { return NT >= 6 ? 4 : (NT >= 4 ? 6 : (NT == 3 ? 8 : (NT == 2 ? 12 : 16))); }
// ring slots per compute wave: 6 waves x DA x MT KiB <= 144 KiB (the reduction slab, 48 KiB or 96 KiB for SwiGLU, aliases it)
// (first cut: 5 slots / 8 stages in flight at MT = 4 = 64 KiB of activations in flight per CU -- slower than gemm_stream.hip on
// every shape: an L2 "hit" is ~2 us away under the weight stream, so 64 KiB in flight is ~32 GB/s per CU, and the activations,
// not the weights, set the pace.  Now: the largest ring LDS holds and as many DMA instructions in flight as vmcnt counts.)
constexpr int d2_adepth(int MT) { return MT >= 4 ? 6 : (MT == 3 ? 8 : (MT == 2 ? 12 : 16)); }
constexpr int d2_inflight(int MT) { return MT >= 4 ? 15 : (MT == 3 ? 20 : (MT == 2 ? 30 : 40)); }   // loader: stages in flight, MT x this <= 60 DMA instructions (vmcnt counts to 63)

struct D2Ctx {
    __amdgpu_buffer_rsrc_t rW;
    const unsigned char* A_frag;
    unsigned voff;            // lane * 16
    unsigned w_tile_bytes;    // one 16-row weight tile over the full K
    unsigned w_bytes;         // size of the weight buffer (bounds check = the load mask)
    int wave, lane;
    int s_beg, s_end;         // K stages [s_beg, s_end) of this workgroup's slice
    int T;                    // stages per compute wave (the same for all six: ceil)
    int ks, splitk;
    float* partial;
    int M, N;
    unsigned ring_lds;        // LDS byte address of the ring
    d2_lds_ptr_t ring;
    d2_ctr_ptr_t prod;        // [D2_NC] stages delivered to compute wave w (monotonic over the chunks of the launch)
    d2_ctr_ptr_t cons;        // [D2_NC] stages compute wave w has finished reading
};

__device__ __forceinline__ bool d2_wait_ge(d2_ctr_ptr_t p, unsigned want) {
    for (int spin = 0; spin < D2_SPIN_LIMIT; ++spin) {
        if (*p >= want) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// one LDS-DMA piece: 64 lanes x 16 B from `gsrc` (per lane) to LDS address `lds_dst` (wave-uniform) + lane * 16
__device__ __forceinline__ void d2_dma16(const unsigned char* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void d2_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- loader wave lw (0 / 1): stage sequence n = 0 .. 3T - 1 -> (t = n / 3, compute wave cw = lw + 2 (n % 3)); stage (cw, t) is K stage
// s_beg + cw + 6 t (clamped into the slice: the dummy stages of the last t are fetched and ignored) and lands in ring slot
// (cw, (tg0 + t) % DA).  Issue runs up to P stages ahead of publication, publication as far ahead of consumption as the ring allows.
template <int MT>
__device__ __forceinline__ void d2_loader(const D2Ctx& c, int lw, unsigned tg0) {
    constexpr int DA = d2_adepth(MT), P = d2_inflight(MT);
    const int NS = 3 * c.T;
    const unsigned char* abase = c.A_frag + c.voff;
    auto publish = [&](int m) {
        const int t = m / 3, cw = lw + 2 * (m - 3 * t);
        c.prod[cw] = tg0 + (unsigned)t + 1u;
    };
    for (int n = 0; n < NS; ++n) {
        const int t = n / 3, cw = lw + 2 * (n - 3 * t);
        const unsigned tg = tg0 + (unsigned)t;
        if (tg >= (unsigned)DA) d2_wait_ge(c.cons + cw, tg - DA + 1u);      // the slot's previous tenant has been read
        asm volatile("" ::: "memory");
        int ks = c.s_beg + cw + D2_NC * t;
        ks = ks < c.s_end ? ks : c.s_end - 1;
        const unsigned char* src = abase + (size_t)ks * (MT * 1024);
        const unsigned dst = __builtin_amdgcn_readfirstlane(c.ring_lds + (unsigned)((cw * DA + (int)(tg % DA)) * MT) * 1024u);
#pragma unroll
        for (int i = 0; i < MT; ++i) d2_dma16(src + i * 1024, dst + i * 1024);
        if (n >= P) {
            d2_vmcnt<P * MT>();      // the P stages issued after stage n - P may still be in flight; stage n - P has landed
            publish(n - P);
        }
    }
    d2_vmcnt<0>();
    for (int m = NS > P ? NS - P : 0; m < NS; ++m) publish(m);
}

// deferred RMSNorm, consumer side (see gemm_stream.hip): rstd of every activation row from the producer's partial sums of squares
__device__ __forceinline__ void d2_row_rstd(const vcla_gemm_args& a, int wave, int lane, float* rstd_s) {
    const int parts = a.a_row_ssq_parts;
    const int r = wave * 8 + (lane >> 3), seg = lane & 7;
    float q = 0.f;
    if (r < a.M) {
        const float* src = a.a_row_ssq + (int64_t)r * parts;
        for (int p = seg; p < parts; p += 8) q += src[p];
    }
    q += __shfl_xor(q, 1, 64);
    q += __shfl_xor(q, 2, 64);
    q += __shfl_xor(q, 4, 64);
    if (seg == 0 && r < 64) rstd_s[r] = r < a.M ? rsqrtf(q / (float)a.K + a.a_norm_eps) : 0.f;
}

// One chunk of NT weight tiles [c0, c0 + NT) x all rows over the workgroup's K slice, then the cross-wave reduction + epilogue.
template <int EPI, typename OutT, int MT, int NT>
__device__ __forceinline__ void d2_chunk(const D2Ctx& c, int c0, unsigned tg0, f32x4_t* slab, float* rstd_s, bool first) {
    constexpr int TPU = EPI == VCLA_EPI_SWIGLU ? 2 : 1;
    constexpr int DW = d2_wdepth(NT), DA = d2_adepth(MT);
    static_assert(NT % TPU == 0, "SwiGLU chunks hold whole gate/up pairs");
    f32x4_t acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (c.wave >= D2_NC) {
        // A wait hipcc can SEE (the builtin, not asm), at the head of the loader's code: in the structurized control-flow graph the
        // compute branch flows into this one, and its masked, never-consumed weight loads stay "pending" in hipcc's scoreboard -- the
        // first reuse of one of their registers inside the loader loop then gets a compiler-inserted s_waitcnt vmcnt(0) on EVERY
        // iteration, which at run time drains the loader's hand-counted DMA queue.  (A no-op when executed: a loader wave has
        // nothing outstanding here.)
        __builtin_amdgcn_s_waitcnt(0);
        d2_loader<MT>(c, c.wave - D2_NC, tg0);
    } else {
        const int cw = c.wave;
        u32x4_t rw[DW][NT];
        const unsigned w_chunk_off = (unsigned)c0 * c.w_tile_bytes;
        // stage t of this wave = K stage s_beg + cw + 6 t; past the slice the VECTOR offset is pushed beyond the buffer: the
        // bounds check returns zeros without touching memory
#define D2_LOAD(s_, t_)                                                                                              \
    {                                                                                                                \
        const int ks_ = c.s_beg + cw + D2_NC * (t_);                                                                 \
        const bool in_ = ks_ < c.s_end;                                                                              \
        const unsigned vo_ = in_ ? c.voff : 0x80000000u;      /* the bounds check covers voffset only, not soffset */ \
        const unsigned wo_ = in_ ? w_chunk_off + ((unsigned)ks_ << 10) : 0u;                                          \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                               \
            rw[s_][j] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(c.rW, vo_, wo_ + j * c.w_tile_bytes, 2 /* nt */)); \
        /* pin the issue order: left alone, hipcc issues the prologue's stages in REVERSE (stage 0 last), and the loop header then */ \
        /* merges "stage 0 is the youngest load" with the back edge's counted wait into s_waitcnt vmcnt(0) on every trip */          \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    }
#pragma unroll
        for (int s = 0; s < DW; ++s) D2_LOAD(s, s)
        for (int t0 = 0; t0 < c.T; t0 += DW) {
#pragma unroll
            for (int s = 0; s < DW; ++s) {
                const int t = t0 + s;
                if (t < c.T) {                                    // wave-uniform; no VMEM inside
                    const unsigned tg = tg0 + (unsigned)t;
                    d2_wait_ge(c.prod + cw, tg + 1u);
                    asm volatile("" ::: "memory");
                    const d2_lds_ptr_t slot = c.ring + (unsigned)((cw * DA + (int)(tg % DA)) * MT) * 1024u + c.voff;
                    bf16x8_t af[MT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) af[i] = *(d2_frag_ptr_t)(slot + i * 1024);
                    asm volatile("" ::: "memory");
                    c.cons[cw] = tg + 1u;                         // LDS executes a wave's operations in order: the reads above are done
                    if (c.s_beg + cw + D2_NC * t < c.s_end) {
#pragma unroll
                        for (int j = 0; j < NT; ++j)
#pragma unroll
                            for (int i = 0; i < MT; ++i)
                                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, rw[s][j]), af[i], acc[j][i], 0, 0, 0);
                    }
                }
                D2_LOAD(s, t + DW)
            }
        }
#undef D2_LOAD
    }

    // ---- ring -> reduction slab: every DMA has landed (the loaders drained) and every slot has been read
    __syncthreads();
    typedef const __attribute__((address_space(4))) vcla_gemm_args* kernarg_p;   // the argument block is the first kernel argument
    kernarg_p ap_ = (kernarg_p)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ap_));
    vcla_gemm_args a;
    __builtin_memcpy(&a, (const void*)ap_, sizeof(a));
    if (first && a.a_row_ssq) d2_row_rstd(a, c.wave, c.lane, rstd_s);
    constexpr int NU = (NT / TPU) * MT;
#pragma unroll
    for (int r0 = 0; r0 < NU; r0 += D2_ROUND) {
        if (c.wave < D2_NC) {
#pragma unroll
            for (int uu = 0; uu < D2_ROUND; ++uu) {
                if (r0 + uu < NU) {   // compile-time after unrolling: the accumulator indices below are literals
                    const int u = r0 + uu;
#pragma unroll
                    for (int tt = 0; tt < TPU; ++tt) slab[((c.wave * D2_ROUND + uu) * TPU + tt) * 64 + c.lane] = acc[(u / MT) * TPU + tt][u % MT];
                }
            }
        }
        __syncthreads();
        const int u = r0 + c.wave;
        if (u < NU) {
            f32x4_t sum[1][TPU];
#pragma unroll
            for (int tt = 0; tt < TPU; ++tt) sum[0][tt] = slab[((0 * D2_ROUND + c.wave) * TPU + tt) * 64 + c.lane];
#pragma unroll
            for (int w2 = 1; w2 < D2_NC; ++w2)
#pragma unroll
                for (int tt = 0; tt < TPU; ++tt) {
                    const f32x4_t p = slab[((w2 * D2_ROUND + c.wave) * TPU + tt) * 64 + c.lane];
                    sum[0][tt][0] += p[0]; sum[0][tt][1] += p[1]; sum[0][tt][2] += p[2]; sum[0][tt][3] += p[3];
                }
            const int jj = u / MT, i = u - jj * MT;
            if (c.splitk > 1) {
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
                gemm_epilogue<EPI, OutT, 1, TPU>(a, sum, i * 16, (c0 + jj * TPU) * 16, c.lane);
            }
        }
        __syncthreads();
    }
}

template <int EPI, typename OutT, int MT>
__global__ __launch_bounds__(D2_WAVES * 64) void gemm_dstream2_kernel(vcla_gemm_args a, int units_total) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char d2_smem[];
    constexpr int TPU = EPI == VCLA_EPI_SWIGLU ? 2 : 1;
    constexpr int NTW = EPI == VCLA_EPI_SWIGLU ? 6 : 4;
    constexpr int DA = d2_adepth(MT);
    constexpr size_t RING = (size_t)D2_NC * DA * MT * 1024, SLAB = (size_t)D2_NC * D2_ROUND * TPU * 1024;
    constexpr size_t BODY = RING > SLAB ? RING : SLAB;
    f32x4_t* slab = reinterpret_cast<f32x4_t*>(d2_smem);                   // [src wave][unit in round][tile of unit][lane], aliases the ring
    float* rstd_s = reinterpret_cast<float*>(d2_smem + BODY);              // [64]
    const d2_ctr_ptr_t ctr = (d2_ctr_ptr_t)((d2_lds_ptr_t)d2_smem + BODY + 256);   // prod[6], cons[6] (padded to 8 each)
    D2Ctx c;
    c.lane = threadIdx.x & 63;
    c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 16) ctr[threadIdx.x] = 0u;
    const int S = a.ds_splitk > 1 ? a.ds_splitk : 1;
    const int G = gridDim.x / S, g = blockIdx.x / S;           // G groups of S workgroups: one K slice each, the same tiles
    c.splitk = S; c.ks = blockIdx.x - g * S; c.partial = (float*)a.splitk_ws; c.M = a.M; c.N = a.N;
    const int t_beg = (int)((int64_t)g * units_total / G) * TPU, t_end = (int)((int64_t)(g + 1) * units_total / G) * TPU;
    const int KST = a.K / 32;                                  // stages along K
    c.s_beg = (int)((int64_t)c.ks * KST / S);
    c.s_end = (int)((int64_t)(c.ks + 1) * KST / S);
    c.T = (c.s_end - c.s_beg + D2_NC - 1) / D2_NC;
    const int n_pad = (a.N + 127) / 128 * 128;
    c.w_tile_bytes = (unsigned)a.K * 32u;
    c.w_bytes = (unsigned)((int64_t)(n_pad / 16) * c.w_tile_bytes);
    c.rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.W_frag), 0, (int)c.w_bytes, 0x00020000);
    c.A_frag = (const unsigned char*)a.A_frag;
    c.voff = c.lane * 16;
    c.ring = (d2_lds_ptr_t)d2_smem;
    c.ring_lds = (unsigned)(uintptr_t)c.ring;
    c.prod = ctr; c.cons = ctr + 8;
    __syncthreads();                                           // counters are zero before anyone polls them

    unsigned tg0 = 0;                                          // stages per compute wave issued by earlier chunks
    for (int c0 = t_beg; c0 < t_end; c0 += NTW) {
        const int nt = (t_end - c0) < NTW ? (t_end - c0) : NTW;     // tiles of this chunk (workgroup-uniform)
        const bool first = c0 == t_beg;
        if constexpr (TPU == 2) {
            if (nt == 6) d2_chunk<EPI, OutT, MT, 6>(c, c0, tg0, slab, rstd_s, first);
            else if (nt == 4) d2_chunk<EPI, OutT, MT, 4>(c, c0, tg0, slab, rstd_s, first);
            else d2_chunk<EPI, OutT, MT, 2>(c, c0, tg0, slab, rstd_s, first);
        } else {
            if (nt == 4) d2_chunk<EPI, OutT, MT, 4>(c, c0, tg0, slab, rstd_s, first);
            else if (nt == 3) d2_chunk<EPI, OutT, MT, 3>(c, c0, tg0, slab, rstd_s, first);
            else if (nt == 2) d2_chunk<EPI, OutT, MT, 2>(c, c0, tg0, slab, rstd_s, first);
            else d2_chunk<EPI, OutT, MT, 1>(c, c0, tg0, slab, rstd_s, first);
        }
        tg0 += (unsigned)c.T;
    }
}

int vcla_ds_reduce_launch(const vcla_gemm_args* a, hipStream_t s);                           // gemm_stream.hip

template <int EPI, typename OutT, int MT>
static int d2_launch(const vcla_gemm_args* a, int units, int grid, hipStream_t s) {
    constexpr int TPU = EPI == VCLA_EPI_SWIGLU ? 2 : 1;
    constexpr int DA = d2_adepth(MT);
    constexpr size_t RING = (size_t)D2_NC * DA * MT * 1024, SLAB = (size_t)D2_NC * D2_ROUND * TPU * 1024;
    const size_t lds = (RING > SLAB ? RING : SLAB) + 256 + 64;
    auto kern = gemm_dstream2_kernel<EPI, OutT, MT>;
    static bool attr_set[VCLA_MAX_DEVICES] = {};   // per instantiation and device
    { const int rc_ = vcla_raise_dyn_lds((const void*)kern, lds, attr_set); if (rc_) return rc_; }
    kern<<<grid, D2_WAVES * 64, lds, s>>>(*a, units);
    VCLA_CHECK_LAUNCH("gemm_dstream2_kernel");
    if (EPI == VCLA_EPI_NONE && a->ds_splitk > 1) return vcla_ds_reduce_launch(a, s);
    return VCLA_OK;
}

template <int EPI, typename OutT>
static int d2_pick_mt(const vcla_gemm_args* a, int units, int grid, hipStream_t s) {
    const int mt = (a->M + 15) / 16;
    if (mt <= 1) return d2_launch<EPI, OutT, 1>(a, units, grid, s);
    if (mt == 2) return d2_launch<EPI, OutT, 2>(a, units, grid, s);
    if (mt == 3) return d2_launch<EPI, OutT, 3>(a, units, grid, s);
    return d2_launch<EPI, OutT, 4>(a, units, grid, s);
}

// called by vcla_gemm_dstream_launch (gemm_stream.hip) for bf16 weights; arguments were validated in vcla_gemm
int vcla_gemm_dstream2_launch(const vcla_gemm_args* a, int units, int grid, hipStream_t s) {
    if (a->epilogue == VCLA_EPI_SWIGLU) return d2_pick_mt<VCLA_EPI_SWIGLU, bf16_t>(a, units, grid, s);
    if (a->out_f32) return d2_pick_mt<VCLA_EPI_NONE, float>(a, units, grid, s);
    return d2_pick_mt<VCLA_EPI_NONE, bf16_t>(a, units, grid, s);
}
