// decode_engine.hip -- ONE persistent launch per B = 1 decode step (bf16, LLaMA-7B geometry): the 32 decoder layers and the lm_head.
//
// Replaces the 161-launch step of engine.hip's `llama_layer` GEMV branch (per layer: qkv GEMV, fused attention, o_proj GEMV, gate/up GEMV,
// down GEMV; spec: hf llama/modeling_llama.py:284-325 under the per-token loop of generation/utils.py that
// /root/reference/models/visualcla/modeling_visualcla.py:382-391 enters).  Why: at B = 1 every launch pays its own ramp (x staging, the first
// HBM round trip, the drain) and the weight stream idles during the attention -- 2.66 ms per token for 13.4 GB = 63 % of the HBM peak although
// the long GEMVs alone reach 74 %.  MI355X_MICROARCH.md ("engine-vs-launches", "prefetch-credit", "ldsdma-fill") measures the structure used here:
//   * 256 workgroups, one per CU, 4 waves each: wave 0 is a LOADER, waves 1 - 3 are CONSUMERS;
//   * the loader streams its CU's share of ALL weights of the step -- its 16-KiB slots of the engine twin ("llama.engine.w", built by
//     weights.add_engine_stream: [slot][CU][16 KiB] in consumption order; SLOT-major, so that the 256 loaders together sweep one moving window of HBM
//     instead of 256 private regions: profiles/r06_engine_placement.txt) -- with non-temporal `global_load_lds_dwordx4` into a 9 x 16 KiB LDS
//     ring.  It knows nothing about operators: it runs ahead across every dependency edge until the ring is full (the prefetch credit);
//   * a consumer wave takes a landed slot, multiplies it against the operator's input vector (bf16 in registers / LDS, `v_dot2c_f32_bf16`),
//     reduces over the wave and publishes the outputs;
//   * operator outputs travel between CUs as 8-byte {epoch tag, two bf16} granules: ONE relaxed agent-scope (sc1, write-through) store per
//     granule, swept by ONE consumer wave per CU with relaxed agent-scope loads until every tag matches -- the data is the flag, no fences;
//   * the loader never keeps more than two fills (32 KiB) in flight: a CU's mailbox polls queue behind its own DMA;
//   * the attention of head h runs on the consumers of CU 8 h + h % 8 (one per head, spread over the XCDs) while every loader keeps
//     prefetching wo / wgu; from args.split_min cached keys on, the 8 CUs of the head's group share the walk over its cache (eg_attention).
// Row ownership (so that no operator needs more than the all-gather of its input vector): CU c owns q / k / v rows h*128 + s*16 .. + 15 of
// head h = c / 8, s = c % 8; rows 16 c .. + 15 of o_proj and down_proj (so the residual of its outputs is what it produced itself two
// operators earlier); SwiGLU units upc * c .. ; lm_head rows 2 s_lm c ...  down_proj's K dimension is stored in granule order (44 slots per
// producer CU: 43 activations + one zero), so the gathered activation mailbox IS its input vector.
// Values: bf16 between operators exactly where the launch path rounds (qkv, attention output, both residual sums, SwiGLU output); the
// RMSNorm output is rounded to bf16 once, as x * rstd * gamma (HF rounds it twice; the launch path keeps it in fp32): inside the bf16 bounds
// the 7B parity tests state.  Every spin is bounded; a timeout records a site code in the workspace and lets every wave run out.
#include "decode_engine.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* eg_lds_ptr_t;
typedef __attribute__((address_space(1))) unsigned long long eg_gu64;
typedef __attribute__((ext_vector_type(2))) __bf16 eg_bf16x2_t;

#define EG_RING_BYTES (EG_NRING * EG_SLOT)
#define EG_XIN_BYTES 8192                  // 4096 bf16: the normalised input vector of qkv / gate-up (o_proj and down_proj read theirs from the mailbox)
#define EG_MISC_OFF (EG_RING_BYTES + EG_XIN_BYTES)
#define EG_LDS_BYTES (EG_MISC_OFF + 4096)
#define EG_SPIN_LDS (1u << 22)             // ~0.4 s of LDS polls
#define EG_SPIN_GLB (1u << 17)             // ~0.2 s of mailbox sweeps

struct EgMisc {                            // LDS words shared by the four waves (single writer each)
    unsigned filled;                       // loader: number of slots that have landed (monotonic)
    unsigned freed[EG_NRING];              // consumers: ring position p holds (g + 1) of the last slot released there
    unsigned xin_ready;                    // leader: sequence number of the operator whose input vector is staged
    unsigned cons_done[3];                 // consumer w: sequence number of the last operator it finished
    unsigned attn_ready, attn_done[3];
    unsigned unused_;                      // (was: a flag that made the loader thin itself during sweeps; it now never keeps more than two fills in flight)
    unsigned fail;
    unsigned pad_[12];                     // the 32 words above are zeroed at kernel start
    float resid0[16], resid1[16];          // the CU's 16 rows of the layer input x / of x + o_proj(...) (bf16 values)
    float dpart[3][16];                    // down_proj: per-consumer partial sums of the CU's 16 rows
    float amax_v[3];                       // lm_head: per-consumer maximum of its logits and the lowest row that holds it
    int amax_i[3];
    unsigned qpk[64];                      // attention: roped q, packed bf16 pairs
    float knew[128], vnew[128];
    float part[3][132];                    // per-consumer (o[128], m, l)
};
static_assert(sizeof(EgMisc) <= 4096, "misc region");

__device__ __forceinline__ unsigned eg_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void eg_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void eg_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); }
__device__ __forceinline__ void eg_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); }

__device__ __forceinline__ void eg_dma16_nt(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void eg_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// a failed wait: remember where (first failure wins), tell the workgroup
__device__ __forceinline__ void eg_fail(EgMisc* m, unsigned* state, unsigned code) {
    eg_st(&m->fail, code);
    if ((threadIdx.x & 63) == 0) atomicCAS(state + 1, 0u, code | (blockIdx.x << 16));
}
// wait until *p >= need (LDS word written by another wave of this workgroup)
__device__ __forceinline__ bool eg_wait_ge(const unsigned* p, unsigned need, EgMisc* m, unsigned* state, unsigned code) {
    for (unsigned it = 0;; ++it) {
        if (eg_ld(p) >= need) return true;
        if ((it & 63) == 63 && eg_ld(&m->fail)) return false;
        if (it > EG_SPIN_LDS) { eg_fail(m, state, code); return false; }
        __builtin_amdgcn_s_sleep(1);
    }
}

template <int CTRL> __device__ __forceinline__ float eg_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row; every lane of the row ends up with the total
__device__ __forceinline__ float eg_row_sum(float v) {
    v += eg_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
    v += eg_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
    v += eg_dpp<0x141>(v);     // row_half_mirror
    v += eg_dpp<0x140>(v);     // row_mirror
    return v;
}
// sum over the wave; every lane ends up with the total (v_permlane16_swap / v_permlane32_swap: no SGPR round trip, no LDS)
__device__ __forceinline__ float eg_wave_sum(float v) {
    v = eg_row_sum(v);
    const unsigned b = __builtin_bit_cast(unsigned, v);
    const auto s = __builtin_amdgcn_permlane16_swap(b, b, false, false);       // rows (0, 0, 2, 2) | (1, 1, 3, 3)
    v = __builtin_bit_cast(float, (unsigned)s[0]) + __builtin_bit_cast(float, (unsigned)s[1]);
    const unsigned c = __builtin_bit_cast(unsigned, v);
    const auto t = __builtin_amdgcn_permlane32_swap(c, c, false, false);       // halves (lo, lo) | (hi, hi)
    return __builtin_bit_cast(float, (unsigned)t[0]) + __builtin_bit_cast(float, (unsigned)t[1]);
}
// 8 bf16 x 8 bf16 on v_dot2c_f32_bf16.  (Pairs are taken with shufflevector from the whole vector: __builtin_bit_cast of an ext-vector ELEMENT is
// miscompiled by this clang -- every element collapses to .x; see attention_decode.hip.)
#define EG_PAIR(v, i) __builtin_shufflevector(v, v, 2 * (i), 2 * (i) + 1)
// four independent accumulators (one per dword of the 16-byte piece)
__device__ __forceinline__ void eg_dot8(const u32x4_t& w, const u32x4_t& x, float (&acc)[4]) {
    const bf16x8_t a = __builtin_bit_cast(bf16x8_t, w), b = __builtin_bit_cast(bf16x8_t, x);
    acc[0] = __builtin_amdgcn_fdot2_f32_bf16(EG_PAIR(a, 0), EG_PAIR(b, 0), acc[0], false);
    acc[1] = __builtin_amdgcn_fdot2_f32_bf16(EG_PAIR(a, 1), EG_PAIR(b, 1), acc[1], false);
    acc[2] = __builtin_amdgcn_fdot2_f32_bf16(EG_PAIR(a, 2), EG_PAIR(b, 2), acc[2], false);
    acc[3] = __builtin_amdgcn_fdot2_f32_bf16(EG_PAIR(a, 3), EG_PAIR(b, 3), acc[3], false);
}
__device__ __forceinline__ float eg_dot8s(const u32x4_t& w, const u32x4_t& x, float acc) {
    const bf16x8_t a = __builtin_bit_cast(bf16x8_t, w), b = __builtin_bit_cast(bf16x8_t, x);
    acc = __builtin_amdgcn_fdot2_f32_bf16(EG_PAIR(a, 0), EG_PAIR(b, 0), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(EG_PAIR(a, 1), EG_PAIR(b, 1), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(EG_PAIR(a, 2), EG_PAIR(b, 2), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(EG_PAIR(a, 3), EG_PAIR(b, 3), acc, false);
    return acc;
}
__device__ __forceinline__ float eg_lo(unsigned d) { return __uint_as_float(d << 16); }
__device__ __forceinline__ float eg_hi(unsigned d) { return __uint_as_float(d & 0xffff0000u); }

__device__ __forceinline__ void eg_publish(unsigned long long* mb, int idx, unsigned epoch, unsigned data) {
    __hip_atomic_store((eg_gu64*)(mb + idx), ((unsigned long long)epoch << 32) | data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long eg_peek(const unsigned long long* mb, int idx) {
    return __hip_atomic_load((const eg_gu64*)(mb + idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// everything a consumer wave carries around
struct EgCtx {
    const vcla_engine_args* a;
    unsigned char* ring;
    unsigned char* xin;
    EgMisc* m;
    unsigned* state;
    int w, lane, cu;
    unsigned eb;              // epoch base of this launch
    int pos;                  // position of the token being decoded = number of cached keys
};
__device__ __forceinline__ unsigned eg_epoch(const EgCtx& c, int layer, int edge) { return c.eb + (unsigned)(layer * 8 + edge); }
__device__ __forceinline__ unsigned long long* eg_mb(const EgCtx& c, int layer, int off) {
    return c.a->mbox + (size_t)(layer & 1) * EG_MB_PER_PARITY + off;
}
// The layer loop is one huge function after inlining, and LICM hoists every lane- / wave- / CU-derived address of every phase in front of it, where
// they stay live across all phases and crowd out the slot registers.  Re-defining the three roots at the start of a phase makes everything derived
// from them local to the phase.  (Through a VGPR + readfirstlane: an "s" asm operand needs a value hipcc can PROVE uniform.)
__device__ __forceinline__ void eg_fresh(EgCtx& c) {
    asm volatile("" : "+v"(c.lane));
    int w = c.w, cu = c.cu;
    asm volatile("" : "+v"(w), "+v"(cu));
    c.w = __builtin_amdgcn_readfirstlane(w);
    c.cu = __builtin_amdgcn_readfirstlane(cu);
}
// debug stamps (timeline != NULL only): lane 0 of the leader wave of every CU
__device__ __forceinline__ void eg_stamp(const EgCtx& c, int layer, int k) {
    if (c.a->timeline && c.lane == 0) c.a->timeline[(size_t)c.cu * EG_TL_STRIDE + layer * 16 + k] = wall_clock64();
}

// ---- sweep NCH x 1024 granules starting at mb (ONE wave): v[k] = data of granule lane + 64 k
template <int NCH>
__device__ __forceinline__ bool eg_sweep(EgCtx& c, const unsigned long long* mb, unsigned epoch, unsigned (&v)[NCH * 16], int n_gran, unsigned code) {
    // all NCH x 16 loads of the lane are in flight together (a pass is latency-bound: ~1 us for 16 loads, not much more for 32)
    for (unsigned it = 0;; ++it) {
        bool good = true;
#pragma unroll
        for (int k = 0; k < NCH * 16; ++k) {
            const int idx = c.lane + 64 * k;
            const int idc = idx < n_gran ? idx : n_gran - 1;           // past the end: re-read the last granule (tag checked like any other)
            const unsigned long long x = eg_peek(mb, idc);
            v[k] = (unsigned)x;
            good = good && (unsigned)(x >> 32) == epoch;
        }
        if (__all(good)) break;
        if ((it & 15) == 15 && eg_ld(&c.m->fail)) return false;
        if (it > EG_SPIN_GLB) { eg_fail(c.m, c.state, code); return false; }
        __builtin_amdgcn_s_sleep(8);
    }
    return true;
}

// ---- leader: stage RMSNorm(x) * gamma (bf16) as the next operator's input.  x = the 2048-granule mailbox `mb`, or x_in (plain memory, layer 0)
__device__ __forceinline__ bool eg_stage_norm(EgCtx& c, const unsigned long long* mb, unsigned epoch, const float* gamma, bool from_mem) {
    float2 gm[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) gm[k] = *reinterpret_cast<const float2*>(gamma + 2 * (c.lane + 64 * k));
    unsigned v[32];
    if (from_mem) {
        const unsigned* xi = reinterpret_cast<const unsigned*>(c.a->x_in);
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = xi[c.lane + 64 * k];
    } else {
        const bool good = eg_sweep<2>(c, mb, epoch, v, 2048, 0x11);
        if (!good) return false;
    }
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) { const float x0 = eg_lo(v[k]), x1 = eg_hi(v[k]); ss += x0 * x0 + x1 * x1; }
    ss = eg_wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)EG_D + c.a->eps);
    unsigned* xo = reinterpret_cast<unsigned*>(c.xin);
#pragma unroll
    for (int k = 0; k < 32; ++k) xo[c.lane + 64 * k] = pack_bf2(eg_lo(v[k]) * rstd * gm[k].x, eg_hi(v[k]) * rstd * gm[k].y);
    if (from_mem && c.lane < 16) c.m->resid0[c.lane] = bf2f(c.a->x_in[16 * c.cu + c.lane]);
    return true;
}
// ---- a landed slot: LDS -> registers (16 x 16 bytes per lane), ring position released at once
__device__ __forceinline__ bool eg_fetch(EgCtx& c, int g, u32x4_t (&wv)[16], unsigned code) {
    if (!eg_wait_ge(&c.m->filled, (unsigned)g + 1, c.m, c.state, code)) return false;
    eg_acquire();
    const int rp = g % EG_NRING;
    const unsigned char* base = c.ring + rp * EG_SLOT + c.lane * 16;
#pragma unroll
    for (int p = 0; p < 16; ++p) wv[p] = *reinterpret_cast<const u32x4_t*>(base + p * 1024);
    eg_release();                                  // (waits for the reads above)
    eg_st(&c.m->freed[rp], (unsigned)g + 1);
    return true;
}
// ---- two rows of 4096 weights (one slot of a row-major operator) against the x registers -> two wave-uniform sums
__device__ __forceinline__ void eg_rows2(const u32x4_t (&wv)[16], const u32x4_t (&xr)[8], float& t0, float& t1) {
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 8; ++p) { eg_dot8(wv[p], xr[p], a0); eg_dot8(wv[8 + p], xr[p], a1); }
    t0 = eg_wave_sum((a0[0] + a0[1]) + (a0[2] + a0[3]));
    t1 = eg_wave_sum((a1[0] + a1[1]) + (a1[2] + a1[3]));
}

enum { EG_OP_QKV = 0, EG_OP_O = 1, EG_OP_GU = 2, EG_OP_LM = 3, EG_OP_DN = 4 };

// Slots consumers 1 / 2 pull into REGISTERS before the operator's input vector exists.  The ring alone is 8 slots = ~4.7 us of stream; the edges of a
// layer last 4 - 12 us (attention + its all-gather: 11), and whenever the ring is full the weight stream stops.  A consumer wave owns a SIMD (512
// registers), so consumers 1 and 2 take the FIRST 2 x NP slots of every operator (alternating) into registers as they land -- the leader sweeps the
// mailbox then, and a slot it owned would block the ring behind it -- and the three share the rest round-robin.
#ifndef EG_PRE
#define EG_PRE 1            // qkv, down_proj, lm_head (more than 1 / 2 / 1 sends hipcc into scratch: ~330 registers of phase state stay live whatever the count)
#endif
#ifndef EG_PRE_GU
#define EG_PRE_GU 2         // gate/up: whole pairs (the two activations of one granule)
#endif
#ifndef EG_PRE_O
#define EG_PRE_O 1          // o_proj: live ACROSS the attention code
#endif
#define EG_PRE_MAX 4
template <int OP> struct EgUnit {          // slots per unit (gate/up: the pair of one granule), preloaded slots / units per wave, first round-robin unit
    static constexpr int US = OP == EG_OP_GU ? 2 : 1, NP = OP == EG_OP_O ? EG_PRE_O : OP == EG_OP_GU ? EG_PRE_GU : EG_PRE, PU = NP / US, F = 2 * PU;
    static_assert(NP % US == 0 && NP <= EG_PRE_MAX && NP >= 1, "preload counts");
};

// consumers 1 / 2: preload units (w - 1) + 2 i, i < PU.  Unconditional (the geometry guarantees every operator has at least 2 * EG_PRE_MAX slots): a
// conditionally defined register array becomes a loop-carried value of the layer loop and is kept live across every phase.
template <int OP, int NP>
__device__ __forceinline__ bool eg_preload(EgCtx& c, int g0, u32x4_t (&pre)[NP][16]) {
    using U = EgUnit<OP>;
    static_assert(NP == U::PU * U::US, "preload registers");
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        // K-major operators (no gather phase: the leader is a consumer like the others): slots w, w + 3, ...; row-major ones: consumers 1 / 2 alternate
        const int slot = (OP == EG_OP_O || OP == EG_OP_DN) ? c.w + 3 * i : ((c.w - 1) + 2 * (i / U::US)) * U::US + (i % U::US);
        if (!eg_fetch(c, g0 + slot, pre[i], 0x21)) return false;
    }
    return true;
}

// ---- a row-major operator (K = 4096) over nslots slots starting at global slot g0.  LEADER (consumer 0) has no preloaded slots (`pre` unused)
template <int OP, bool LEADER>
__device__ __forceinline__ bool eg_run_rows(EgCtx& c, int g0, int nslots, int layer, unsigned seq, const u32x4_t (*pre)[16]) {
    using U = EgUnit<OP>;
    const vcla_engine_args& a = *c.a;
    const int gpc = (a.g.upc + 1) >> 1;
    if (!eg_wait_ge(&c.m->xin_ready, seq, c.m, c.state, 0x22)) return false;
    eg_acquire();
    u32x4_t xr[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) xr[p] = *reinterpret_cast<const u32x4_t*>(c.xin + p * 1024 + c.lane * 16);
    float held = 0.f;                              // gate/up: the first activation of the pair being assembled
    float best = -INFINITY;                        // lm_head: running maximum of this wave's logits (rows increase: strictly greater keeps the lowest index)
    int best_i = 0x7fffffff;
    auto finish = [&](int j, float t0, float t1) {
        if constexpr (OP == EG_OP_GU) {
            const float v = act_silu(t0) * t1;
            const bool last_single = (j & 1) == 0 && j + 1 >= nslots;
            if ((j & 1) == 0 && !last_single) { held = v; return; }
            const unsigned d = (j & 1) ? pack_bf2(held, v) : pack_bf2(v, 0.f);
            if (c.lane == 0) eg_publish(eg_mb(c, layer, EG_MB_ACT), gpc * c.cu + (j >> 1), eg_epoch(c, layer, 4), d);
        } else if constexpr (OP == EG_OP_QKV) {
            const int part = j >> 3, jj = j & 7;
            const int row = part * EG_D + (c.cu >> 3) * EG_HD + (c.cu & 7) * 16 + 2 * jj;
            if (c.lane == 0) eg_publish(eg_mb(c, layer, EG_MB_QKV), row >> 1, eg_epoch(c, layer, 1), pack_bf2(t0, t1));
        } else {
            const int row = 2 * a.g.s_lm * c.cu + 2 * j;
            if (c.lane == 0) {
                if (row < a.g.vocab) a.logits[row] = t0;
                if (row + 1 < a.g.vocab) a.logits[row + 1] = t1;
            }
            if (row < a.g.vocab && t0 > best) { best = t0; best_i = row; }
            if (row + 1 < a.g.vocab && t1 > best) { best = t1; best_i = row + 1; }
        }
    };
    if constexpr (!LEADER) {
#pragma unroll
        for (int i = 0; i < U::NP; ++i) {
            const int slot = ((c.w - 1) + 2 * (i / U::US)) * U::US + (i % U::US);
            float t0, t1;
            eg_rows2(pre[i], xr, t0, t1);
            finish(slot, t0, t1);
        }
    }
    for (int u = U::F + c.w; u * U::US < nslots; u += 3) {
#pragma unroll
        for (int k = 0; k < U::US; ++k) {
            const int slot = u * U::US + k;
            if (slot < nslots) {
                u32x4_t wv[16];
                float t0, t1;
                if (!eg_fetch(c, g0 + slot, wv, 0x21)) return false;
                eg_rows2(wv, xr, t0, t1);
                finish(slot, t0, t1);
            }
        }
    }
    if constexpr (OP == EG_OP_LM) {
        if (c.lane == 0) { c.m->amax_v[c.w] = best; c.m->amax_i[c.w] = best_i; }
    }
    eg_release();
    eg_st(&c.m->cons_done[c.w], seq);
    return true;
}

// ---- the greedy tail (args.tail_ids_out != NULL): every CU's leader publishes the maximum of its share of the logits, CU 0 picks the winner and does the
// step's bookkeeping -- what vcla_argmax + post_select_kernel did in two more launches (same tie rule: the lowest index among equal maxima; NaN never wins)
__device__ __forceinline__ bool eg_greedy_tail(EgCtx& c, int L) {
    const vcla_engine_args& a = *c.a;
    EgMisc* m = c.m;
    float best = m->amax_v[0];
    int bi = m->amax_i[0];
#pragma unroll
    for (int w = 1; w < 3; ++w) {
        const float v = m->amax_v[w];
        const int i = m->amax_i[w];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
    unsigned long long* mb = eg_mb(c, L, EG_MB_QKV);            // (nobody uses this layer parity's q / k / v mailbox any more)
    const unsigned ep = eg_epoch(c, L, 5);
    if (c.lane == 0) {
        eg_publish(mb, c.cu, ep, __float_as_uint(best));
        eg_publish(mb, EG_NCU + c.cu, ep, (unsigned)bi);
    }
    if (c.cu != 0) return true;
    float v[4];
    int ix[4];
    for (unsigned it = 0;; ++it) {
        bool good = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long xv = eg_peek(mb, c.lane + 64 * k), xi = eg_peek(mb, EG_NCU + c.lane + 64 * k);
            v[k] = __uint_as_float((unsigned)xv);
            ix[k] = (int)(unsigned)xi;
            good = good && (unsigned)(xv >> 32) == ep && (unsigned)(xi >> 32) == ep;
        }
        if (__all(good)) break;
        if ((it & 15) == 15 && eg_ld(&m->fail)) return false;
        if (it > EG_SPIN_GLB) { eg_fail(m, c.state, 0x51); return false; }
        __builtin_amdgcn_s_sleep(4);
    }
    best = v[0]; bi = ix[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (v[k] > best || (v[k] == best && ix[k] < bi)) { best = v[k]; bi = ix[k]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    const long long id = bi == 0x7fffffff ? 0 : bi;
    const int step = c.pos - a.pos0;                            // = *pos_dev at the start of this launch
    if (c.lane == 0) {
        a.tail_cur[0] = id;
        a.tail_ids_out[step - a.tail_step_base] = id;
        *a.tail_pos = step + 1;
    }
    const long long idc = id < a.g.vocab ? id : 0;              // as post_select_kernel: stay in bounds
    const u32x4_t* src = reinterpret_cast<const u32x4_t*>(a.tail_embed + idc * EG_D);
    u32x4_t* dst = reinterpret_cast<u32x4_t*>(a.tail_x);
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[c.lane + 64 * k] = src[c.lane + 64 * k];
    return true;
}

// ---- 256 granules [256 j, 256 j + 256) of a mailbox = the 512 input values slot j of a K-major operator multiplies: lane l takes granules 4 l .. 4 l + 3,
// i.e. exactly the 8 bf16 its 16-byte weight pieces pair with.  Issue (4 loads per lane) and check are separate so that the next slot's slice is in
// flight while this one is used.
struct EgSlice { unsigned long long g[4]; };
__device__ __forceinline__ void eg_slice_issue(const EgCtx& c, const unsigned long long* mb, int j, EgSlice& s) {
#pragma unroll
    for (int q = 0; q < 4; ++q) s.g[q] = eg_peek(mb, 256 * j + 4 * c.lane + q);
}
__device__ __forceinline__ bool eg_slice_take(EgCtx& c, const unsigned long long* mb, int j, unsigned epoch, EgSlice& s, u32x4_t& xk, unsigned code) {
    for (unsigned it = 0;; ++it) {
        bool good = true;
#pragma unroll
        for (int q = 0; q < 4; ++q) good = good && (unsigned)(s.g[q] >> 32) == epoch;
        if (__all(good)) break;
        if ((it & 15) == 15 && eg_ld(&c.m->fail)) return false;
        if (it > EG_SPIN_GLB) { eg_fail(c.m, c.state, code); return false; }
        __builtin_amdgcn_s_sleep(2);
        eg_slice_issue(c, mb, j, s);
    }
    xk = u32x4_t{(unsigned)s.g[0], (unsigned)s.g[1], (unsigned)s.g[2], (unsigned)s.g[3]};
    return true;
}

// ---- a K-major operator (o_proj: OP = EG_OP_O, down_proj: EG_OP_DN): slot j = the CU's 16 output rows x 512 inputs.  No gather phase: every wave reads
// the input slice of its slot straight from the producers' mailbox `mb` (attention output / SwiGLU activations), so the operator starts as soon as the
// first granules are published.  The leader adds the three consumers' partial sums and the residual, rounds to bf16 and publishes the 16 outputs
// (o_proj: x1 = x + o_proj(...); down_proj: the next layer's x).
template <int OP, bool LEADER>
__device__ __forceinline__ bool eg_run_kmajor(EgCtx& c, int g0, int nslots, int layer, unsigned seq, const unsigned long long* mb, unsigned epoch,
                                              const u32x4_t (*pre)[16]) {
    constexpr int NP = EgUnit<OP>::NP;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // this wave's slots: w, w + 3, ... -- the first NP of them are already in registers (eg_preload)
    auto slot_of = [&](int t) -> int { return c.w + 3 * t; };
    // a mailbox read takes ~2 us under the weight stream and a wave gets a slot every ~1.8 us: three slices in flight
    EgSlice q0, q1, q2;
    if (slot_of(0) < nslots) eg_slice_issue(c, mb, slot_of(0), q0);
    if (slot_of(1) < nslots) eg_slice_issue(c, mb, slot_of(1), q1);
    if (slot_of(2) < nslots) eg_slice_issue(c, mb, slot_of(2), q2);
    int t = 0;
#pragma unroll
    for (int i = 0; i < NP; ++i, ++t) {
        u32x4_t xk;
        if (!eg_slice_take(c, mb, slot_of(t), epoch, q0, xk, 0x27)) return false;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = eg_dot8s(pre[i][r], xk, acc[r]);
        q0 = q1; q1 = q2;
        if (slot_of(t + 3) < nslots) eg_slice_issue(c, mb, slot_of(t + 3), q2);
    }
    for (; slot_of(t) < nslots; ++t) {
        u32x4_t wv[16];
        if (!eg_fetch(c, g0 + slot_of(t), wv, 0x24)) return false;
        u32x4_t xk;
        if (!eg_slice_take(c, mb, slot_of(t), epoch, q0, xk, 0x28)) return false;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = eg_dot8s(wv[r], xk, acc[r]);
        q0 = q1; q1 = q2;
        if (slot_of(t + 3) < nslots) eg_slice_issue(c, mb, slot_of(t + 3), q2);
    }
    float mine = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float t = eg_wave_sum(acc[i]); mine = c.lane == i ? t : mine; }
    if (c.lane < 16) c.m->dpart[c.w][c.lane] = mine;
    eg_release();
    eg_st(&c.m->cons_done[c.w], seq);
    if constexpr (LEADER) {
        if (!eg_wait_ge(&c.m->cons_done[1], seq, c.m, c.state, 0x25) || !eg_wait_ge(&c.m->cons_done[2], seq, c.m, c.state, 0x26)) return false;
        eg_acquire();
        const int l = c.lane & 15;
        const float res = OP == EG_OP_O ? c.m->resid0[l] : c.m->resid1[l];
        const float v = (c.m->dpart[0][l] + c.m->dpart[1][l]) + c.m->dpart[2][l] + res;
        const float vr = Act<bf16_t>::rnd(v);
        const float nb = __shfl_down(vr, 1, 64);
        if (c.lane < 16) { if (OP == EG_OP_O) c.m->resid1[c.lane] = vr; else c.m->resid0[c.lane] = vr; }
        if (c.lane < 16 && (c.lane & 1) == 0) {
            if (OP == EG_OP_O) eg_publish(eg_mb(c, layer, EG_MB_X1), 8 * c.cu + (c.lane >> 1), eg_epoch(c, layer, 3), pack_bf2(vr, nb));
            else eg_publish(eg_mb(c, layer + 1, EG_MB_X), 8 * c.cu + (c.lane >> 1), eg_epoch(c, layer + 1, 0), pack_bf2(vr, nb));
        }
    }
    return true;
}

// ---- attention of one head on this CU's three consumers (RoPE + cache append + single-pass online softmax over the cache; the arithmetic of
// attn_decode_flash_kernel<128, NW, MASK>, attention_decode.hip, with NW = 3).  U keys per lane group and batch, two batches in flight.
// (8 keys per batch = 192 keys requested before q exists; the MASK instantiation is 8 registers short for that with o_proj slots held across this code: 6)
// SPLIT (long contexts, args.split_min): the head's cached keys are dealt over all 8 CUs of its group (CU 8 h + s takes the key streams 12 s .. 12 s + 11 of
// 96); the 7 helpers (OWNER = false) publish their unnormalised (o[128], m, l) as fp32 granules and the owner merges them with its own share and the new
// token.  One more hand-off on the layer's critical path (~2 - 3 us), 7/8 of the cache walk off it (~1 us per 100 keys on ONE CU).
template <bool MASK, bool LEADER, bool OWNER, bool split, bool SPLITK = split>
__device__ __forceinline__ bool eg_attention(EgCtx& c, int layer) {
    constexpr int D = EG_HD, LPK = 16, KPW = 4, U = (MASK || SPLITK) ? 6 : 8;       // (8 keys per batch only where the registers allow: the short-context kernel without a mask)
    constexpr int KPB = split ? 96 : 12;                             // key streams of the head: 12 per CU
    static_assert(OWNER || split, "a helper exists only when the walk is split");
    const vcla_engine_args& a = *c.a;
    EgMisc* m = c.m;
    const int h = c.cu >> 3, pos = c.pos, lane = c.lane;
    const size_t per = (size_t)EG_H * a.ctx_max * D;
    bf16_t* kbase = a.kv + (size_t)(2 * layer) * per + (size_t)h * a.ctx_max * D;
    bf16_t* vbase = kbase + per;
    const int32_t* km = a.key_mask;
    const int cch = lane % LPK, lgrp = c.w * KPW + lane / LPK, grp = (split ? (c.cu & 7) * 12 : 0) + lgrp;
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(kbase, 0, a.ctx_max * D * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(vbase, 0, a.ctx_max * D * 2, 0x00020000);
    u32x4_t kA[U], vA[U], kB[U], vB[U];
    int mA[U], mB[U];
    const int last = pos > 0 ? pos - 1 : 0;
#define EG_FD_LOAD(KB_, VB_, MB_, t_)                                                                                \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                                  \
        int j_ = grp + ((t_) * U + u) * KPB;                                                                         \
        j_ = j_ < pos ? j_ : last;                                                                                   \
        const unsigned off_ = (unsigned)(j_ * D * 2 + cch * 16);                                                     \
        KB_[u] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rK, off_, 0, 2 /* nt */));         \
        VB_[u] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rV, off_, 0, 2 /* nt */));         \
        MB_[u] = MASK ? km[j_] : 1;                                                                                  \
    }                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);
    const int nb = (pos + U * KPB - 1) / (U * KPB);
    if (nb > 0) {            // the cache rows do not depend on this step's q: requested before the qkv granules are even complete
        EG_FD_LOAD(kA, vA, mA, 0)
        EG_FD_LOAD(kB, vB, mB, 1)
    }
    const unsigned seq = (unsigned)layer + 1;
    if constexpr (LEADER) {
        // q / k / v of the head: 64 granules each (lane t holds elements 2 t, 2 t + 1; the rotate-half partner sits in lane t ^ 32)
        const unsigned long long* mb = eg_mb(c, layer, EG_MB_QKV);
        const unsigned ep = eg_epoch(c, layer, 1);
        unsigned qd = 0, kd = 0, vd = 0;
        // the rotation of this position does not depend on the granules: requested before the wait for them, not behind it
        const int i0 = 2 * (lane & 31);                       // rotation index of this lane's pair
        const float2 cs = *reinterpret_cast<const float2*>(a.rope_cos + (size_t)pos * (D / 2) + i0);
        const float2 sn = *reinterpret_cast<const float2*>(a.rope_sin + (size_t)pos * (D / 2) + i0);
        for (unsigned it = 0;; ++it) {
            const unsigned long long xq = eg_peek(mb, h * 64 + lane);
            const unsigned long long xk = OWNER ? eg_peek(mb, 2048 + h * 64 + lane) : xq, xv = OWNER ? eg_peek(mb, 4096 + h * 64 + lane) : xq;      // (a helper needs q only)
            qd = (unsigned)xq; kd = (unsigned)xk; vd = (unsigned)xv;
            const bool good = (unsigned)(xq >> 32) == ep && (unsigned)(xk >> 32) == ep && (unsigned)(xv >> 32) == ep;
            if (__all(good)) break;
            if ((it & 15) == 15 && eg_ld(&m->fail)) return false;
            if (it > EG_SPIN_GLB) { eg_fail(m, c.state, 0x31); return false; }
            __builtin_amdgcn_s_sleep(2);
        }
        eg_stamp(c, layer, 10);
        const float c0 = Act<bf16_t>::rnd(cs.x), c1 = Act<bf16_t>::rnd(cs.y), s0 = Act<bf16_t>::rnd(sn.x), s1 = Act<bf16_t>::rnd(sn.y);
        const bool hi_half = lane >= 32;
        const float q0 = eg_lo(qd), q1 = eg_hi(qd), k0 = eg_lo(kd), k1 = eg_hi(kd);
        const float pq0 = __shfl_xor(q0, 32, 64), pq1 = __shfl_xor(q1, 32, 64), pk0 = __shfl_xor(k0, 32, 64), pk1 = __shfl_xor(k1, 32, 64);
        // first half: x * c - partner * s; second half: x * c + partner * s  (hf llama rotate_half; every product and sum rounded to bf16 once)
        const float rq0 = Act<bf16_t>::rnd(hi_half ? q0 * c0 + pq0 * s0 : q0 * c0 - pq0 * s0);
        const float rq1 = Act<bf16_t>::rnd(hi_half ? q1 * c1 + pq1 * s1 : q1 * c1 - pq1 * s1);
        const float rk0 = Act<bf16_t>::rnd(hi_half ? k0 * c0 + pk0 * s0 : k0 * c0 - pk0 * s0);
        const float rk1 = Act<bf16_t>::rnd(hi_half ? k1 * c1 + pk1 * s1 : k1 * c1 - pk1 * s1);
        m->qpk[lane] = pack_bf2(rq0, rq1);
        if constexpr (OWNER) {
            m->knew[2 * lane] = rk0; m->knew[2 * lane + 1] = rk1;
            m->vnew[2 * lane] = eg_lo(vd); m->vnew[2 * lane + 1] = eg_hi(vd);
            reinterpret_cast<unsigned*>(kbase + (size_t)pos * D)[lane] = pack_bf2(rk0, rk1);
            reinterpret_cast<unsigned*>(vbase + (size_t)pos * D)[lane] = vd;
        }
        eg_release();
        eg_st(&m->attn_ready, seq);
        eg_stamp(c, layer, 11);
    } else {
        if (!eg_wait_ge(&m->attn_ready, seq, m, c.state, 0x32)) return false;
    }
    eg_acquire();
    const u32x4_t qv = *reinterpret_cast<const u32x4_t*>(m->qpk + cch * 4);
    const float sl2 = a.scale * 1.44269504088896340736f;
    float m_run = -INFINITY, l_run = 0.f, o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#define EG_FD_COMPUTE(KB_, VB_, MB_, t_)                                                                             \
    {                                                                                                                \
        float s_[U];                                                                                                 \
        float mb_ = -INFINITY;                                                                                       \
        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                              \
            const int j_ = grp + ((t_) * U + u) * KPB;                                                               \
            const float d_ = eg_row_sum(eg_dot8s(qv, KB_[u], 0.f));                                                  \
            s_[u] = (j_ < pos && MB_[u] != 0) ? d_ * sl2 : -INFINITY;                                                \
            mb_ = fmaxf(mb_, s_[u]);                                                                                 \
        }                                                                                                            \
        const float mn_ = fmaxf(m_run, mb_);                                                                         \
        const float mu_ = mn_ == -INFINITY ? 0.f : mn_;                                                              \
        const float al_ = __builtin_amdgcn_exp2f(m_run - mu_);                                                       \
        m_run = mn_;                                                                                                 \
        float ps_ = 0.f;                                                                                             \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) o[e] *= al_;                                                   \
        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                              \
            const float p_ = __builtin_amdgcn_exp2f(s_[u] - mu_);                                                    \
            ps_ += p_;                                                                                               \
            float vv_[8];                                                                                            \
            bf8_to_f32(make_uint4(VB_[u].x, VB_[u].y, VB_[u].z, VB_[u].w), vv_);                                     \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(p_, vv_[e], o[e]);                   \
        }                                                                                                            \
        l_run = l_run * al_ + ps_;                                                                                   \
    }                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);
    for (int t = 0; t < nb; t += 2) {
        EG_FD_COMPUTE(kA, vA, mA, t)
        EG_FD_LOAD(kA, vA, mA, t + 2)
        EG_FD_COMPUTE(kB, vB, mB, t + 1)
        EG_FD_LOAD(kB, vB, mB, t + 3)
    }
#undef EG_FD_LOAD
#undef EG_FD_COMPUTE
    if constexpr (LEADER) eg_stamp(c, layer, 13);
    if constexpr (OWNER) {   // the new token (key / value in LDS): every group computes it, group 0 of the owner's workgroup folds it in
        float d_ = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const unsigned qq = m->qpk[cch * 4 + e / 2];
            d_ = __builtin_fmaf(eg_lo(qq), m->knew[cch * 8 + e], d_);
            d_ = __builtin_fmaf(eg_hi(qq), m->knew[cch * 8 + e + 1], d_);
        }
        d_ = eg_row_sum(d_);
        const float s_ = (lgrp == 0 && (!MASK || km[pos] != 0)) ? d_ * sl2 : -INFINITY;
        const float mn_ = fmaxf(m_run, s_);
        const float mu_ = mn_ == -INFINITY ? 0.f : mn_;
        const float al_ = __builtin_amdgcn_exp2f(m_run - mu_), p_ = __builtin_amdgcn_exp2f(s_ - mu_);
        m_run = mn_;
        l_run = l_run * al_ + p_;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(p_, m->vnew[cch * 8 + e], o[e] * al_);
    }
#pragma unroll
    for (int off = LPK; off < 64; off <<= 1) {       // merge the 4 lane groups of the wave
        const float m2 = __shfl_xor(m_run, off, 64), l2 = __shfl_xor(l_run, off, 64);
        const float mn_ = fmaxf(m_run, m2);
        const float mu_ = mn_ == -INFINITY ? 0.f : mn_;
        const float a1 = __builtin_amdgcn_exp2f(m_run - mu_), a2 = __builtin_amdgcn_exp2f(m2 - mu_);
        l_run = l_run * a1 + l2 * a2;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = o[e] * a1 + __shfl_xor(o[e], off, 64) * a2;
        m_run = mn_;
    }
    if constexpr (LEADER) eg_stamp(c, layer, 14);
    if (lane < LPK) {
        float* pw = m->part[c.w];
#pragma unroll
        for (int e = 0; e < 8; ++e) pw[lane * 8 + e] = o[e];
        if (lane == 0) { pw[D] = m_run; pw[D + 1] = l_run; }
    }
    eg_release();
    eg_st(&m->attn_done[c.w], seq);
    if constexpr (LEADER) {
        eg_stamp(c, layer, 12);
        if (!eg_wait_ge(&m->attn_done[1], seq, m, c.state, 0x33) || !eg_wait_ge(&m->attn_done[2], seq, m, c.state, 0x34)) return false;
        eg_acquire();
        float mf = fmaxf(fmaxf(m->part[0][D], m->part[1][D]), m->part[2][D]);      // (every lane: the 3 waves of this CU)
        float lf = 0.f, o8[8];
        {
            const float mu_ = mf == -INFINITY ? 0.f : mf;
            const int l16 = lane & 15;
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 3; ++w2) {
                const float* pw = m->part[w2];
                const float a_ = __builtin_amdgcn_exp2f(pw[D] - mu_);
                lf += pw[D + 1] * a_;
#pragma unroll
                for (int e = 0; e < 8; ++e) o8[e] += pw[l16 * 8 + e] * a_;
            }
        }
        if constexpr (split) {
            unsigned long long* mbp = eg_mb(c, layer, EG_MB_PART) + (size_t)h * 8 * EG_PART_GRAN;
            const unsigned epp = eg_epoch(c, layer, 6);
            if constexpr (!OWNER) {
                // a helper: its unnormalised share goes to the owner as fp32 granules
                unsigned long long* mine = mbp + (size_t)(c.cu & 7) * EG_PART_GRAN;
                if (lane < 16) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) eg_publish(mine, lane * 8 + e, epp, __float_as_uint(o8[e]));
                    if (lane == 0) { eg_publish(mine, D, epp, __float_as_uint(mf)); eg_publish(mine, D + 1, epp, __float_as_uint(lf)); }
                }
                return true;
            } else {
                // the owner: lane group g merges the shares of CUs g and g + 4 of the head's group (its own: the registers above), then the groups meet
                const int own = c.cu & 7, g = lane >> 4, l16 = lane & 15;
                // (one share at a time: both in registers at once spill next to the o_proj slots held across this code)
                float rm = -INFINITY, rl = 0.f, ro[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) ro[e] = 0.f;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int p = g + 4 * k;
                    const unsigned long long* src = mbp + (size_t)p * EG_PART_GRAN;
                    float pm = mf, pl = lf, po[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) po[e] = o8[e];
                    for (unsigned it = 0;; ++it) {
                        bool good = true;
                        if (p != own) {
                            const unsigned long long xm = eg_peek(src, D), xl = eg_peek(src, D + 1);
                            good = (unsigned)(xm >> 32) == epp && (unsigned)(xl >> 32) == epp;
                            pm = __uint_as_float((unsigned)xm); pl = __uint_as_float((unsigned)xl);
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const unsigned long long xo = eg_peek(src, l16 * 8 + e);
                                good = good && (unsigned)(xo >> 32) == epp;
                                po[e] = __uint_as_float((unsigned)xo);
                            }
                        }
                        if (__all(good)) break;
                        if ((it & 15) == 15 && eg_ld(&m->fail)) return false;
                        if (it > EG_SPIN_GLB) { eg_fail(m, c.state, 0x35); return false; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    const float mn_ = fmaxf(rm, pm);
                    const float mu_ = mn_ == -INFINITY ? 0.f : mn_;
                    const float a0 = __builtin_amdgcn_exp2f(rm - mu_), a1 = __builtin_amdgcn_exp2f(pm - mu_);
                    rm = mn_; rl = rl * a0 + pl * a1;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ro[e] = ro[e] * a0 + po[e] * a1;
                }
                mf = rm; lf = rl;
#pragma unroll
                for (int e = 0; e < 8; ++e) o8[e] = ro[e];
#pragma unroll
                for (int off = 16; off < 64; off <<= 1) {
                    const float m2 = __shfl_xor(mf, off, 64), l2 = __shfl_xor(lf, off, 64);
                    const float mn_ = fmaxf(mf, m2);
                    const float mu_ = mn_ == -INFINITY ? 0.f : mn_;
                    const float a1 = __builtin_amdgcn_exp2f(mf - mu_), a2 = __builtin_amdgcn_exp2f(m2 - mu_);
                    lf = lf * a1 + l2 * a2;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o8[e] = o8[e] * a1 + __shfl_xor(o8[e], off, 64) * a2;
                    mf = mn_;
                }
            }
        }
        if constexpr (OWNER) {
            if (lane < 16) {
                const float inv = lf > 0.f ? 1.0f / lf : 0.f;
                unsigned long long* mbo = eg_mb(c, layer, EG_MB_AO);
                const unsigned epo = eg_epoch(c, layer, 2);
#pragma unroll
                for (int e = 0; e < 8; e += 2) eg_publish(mbo, h * 64 + lane * 4 + e / 2, epo, pack_bf2(o8[e] * inv, o8[e + 1] * inv));
            }
        }
    }
    return true;
}

// One consumer wave over the whole step.  LEADER = consumer 0: sweeps the mailboxes, stages every operator's input vector, finishes down_proj,
// owns no slot among the first 2 * NP of an operator; consumers 1 / 2 pull those into registers while the leader sweeps.
// SPLITK: this instantiation contains the split attention (kernels launched for caches that can outgrow args.split_min; the short-context kernel is
// the code of the round's first form, bit for bit: carrying the split variants costs it 0.5 - 1 % -- SGPR spills, a third more code)
template <bool MASK, bool LEADER, bool SPLITK>
__device__ __forceinline__ void eg_consumer(EgCtx& c) {
    const vcla_engine_args& a = *c.a;
    const vcla_engine_geom& G = a.g;
    EgMisc* m = c.m;
    const bool attn_cu = (c.cu & 7) == ((c.cu >> 3) & 7);
    // long context: the 8 CUs of a head's group share its cache walk (eg_attention).  (`split_min > 0` is implied by SPLITK; written out because hipcc's
    // register allocation of the WHOLE kernel hinges on it: without the redundant test the split-capable kernels spill 16 - 25 registers to scratch)
    const bool split = SPLITK && a.split_min > 0 && c.pos >= a.split_min;
    // the leader may overwrite xin only when the other consumers have finished the operator that reads it
    auto others_done = [&](unsigned seq) -> bool {
        return eg_wait_ge(&m->cons_done[1], seq, m, c.state, 0x41) && eg_wait_ge(&m->cons_done[2], seq, m, c.state, 0x42);
    };
    auto ready = [&](unsigned seq) { eg_release(); eg_st(&m->xin_ready, seq); };
    constexpr int NPQ = LEADER ? 1 : EgUnit<EG_OP_QKV>::NP, NPO = EgUnit<EG_OP_O>::NP, NPG = LEADER ? 1 : EgUnit<EG_OP_GU>::NP, NPD = EgUnit<EG_OP_DN>::NP;
    for (int l = 0; l < G.n_layers; ++l) {
        const int g0 = l * G.slots_layer;
        const unsigned s0 = (unsigned)l * 4 + 1;
        eg_fresh(c);
        if constexpr (LEADER) eg_stamp(c, l, 0);
        {
            u32x4_t pre[NPQ][16];
            if constexpr (LEADER) {
                if (l > 0 && !others_done(s0 - 1)) return;
                if (!eg_stage_norm(c, eg_mb(c, l, EG_MB_X), eg_epoch(c, l, 0), a.gamma + (size_t)(2 * l) * EG_D, l == 0)) return;
                ready(s0);
                eg_stamp(c, l, 1);
            } else {
                if (!eg_preload<EG_OP_QKV>(c, g0, pre)) return;
            }
            eg_fresh(c);
            if (!eg_run_rows<EG_OP_QKV, LEADER>(c, g0, EG_S_QKV, l, s0, pre)) return;
        }
        if constexpr (LEADER) eg_stamp(c, l, 2);
        eg_fresh(c);
        {
            // o_proj's first slots go into registers BEFORE the attention: the ring then takes the first gate/up slots while the attention runs
            u32x4_t pre_o[NPO][16];
            if (!eg_preload<EG_OP_O>(c, g0 + EG_S_QKV, pre_o)) return;
            if constexpr (SPLITK) {
                if (attn_cu) {
                    if (split) { if (!eg_attention<MASK, LEADER, true, true>(c, l)) return; }
                    else { if (!eg_attention<MASK, LEADER, true, false>(c, l)) return; }
                } else if (split) { if (!eg_attention<MASK, LEADER, false, true>(c, l)) return; }
            } else {
                if (attn_cu && !eg_attention<MASK, LEADER, true, false>(c, l)) return;
            }
            eg_fresh(c);
            if constexpr (LEADER) eg_stamp(c, l, 3);
            if (!eg_run_kmajor<EG_OP_O, LEADER>(c, g0 + EG_S_QKV, EG_S_O, l, s0 + 1, eg_mb(c, l, EG_MB_AO), eg_epoch(c, l, 2), pre_o)) return;
        }
        if constexpr (LEADER) eg_stamp(c, l, 5);
        eg_fresh(c);
        {
            u32x4_t pre[NPG][16];
            if constexpr (LEADER) {
                if (!others_done(s0)) return;               // (qkv's readers of xin; o_proj does not touch it)
                if (!eg_stage_norm(c, eg_mb(c, l, EG_MB_X1), eg_epoch(c, l, 3), a.gamma + (size_t)(2 * l + 1) * EG_D, false)) return;
                ready(s0 + 2);
                eg_stamp(c, l, 6);
            } else {
                if (!eg_preload<EG_OP_GU>(c, g0 + EG_S_QKV + EG_S_O, pre)) return;
            }
            eg_fresh(c);
            if (!eg_run_rows<EG_OP_GU, LEADER>(c, g0 + EG_S_QKV + EG_S_O, G.upc, l, s0 + 2, pre)) return;
        }
        if constexpr (LEADER) eg_stamp(c, l, 7);
        eg_fresh(c);
        {
            u32x4_t pre[NPD][16];
            if (!eg_preload<EG_OP_DN>(c, g0 + EG_S_QKV + EG_S_O + G.upc, pre)) return;
            eg_fresh(c);
            if (!eg_run_kmajor<EG_OP_DN, LEADER>(c, g0 + EG_S_QKV + EG_S_O + G.upc, G.s_dn, l, s0 + 3, eg_mb(c, l, EG_MB_ACT), eg_epoch(c, l, 4), pre)) return;
        }
        if constexpr (LEADER) eg_stamp(c, l, 9);
    }
    const int L = G.n_layers;
    const unsigned sl = (unsigned)L * 4 + 1;
    eg_fresh(c);
    {
        u32x4_t pre[NPQ][16];
        if constexpr (LEADER) {
            if (!others_done(sl - 1)) return;
            if (!eg_stage_norm(c, eg_mb(c, L, EG_MB_X), eg_epoch(c, L, 0), a.gamma + (size_t)(2 * L) * EG_D, false)) return;
            ready(sl);
        } else {
            if (!eg_preload<EG_OP_LM>(c, L * G.slots_layer, pre)) return;
        }
        eg_fresh(c);
        if (!eg_run_rows<EG_OP_LM, LEADER>(c, L * G.slots_layer, G.s_lm, L, sl, pre)) return;
    }
    if constexpr (LEADER) {
        if (a.tail_ids_out) {
            if (!others_done(sl)) return;
            eg_acquire();
            if (!eg_greedy_tail(c, L)) return;
        }
        // the launch sequence number moves on once per successful launch (every workgroup read it before its first publish, and this store sits
        // behind the last all-gather, which needed all of them)
        if (c.cu == 0 && c.lane == 0) c.state[0] = (c.eb >> 10) + 1;
    }
}

__device__ __forceinline__ void eg_loader(const vcla_engine_args& a, unsigned ring_u, EgMisc* m, unsigned* state, int cu, int lane) {
    const int total = a.g.slots_total;
    const unsigned char* src = a.stream + (size_t)cu * a.g.cu_stride + lane * 16;
    const size_t slot_stride = a.g.slot_stride;
    unsigned pub = 0;
    unsigned long long stall = 0, n_stall = 0;
    const unsigned long long t_begin = a.timeline ? wall_clock64() : 0ull;
    for (int g = 0; g < total; ++g) {
        const int p = g % EG_NRING;
        if (g >= EG_NRING && eg_ld(&m->freed[p]) < (unsigned)(g - EG_NRING + 1)) {
            // ring full: nothing to issue, so everything issued may as well be published
            const unsigned long long t0 = a.timeline ? wall_clock64() : 0ull;
            eg_vmcnt<0>();
            if (pub < (unsigned)g) { pub = g; eg_st(&m->filled, pub); }
            if (!eg_wait_ge(&m->freed[p], (unsigned)(g - EG_NRING + 1), m, state, 0x01)) break;
            if (a.timeline) {
                const unsigned long long dt = wall_clock64() - t0;
                stall += dt; ++n_stall;
                if (lane == 0 && g < a.g.n_layers * a.g.slots_layer)       // by slot of the layer (an atomic without return: nothing the loader would wait for)
                    (void)__hip_atomic_fetch_add(a.timeline + (size_t)cu * EG_TL_STRIDE + 1024 + g % a.g.slots_layer, dt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const unsigned dst = ring_u + p * EG_SLOT;
        const unsigned char* s = src + (size_t)g * slot_stride;
#pragma unroll
        for (int i = 0; i < 16; ++i) eg_dma16_nt(s + i * 1024, dst + i * 1024);
        // ONE fill behind the one just issued: with the slot-major stream HBM answers in ~1 us, 16 - 32 KiB in flight per CU keep it busy, and every
        // further miss in flight only delays the CU's own mailbox polls.  Per token on one box (vmcnt 0 / 8 / 12 / 16 / 20 / 32 / 48): 3.10 / 2.74 / 2.57 /
        // 2.13 - 2.17 / 2.15 / 2.15 - 2.18 / 2.19 - 2.22 ms.  (The CU-major stream wanted three fills in flight and a thinner loader during sweeps.)
        eg_vmcnt<16>();
        const unsigned landed = g;                                        // slots < g have landed
        if (landed > pub) { pub = landed; eg_st(&m->filled, pub); }
    }
    eg_vmcnt<0>();
    eg_st(&m->filled, (unsigned)total);
    if (a.timeline && lane == 0) {
        unsigned long long* tl = a.timeline + (size_t)cu * EG_TL_STRIDE + 2040;
        tl[0] = t_begin; tl[1] = wall_clock64(); tl[2] = stall; tl[3] = n_stall;
    }
}

template <bool MASK, bool SPLITK>
__global__ __launch_bounds__(256, 1) void decode_engine_kernel(vcla_engine_args a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char eg_lds[];
    EgMisc* m = reinterpret_cast<EgMisc*>(eg_lds + EG_MISC_OFF);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned* state = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(a.mbox) + EG_WS_STATE_OFF);
    for (int i = threadIdx.x; i < 32; i += 256) reinterpret_cast<unsigned*>(m)[i] = 0u;      // the flag words
    const unsigned seq = state[0];
    const int pos = a.pos0 + (a.pos_dev ? *a.pos_dev : 0);
    __syncthreads();
    if (wave == 0) {
        eg_loader(a, (unsigned)(uintptr_t)(eg_lds_ptr_t)eg_lds, m, state, blockIdx.x, lane);
    } else {
        EgCtx c;
        c.a = &a; c.ring = eg_lds; c.xin = eg_lds + EG_RING_BYTES; c.m = m; c.state = state;
        c.w = wave - 1; c.lane = lane; c.cu = blockIdx.x; c.eb = (seq << 10) + 1u; c.pos = pos;
        if (a.fault && blockIdx.x == 7) return;                  // test hook: a workgroup that never publishes (tests/test_gpu_engine.py)
        if (wave == 1) eg_consumer<MASK, true, SPLITK>(c); else eg_consumer<MASK, false, SPLITK>(c);
    }
}

bool vcla_engine_geometry(int hidden, int heads, int inter, int vocab, int n_layers, vcla_engine_geom* g) {
    if (hidden != EG_D || heads != EG_H || inter <= 0 || inter % EG_NCU || n_layers <= 0 || n_layers > 120 || vocab <= 0) return false;
    const int upc = inter / EG_NCU, gpc = (upc + 1) / 2;
    if (EG_NCU * gpc > EG_MB_MAX_ACT) return false;
    const int s_lm_ = (vocab + 2 * EG_NCU - 1) / (2 * EG_NCU);
    if (upc < 2 * EG_PRE_MAX || gpc < 2 * EG_PRE_MAX || s_lm_ < 2 * EG_PRE_MAX) return false;     // every operator holds the unconditional register preloads
    g->n_layers = n_layers; g->inter = inter; g->vocab = vocab;
    g->upc = upc; g->s_dn = gpc;
    g->s_lm = (vocab + 2 * EG_NCU - 1) / (2 * EG_NCU);
    g->slots_layer = EG_S_QKV + EG_S_O + upc + gpc;
    g->slots_total = n_layers * g->slots_layer + g->s_lm;
    g->cu_stride = EG_SLOT; g->slot_stride = (size_t)EG_NCU * EG_SLOT;                          // [slot][CU][16 KiB]
    return true;
}

int vcla_engine_launch(const vcla_engine_args* a, hipStream_t s) {
    static bool attr_set[4][VCLA_MAX_DEVICES] = {};
    // the kernel with the split attention only for caches that can outgrow split_min (a captured step is replayed at every position up to ctx_max)
    const bool sk = a->split_min > 0 && a->ctx_max > a->split_min;
#define EG_GO(MASK_, SK_, I_)                                                                                                  \
    {                                                                                                                          \
        const int rc = vcla_raise_dyn_lds((const void*)decode_engine_kernel<MASK_, SK_>, EG_LDS_BYTES, attr_set[I_]);          \
        if (rc) return rc;                                                                                                     \
        decode_engine_kernel<MASK_, SK_><<<EG_NCU, 256, EG_LDS_BYTES, s>>>(*a);                                                \
    }
    if (a->key_mask) { if (sk) EG_GO(true, true, 3) else EG_GO(true, false, 1) }
    else { if (sk) EG_GO(false, true, 2) else EG_GO(false, false, 0) }
#undef EG_GO
    VCLA_CHECK_LAUNCH("decode_engine_kernel");
    return VCLA_OK;
}
