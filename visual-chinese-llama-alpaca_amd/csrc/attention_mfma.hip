// attention_mfma.hip -- flash attention on v_mfma_f32_16x16x32_bf16 for the bf16 product path
// (ViT bidirectional N=257 d=64, resampler cross attention 64 x 321 d=64, LLaMA causal prefill d=128).
//
// Workgroup = NW waves = NW * 32 query rows of one (batch, head); each wave owns 32 query rows (two 16-row MFMA
// tiles) and walks the keys in tiles of 64 staged through LDS once per workgroup.  NW = 4 for the causal prefill;
// bidirectional attention over a short sequence takes the whole sequence in ONE workgroup when it fits (ViT: 257 rows =
// 9 waves): K / V are staged once per (batch, head) instead of once per 128-row block -- with 4 waves the 257th row cost a
// third workgroup that staged every tile for a single query (ViT attention at B = 64: 102 -> see profiles/).
//
// Both products are issued "swapped" so that the softmax row lives in ONE lane column:
//   S^T = K . Q^T   (A port = K rows from LDS, B port = Q rows from registers)
//       -> lane (q = lane & 15, g = lane >> 4) holds S[q][key = t*16 + g*4 + r], r = 0..3, t = 0..3
//   O^T = V^T . P^T (A port = V^T rows from LDS, B port = P^T straight from the S^T accumulators)
//       -> lane holds O[q][d = dt*16 + g*4 + r]: 4 consecutive d of one query row (8-byte stores), and the
//          online-softmax rescale is a per-lane scalar.
// The MFMA k-slots of the PV product are whatever keys the lane already holds: k-slot (g, j) of key-step s is key
// (2s + j/4)*16 + g*4 + j%4.  The V fragments are fetched in exactly that key order (LDS transpose reads, below), so P never moves
// between lanes.
// Row max / row sum need only 2 cross-lane steps (xor 16, xor 32).
//
// LDS: K tile [64][D] bf16 and V tile [64][D] bf16 (both row-major), 16-byte chunks XOR-swizzled so every ds_read_b128
// fragment read is bank-conflict-free.  Next tile is prefetched into registers under the MFMAs.
#include "vcla_common.h"
#include <stdlib.h>

#define FA_KV 64

bool vcla_attention_mfma_supported(const vcla_attn_args* a);

template <int D> __device__ __forceinline__ int fa_k_off(int key, int ch) {
    if (D == 128) return key * 256 + ((ch ^ (key & 15)) << 4);
    return key * 128 + ((ch ^ ((key >> 1) & 7)) << 4);
}
// V tile: ROW-major [64 keys][D] bf16 (round 3; was V^T with 2-byte scatter stores).  The P V product wants V^T rows as its A operand
// -- 8 keys of one d per lane -- which gfx950's LDS transpose read delivers straight from the row-major image:
// ds_read_b64_tr_b16, per 16-lane group, lane i supplies the address of piece (row i / 4, 8 bytes i % 4) of a 4-row x 16-column
// block and receives COLUMN i of those 4 rows (measured: tools/debug/probe_tr16.py).  Rows = 4 consecutive keys, columns = the 16
// d of an output tile: two such reads give the 8 k-slots (g, j) = keys (2s + j/4)*16 + g*4 + j%4 of a key step, so P still never
// moves between lanes and V is staged with ONE 16-byte store per chunk instead of eight 2-byte ones (the V^T scatter was 4-way
// bank-conflicted and, at 8 stores x 9 waves x 4 tiles, the busiest thing on the CU).  32-byte column blocks are XOR-swizzled with
// key bits 1-2 so that the 8 rows a 32-lane half touches (128 B apart = same banks 2 rows apart) land on different banks.
template <int D> __device__ __forceinline__ int fa_v_off(int key, int d) {   // byte offset of element (key, d); d % 4 == 0 for the tr reads
    constexpr int CB = D / 16;                                                // 32-byte column blocks per row
    const int f = D == 128 ? key : key >> 1;       // rows 256 B apart share every bank; rows 128 B apart share them two rows on
    return key * (D * 2) + ((((d >> 4) ^ f) & (CB - 1)) << 5) + ((d & 15) << 1);
}
typedef __attribute__((ext_vector_type(4))) short fa_s16x4_t;
typedef __attribute__((address_space(3))) fa_s16x4_t* fa_lds_v4_t;

// MAXT > 0: the WHOLE key sequence (<= MAXT tiles of 64 keys) is staged in LDS at once -- every K / V chunk is requested up front
// (one exposed memory latency per workgroup instead of one per tile), ONE barrier, and from then on the waves run their tiles with
// no workgroup synchronisation at all, so the MFMA and the softmax VALU phases of different waves interleave freely.  MAXT = 0: the
// tile-by-tile form (one 16 / 32 KiB tile in LDS, two barriers per tile) for sequences that do not fit.
template <int D, int NW, int MAXT>
__global__ __launch_bounds__(NW * 64) void attn_mfma_kernel(vcla_attn_args a) {
    constexpr int FA_QB = NW * 32, NT = NW * 64;
    constexpr int KST = D / 32;         // MFMA k-steps over the head dim (Q K^T)
    constexpr int DT = D / 16;          // 16-wide output d tiles (P V)
    constexpr int CH = D / 8;           // 16-byte chunks per K/V row
    constexpr int NLD = (FA_KV * CH + NT - 1) / NT;  // staging loads per thread per operand
    constexpr int TILE_BYTES = 2 * FA_KV * D * 2;                                           // [K tile | V tile]
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];                 // TILE_BYTES x max(MAXT, 1)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * FA_QB;           // first query row of this workgroup
    const int qw = q0 + wave * 32;               // first query row of this wave
    const int Tq = a.Tq, Tk = a.Tk;
    const int coff = Tk - Tq;                    // causal: query i sees keys <= i + coff
    const bf16_t* qb = (const bf16_t*)a.q + b * a.q_bs + h * a.q_hs;
    const bf16_t* kb = (const bf16_t*)a.k + b * a.k_bs + h * a.k_hs;
    const bf16_t* vb = (const bf16_t*)a.v + b * a.v_bs + h * a.v_hs;
    bf16_t* ob = (bf16_t*)a.o + b * a.o_bs + h * a.o_hs;
    const int32_t* km = a.key_mask ? a.key_mask + b * a.key_mask_ld : nullptr;
    const int ql = lane & 15, g = lane >> 4;
    const bool wave_active = qw < Tq;            // waves past the end only help staging

    // ---- Q fragments (B port): lane (q = ql, g) holds Q[q][ks*32 + g*8 .. +8]
    bf16x8_t qf[2][KST];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        int qr = qw + qt * 16 + ql;
        if (qr >= Tq) qr = Tq - 1;
#pragma unroll
        for (int s = 0; s < KST; ++s)
            qf[qt][s] = *reinterpret_cast<const bf16x8_t*>(qb + (int64_t)qr * a.q_rs + s * 32 + g * 8);
    }

    f32x4_t o[2][DT];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    const float sl2 = a.scale * 1.44269504088896340736f;  // scores are kept in the log2 domain

    // keys this workgroup needs (causal: up to its last query row)
    int kv_end = Tk;
    if (a.causal) {
        const int last_q = (q0 + FA_QB < Tq ? q0 + FA_QB : Tq) - 1;
        kv_end = last_q + coff + 1 < Tk ? last_q + coff + 1 : Tk;
    }
    // Bidirectional sequences here end ONE key past a multiple of 64 (ViT: 257 = 4 x 64 + 1, 577 = 9 x 64 + 1 tokens; resampler: 321 /
    // 641 keys): as a tile that key would cost a full 64-key tile (20 % of the ViT attention).  Up to 4 such remainder keys are
    // peeled off the tile loop and folded in after it on the VALU (n_extra below): q . k on v_dot2c against the Q fragments the
    // lane already holds, one online-softmax update, 4 x DT FMAs into O.  The tile loop then sees whole tiles only (no mask pass).
    int n_extra = 0;
    if (!a.causal && !km && Tk > FA_KV && (Tk % FA_KV) >= 1 && (Tk % FA_KV) <= 4) { n_extra = Tk % FA_KV; kv_end = Tk - n_extra; }
    const int ntiles = (kv_end + FA_KV - 1) / FA_KV;

    // ---- staging maps
    u32x4_t rk[NLD], rv[NLD];
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int id0 = i * NT + tid, id = id0 < FA_KV * CH ? id0 : FA_KV * CH - 1;   // (threads past the tile re-load its last chunk)
            const int key = id / CH, ch = id % CH;
            int kg = tile * FA_KV + key;
            if (kg >= Tk) kg = Tk - 1;  // clamp; masked below
            rk[i] = *reinterpret_cast<const u32x4_t*>(kb + (int64_t)kg * a.k_rs + ch * 8);
            rv[i] = *reinterpret_cast<const u32x4_t*>(vb + (int64_t)kg * a.v_rs + ch * 8);
        }
    };
    auto store_tile = [&](int slot) {
        unsigned char* ks = lds_all + slot * TILE_BYTES;
        unsigned char* vts = ks + FA_KV * D * 2;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int id = i * NT + tid, key = id / CH, ch = id % CH;
            if (id >= FA_KV * CH) continue;
            *reinterpret_cast<u32x4_t*>(ks + fa_k_off<D>(key, ch)) = rk[i];
            *reinterpret_cast<u32x4_t*>(vts + fa_v_off<D>(key, ch * 8)) = rv[i];
        }
    };

    // ---- one key tile of 64 keys against the wave's 32 query rows
    auto tile_body = [&](int kv0, int slot) {
        constexpr int NTK = 4, NS = 2;
        const unsigned char* ks = lds_all + slot * TILE_BYTES;
        const auto fa_lds_base = (__attribute__((address_space(3))) unsigned char*)lds_all + slot * TILE_BYTES + FA_KV * D * 2;   // the V tile, as an LDS pointer
        // ---- S^T = K Q^T
        f32x4_t sacc[2][NTK];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int t = 0; t < NTK; ++t) sacc[qt][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KST; ++s) {
#pragma unroll
            for (int t = 0; t < NTK; ++t) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(ks + fa_k_off<D>(t * 16 + ql, s * 4 + g));
                sacc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][s], sacc[0][t], 0, 0, 0);
                sacc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][s], sacc[1][t], 0, 0, 0);
            }
        }
        // ---- masks + online softmax; lane holds keys kv0 + t*16 + g*4 + r of query rows qw + qt*16 + ql.
        // At d = 64 this section, not the MFMAs, bounds the kernel (one v_exp + the VALU around it per score against 2 x 64 MFMA
        // flops): the visibility test runs only on tiles that need one (wave-uniform: padding mask, the last tile, the causal
        // diagonal), the row maximum is taken on the raw scores (scale > 0), and scale / max-subtraction are one FMA feeding a raw
        // v_exp_f32.
        const bool need_mask = km != nullptr || kv0 + NTK * 16 > Tk || (a.causal && kv0 + NTK * 16 - 1 > qw + coff);
        if (need_mask) {
            // key-padding mask of the 64 keys of this tile as a wave-uniform bit mask (one load per lane + ballot)
            unsigned long long tmask = ~0ull;
            if (km) {  // wave-uniform
                int key = kv0 + lane;
                key = key < Tk ? key : Tk - 1;
                tmask = __ballot(km[key] != 0);
            }
            const unsigned long long lmask = tmask >> (g * 4);   // bit (t*16 + r) = key t*16 + g*4 + r
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int qrow = qw + qt * 16 + ql;
                const int klim = a.causal ? (qrow + coff < Tk - 1 ? qrow + coff : Tk - 1) : Tk - 1;  // last visible key
#pragma unroll
                for (int t = 0; t < NTK; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kv0 + t * 16 + g * 4 + r;
                        const bool ok = (key <= klim) & (((lmask >> (t * 16 + r)) & 1ull) != 0);
                        sacc[qt][t][r] = ok ? sacc[qt][t][r] : -INFINITY;
                    }
            }
        }
        bf16x8_t pf[2][NS];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < NTK; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[qt][t][r]);
            mx *= sl2;                                   // log2 domain (-inf stays -inf)
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[qt], mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;  // fully masked so far: keep everything at 0
            const float alpha = exp2f(m_run[qt] - m_use);             // exp2(-inf) = 0 on the first tile
            m_run[qt] = m_new;
            float ps = 0.f;
            float pv[NTK * 4];
#pragma unroll
            for (int t = 0; t < NTK; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[qt][t][r], sl2, -m_use));   // raw v_exp_f32: exp2f() adds denormal-range scaling, ~4 VALU per score
                    pv[t * 4 + r] = p;
                    ps += p;
                }
            l_run[qt] = l_run[qt] * alpha + ps;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[qt][dt] *= alpha;
            // P^T fragments: key-step s takes S tiles (2s, 2s+1): slots j<4 from tile 2s, j>=4 from tile 2s+1
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                uint4 u;
                u.x = pack_bf2(pv[(2 * s) * 4 + 0], pv[(2 * s) * 4 + 1]);
                u.y = pack_bf2(pv[(2 * s) * 4 + 2], pv[(2 * s) * 4 + 3]);
                u.z = pack_bf2(pv[(2 * s + 1) * 4 + 0], pv[(2 * s + 1) * 4 + 1]);
                u.w = pack_bf2(pv[(2 * s + 1) * 4 + 2], pv[(2 * s + 1) * 4 + 3]);
                pf[qt][s] = __builtin_bit_cast(bf16x8_t, u);
            }
        }
        // ---- O^T += V^T P^T
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                // lane (ql, g): piece (key row ql / 4, 8 bytes ql % 4) of the 4-key x 16-d blocks of S tiles 2s and 2s + 1
                const int kr = g * 4 + (ql >> 2), dc = dt * 16 + (ql & 3) * 4;
                const fa_s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fa_lds_v4_t)(fa_lds_base + fa_v_off<D>((2 * s) * 16 + kr, dc)));
                const fa_s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fa_lds_v4_t)(fa_lds_base + fa_v_off<D>((2 * s + 1) * 16 + kr, dc)));
                const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
                o[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[0][s], o[0][dt], 0, 0, 0);
                o[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[1][s], o[1][dt], 0, 0, 0);
            }
        }
    };
    auto run_tile = [&](int tile, int slot) {
        const int kv0 = tile * FA_KV;
        // causal: a wave whose rows all precede this tile has nothing to do here
        const bool skip = !wave_active || (a.causal && kv0 > (qw + 31 < Tq ? qw + 31 : Tq - 1) + coff);
        if (skip) return;
        tile_body(kv0, slot);
    };

    // (Measured and dropped, profiles/r03_*: both tiles double-buffered in LDS so that a key tile costs one barrier instead of two --
    // 58.3 -> 62.3 us for the ViT at B = 64, 23.0 -> 25.4 us for the resampler: the second buffer's LDS costs more co-residency
    // than the barrier it removes.)
    if constexpr (MAXT > 0) {
        // whole-sequence form (the launcher guarantees ntiles <= MAXT): request everything, park it, one barrier, compute
        u32x4_t ak[MAXT][NLD], av[MAXT][NLD];
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < ntiles) {
                load_tile(t);
#pragma unroll
                for (int i = 0; i < NLD; ++i) { ak[t][i] = rk[i]; av[t][i] = rv[i]; }
            }
        }
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < ntiles) {
#pragma unroll
                for (int i = 0; i < NLD; ++i) { rk[i] = ak[t][i]; rv[i] = av[t][i]; }
                store_tile(t);
            }
        }
        __syncthreads();
        for (int tile = 0; tile < ntiles; ++tile) run_tile(tile, tile);
    } else {
        if (ntiles > 0) load_tile(0);
        for (int tile = 0; tile < ntiles; ++tile) {
            __syncthreads();  // previous tile fully consumed
            store_tile(0);
            __syncthreads();
            load_tile(tile + 1 < ntiles ? tile + 1 : tile);  // unconditional (last one is a harmless re-load): keeps the
                                                             // prefetch registers out of scratch
            run_tile(tile, 0);
        }
    }

    if (!wave_active) return;
    // ---- remainder keys (see n_extra above): lane (ql, g) holds Q[q][s*32 + g*8 .. +8] -- the K row's same 8-element groups give its
    // share of q . k (xor 16 / 32 completes the 64 / 128 dims); the lane's O values are d = dt*16 + g*4 + r -> 8 bytes of V per dt.
    for (int x = 0; x < n_extra; ++x) {
        const int key = kv_end + x;
        bf16x8_t kx[KST];
#pragma unroll
        for (int s = 0; s < KST; ++s) kx[s] = *reinterpret_cast<const bf16x8_t*>(kb + (int64_t)key * a.k_rs + s * 32 + g * 8);
        uint2 vx[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vx[dt] = *reinterpret_cast<const uint2*>(vb + (int64_t)key * a.v_rs + dt * 16 + g * 4);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float d_ = 0.f;
#pragma unroll
            for (int s = 0; s < KST; ++s) {
                d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qf[qt][s], qf[qt][s], 0, 1), __builtin_shufflevector(kx[s], kx[s], 0, 1), d_, false);
                d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qf[qt][s], qf[qt][s], 2, 3), __builtin_shufflevector(kx[s], kx[s], 2, 3), d_, false);
                d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qf[qt][s], qf[qt][s], 4, 5), __builtin_shufflevector(kx[s], kx[s], 4, 5), d_, false);
                d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qf[qt][s], qf[qt][s], 6, 7), __builtin_shufflevector(kx[s], kx[s], 6, 7), d_, false);
            }
            d_ += __shfl_xor(d_, 16, 64);
            d_ += __shfl_xor(d_, 32, 64);
            const float sx = d_ * sl2;
            const float m_new = fmaxf(m_run[qt], sx);              // finite: sx is
            const float alpha = exp2f(m_run[qt] - m_new), px = __builtin_amdgcn_exp2f(sx - m_new);
            m_run[qt] = m_new;
            l_run[qt] = l_run[qt] * alpha + (g == 0 ? px : 0.f);   // l is a per-lane partial, summed over g in the epilogue
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                o[qt][dt][0] = __builtin_fmaf(px, __uint_as_float(vx[dt].x << 16), o[qt][dt][0] * alpha);
                o[qt][dt][1] = __builtin_fmaf(px, __uint_as_float(vx[dt].x & 0xffff0000u), o[qt][dt][1] * alpha);
                o[qt][dt][2] = __builtin_fmaf(px, __uint_as_float(vx[dt].y << 16), o[qt][dt][2] * alpha);
                o[qt][dt][3] = __builtin_fmaf(px, __uint_as_float(vx[dt].y & 0xffff0000u), o[qt][dt][3] * alpha);
            }
        }
    }
    // ---- epilogue: O[q][d] = o / l ; lane holds d = dt*16 + g*4 + r for query row qw + qt*16 + ql
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float l = l_run[qt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        const int qrow = qw + qt * 16 + ql;
        if (qrow >= Tq) continue;
        bf16_t* orow = ob + (int64_t)qrow * a.o_rs;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            float v[4] = {o[qt][dt][0] * inv, o[qt][dt][1] * inv, o[qt][dt][2] * inv, o[qt][dt][3] * inv};
            Act<bf16_t>::st4(orow + dt * 16 + g * 4, v);
        }
    }
}

// =================================================================== ViT self-attention, whole sequence per workgroup (round 4)
// Bidirectional, unmasked, d = 64, Tq = Tk = NWM * 64 + 1 (ViT-L/14 at 224 px: 257 = 4 * 64 + 1 tokens).  The tile-by-tile kernel
// above spent its time waiting: one key tile is ~0.9 us of MFMA + softmax work per workgroup against ~2 us of global latency per
// tile and two workgroup barriers, and the 9th wave carried ONE valid query row (57 us per layer at B = 64, 12 % of the MFMA peak).
// Here a workgroup is NWM waves, one (batch, head):
//   * every K / V chunk of the sequence is requested up front and parked in LDS (64 + 64 KiB... no: 257 x 128 B x 2 = 64.25 KiB),
//     ONE barrier, no workgroup synchronisation inside the key loop;
//   * each wave owns 64 query rows = FOUR 16-row MFMA tiles: a K / V^T fragment read from LDS feeds 4 MFMAs instead of 2 (the LDS
//     fragment traffic per flop halves), 64 MFMAs per key tile and wave;
//   * the 257th KEY is folded in on the VALU after the tile loop (as above); the 257th QUERY ROW is computed on the VALU too, split
//     over the waves: wave w scores it against the 64 keys of tile w (one key per lane, q . k on v_dot2 from LDS), forms its own
//     (max, sum, sum p v[d = lane]) and parks them in LDS; wave 0 merges the NWM partials at the end.  No wave is spent on one row.
// 4 waves x ~200 registers: two workgroups per CU (LDS 2 x 66 KiB), 8 waves -- the second workgroup's MFMAs run under the first
// one's softmax.
// ABL != 0: timing ablations for tools/bench_kernels.py (VCLA_ATTN_VIT_ABL; results are garbage): 1 = no last-row VALU phase, 2 = no v_exp
// in the softmax, 3 = no key-tile loop at all (staging + epilogue only), 4 = tile loop without the softmax VALU work.
template <int NWM, int ABL = 0>
__global__ __launch_bounds__(NWM * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_vit_kernel(vcla_attn_args a) {
    constexpr int D = 64, KST = 2, DT = 4, CH = 8, QT = 4, NT = NWM * 64, NK = NWM * 64 + 1;
    constexpr int KV_BYTES = NK * D * 2;                              // one operand image: [NK keys][128 B], chunk-swizzled
    constexpr int NLD = (NK * CH + NT - 1) / NT;                      // staging loads per thread per operand
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];    // [K image][V image][NWM x 66 floats: last-row partials]
    unsigned char* ks_all = lds_all;
    unsigned char* vs_all = lds_all + KV_BYTES;
    float* part = reinterpret_cast<float*>(lds_all + 2 * KV_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y;
    const bf16_t* qb = (const bf16_t*)a.q + b * a.q_bs + h * a.q_hs;
    const bf16_t* kb = (const bf16_t*)a.k + b * a.k_bs + h * a.k_hs;
    const bf16_t* vb = (const bf16_t*)a.v + b * a.v_bs + h * a.v_hs;
    bf16_t* ob = (bf16_t*)a.o + b * a.o_bs + h * a.o_hs;
    const int ql = lane & 15, g = lane >> 4;
    const int qw = wave * 64;
    const float sl2 = a.scale * 1.44269504088896340736f;

    // ---- request everything: the wave's Q fragments, the last query row (every lane the whole row: broadcast loads), all of K and V
    bf16x8_t qf[QT][KST];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int s = 0; s < KST; ++s)
            qf[qt][s] = *reinterpret_cast<const bf16x8_t*>(qb + (int64_t)(qw + qt * 16 + ql) * a.q_rs + s * 32 + g * 8);
    bf16x8_t qx[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) qx[c] = *reinterpret_cast<const bf16x8_t*>(qb + (int64_t)(NK - 1) * a.q_rs + c * 8);
    {
        u32x4_t rk[NLD], rv[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int id0 = i * NT + tid, id = id0 < NK * CH ? id0 : NK * CH - 1;
            const int key = id / CH, ch = id % CH;
            rk[i] = *reinterpret_cast<const u32x4_t*>(kb + (int64_t)key * a.k_rs + ch * 8);
            rv[i] = *reinterpret_cast<const u32x4_t*>(vb + (int64_t)key * a.v_rs + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int id = i * NT + tid, key = id / CH, ch = id % CH;
            if (id < NK * CH) {
                *reinterpret_cast<u32x4_t*>(ks_all + fa_k_off<D>(key, ch)) = rk[i];
                *reinterpret_cast<u32x4_t*>(vs_all + fa_v_off<D>(key, ch * 8)) = rv[i];
            }
        }
    }
    __syncthreads();

    // ---- last query row against this wave's 64 keys (wave 0: + the last key), on the VALU
    if constexpr (ABL != 1) {
        auto qdot = [&](int key) {
            float d_ = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const bf16x8_t kc = *reinterpret_cast<const bf16x8_t*>(ks_all + fa_k_off<D>(key, c));
                d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qx[c], qx[c], 0, 1), __builtin_shufflevector(kc, kc, 0, 1), d_, false);
                d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qx[c], qx[c], 2, 3), __builtin_shufflevector(kc, kc, 2, 3), d_, false);
                d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qx[c], qx[c], 4, 5), __builtin_shufflevector(kc, kc, 4, 5), d_, false);
                d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qx[c], qx[c], 6, 7), __builtin_shufflevector(kc, kc, 6, 7), d_, false);
            }
            return d_ * sl2;
        };
        const float sc = qdot(qw + lane);
        const float sx = qdot(NK - 1);                              // wave-uniform (broadcast LDS reads); counted by wave 0 only
        float mw = wave_max(sc);
        if (wave == 0) mw = fmaxf(mw, sx);
        const float p = __builtin_amdgcn_exp2f(sc - mw);
        const float px = wave == 0 ? __builtin_amdgcn_exp2f(sx - mw) : 0.f;
        const float lw = wave_sum(p) + px;
        // o_w[d = lane] = sum over the wave's keys of p_key * V[key][lane]
        float acc = px * bf2f(*reinterpret_cast<const bf16_t*>(vs_all + fa_v_off<D>(NK - 1, lane)));
#pragma unroll 16
        for (int kk = 0; kk < 64; ++kk) {
            const float pk = __shfl(p, kk, 64);
            acc = __builtin_fmaf(pk, bf2f(*reinterpret_cast<const bf16_t*>(vs_all + fa_v_off<D>(qw + kk, lane))), acc);
        }
        part[wave * 66 + lane] = acc;
        if (lane == 0) { part[wave * 66 + 64] = mw; part[wave * 66 + 65] = lw; }
    }

    // ---- the wave's 64 query rows against all key tiles
    f32x4_t o[QT][DT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { m_run[qt] = -INFINITY; l_run[qt] = 0.f; }

    for (int tile = 0; tile < (ABL == 3 ? 0 : NWM); ++tile) {
        const unsigned char* ks = ks_all + tile * (64 * D * 2);
        const auto vbase = (__attribute__((address_space(3))) unsigned char*)lds_all + KV_BYTES + tile * (64 * D * 2);
        f32x4_t sacc[QT][4];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int t = 0; t < 4; ++t) sacc[qt][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KST; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(ks + fa_k_off<D>(t * 16 + ql, s * 4 + g));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) sacc[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][s], sacc[qt][t], 0, 0, 0);
            }
        bf16x8_t pf[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[qt][t][r]);
            mx *= sl2;
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[qt], mx);                 // finite: nothing is masked here
            const float alpha = exp2f(m_run[qt] - m_new);             // exp2(-inf) = 0 on the first tile
            m_run[qt] = m_new;
            float ps = 0.f;
            float pv[16];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p;
                    if constexpr (ABL == 2) p = __builtin_fmaf(sacc[qt][t][r], sl2, -m_new);
                    else if constexpr (ABL == 4) p = sacc[qt][t][r];
                    else p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[qt][t][r], sl2, -m_new));
                    pv[t * 4 + r] = p;
                    if constexpr (ABL != 4) ps += p;
                }
            l_run[qt] = l_run[qt] * alpha + ps;
            if constexpr (ABL != 4) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) o[qt][dt] *= alpha;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 u;
                u.x = pack_bf2(pv[(2 * s) * 4 + 0], pv[(2 * s) * 4 + 1]);
                u.y = pack_bf2(pv[(2 * s) * 4 + 2], pv[(2 * s) * 4 + 3]);
                u.z = pack_bf2(pv[(2 * s + 1) * 4 + 0], pv[(2 * s + 1) * 4 + 1]);
                u.w = pack_bf2(pv[(2 * s + 1) * 4 + 2], pv[(2 * s + 1) * 4 + 3]);
                pf[qt][s] = __builtin_bit_cast(bf16x8_t, u);
            }
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int kr = g * 4 + (ql >> 2), dc = dt * 16 + (ql & 3) * 4;
                const fa_s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fa_lds_v4_t)(vbase + fa_v_off<D>((2 * s) * 16 + kr, dc)));
                const fa_s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fa_lds_v4_t)(vbase + fa_v_off<D>((2 * s + 1) * 16 + kr, dc)));
                const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt][s], o[qt][dt], 0, 0, 0);
            }
    }
    // ---- the last key (NK - 1) on the VALU, from LDS: lane (ql, g) holds Q[q][s*32 + g*8 .. +8] and O[q][dt*16 + g*4 .. +4]
    {
        bf16x8_t kx[KST];
#pragma unroll
        for (int s = 0; s < KST; ++s) kx[s] = *reinterpret_cast<const bf16x8_t*>(ks_all + fa_k_off<D>(NK - 1, s * 4 + g));
        uint2 vx[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vx[dt] = *reinterpret_cast<const uint2*>(vs_all + fa_v_off<D>(NK - 1, dt * 16 + g * 4));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float d_ = 0.f;
#pragma unroll
            for (int s = 0; s < KST; ++s) {
                d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qf[qt][s], qf[qt][s], 0, 1), __builtin_shufflevector(kx[s], kx[s], 0, 1), d_, false);
                d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qf[qt][s], qf[qt][s], 2, 3), __builtin_shufflevector(kx[s], kx[s], 2, 3), d_, false);
                d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qf[qt][s], qf[qt][s], 4, 5), __builtin_shufflevector(kx[s], kx[s], 4, 5), d_, false);
                d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qf[qt][s], qf[qt][s], 6, 7), __builtin_shufflevector(kx[s], kx[s], 6, 7), d_, false);
            }
            d_ += __shfl_xor(d_, 16, 64);
            d_ += __shfl_xor(d_, 32, 64);
            const float sx = d_ * sl2;
            const float m_new = fmaxf(m_run[qt], sx);
            const float alpha = exp2f(m_run[qt] - m_new), px = __builtin_amdgcn_exp2f(sx - m_new);
            m_run[qt] = m_new;
            l_run[qt] = l_run[qt] * alpha + (g == 0 ? px : 0.f);   // l is a per-lane partial, summed over g below
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                o[qt][dt][0] = __builtin_fmaf(px, __uint_as_float(vx[dt].x << 16), o[qt][dt][0] * alpha);
                o[qt][dt][1] = __builtin_fmaf(px, __uint_as_float(vx[dt].x & 0xffff0000u), o[qt][dt][1] * alpha);
                o[qt][dt][2] = __builtin_fmaf(px, __uint_as_float(vx[dt].y << 16), o[qt][dt][2] * alpha);
                o[qt][dt][3] = __builtin_fmaf(px, __uint_as_float(vx[dt].y & 0xffff0000u), o[qt][dt][3] * alpha);
            }
        }
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = l_run[qt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        bf16_t* orow = ob + (int64_t)(qw + qt * 16 + ql) * a.o_rs;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            float v[4] = {o[qt][dt][0] * inv, o[qt][dt][1] * inv, o[qt][dt][2] * inv, o[qt][dt][3] * inv};
            Act<bf16_t>::st4(orow + dt * 16 + g * 4, v);
        }
    }
    // ---- merge the NWM partials of the last query row
    __syncthreads();
    if (wave == 0) {
        float m = -INFINITY;
#pragma unroll
        for (int w = 0; w < NWM; ++w) m = fmaxf(m, part[w * 66 + 64]);
        float l = 0.f, acc = 0.f;
#pragma unroll
        for (int w = 0; w < NWM; ++w) {
            const float f = exp2f(part[w * 66 + 64] - m);
            l = __builtin_fmaf(part[w * 66 + 65], f, l);
            acc = __builtin_fmaf(part[w * 66 + lane], f, acc);
        }
        ob[(int64_t)(NK - 1) * a.o_rs + lane] = f2bf(acc / l);
    }
}

// ---- the same kernel with direct-to-LDS staging, pipelined against the key tiles (NWM = 4 only).
// The register-staged form above is lock-stepped: both co-resident workgroups of a CU request their 96 KiB at launch, wait ~25 us for them
// (135 MB per launch through L2: the staging + epilogue alone measure 33 of the 48 us, VCLA_ATTN_VIT_ABL=3), then compute.  Here every byte
// moves with global_load_lds_dwordx4 issued from inline asm (no staging registers, hipcc neither counts nor drains them) in the order
// the tile loop consumes them, and vmcnt is counted by hand: tile t is computed once the wave's own pieces of tile t have landed and the
// workgroup has met at a bare s_barrier, while the later tiles are still in flight.  LDS images are lane-linear per DMA instruction, so the
// bank swizzles of fa_k_off / fa_v_off are applied on the SOURCE side (the lane of slot p fetches the chunk that belongs there).
// The wave's own Q fragments are the one exception: eight inline-asm REGISTER loads ("=v"), issued first and completed by the same counted wait
// as key tile 0, whose statement names every destination "+v" (cdna_hip_programming.md section 5, form (ii)); hipcc treats the destinations as
// written at the load statement, so the shipped code object is audited for any use of them before the wait
// (tests/test_host_cpu.py::test_vit_attention_asm_register_loads_are_untouched_until_their_wait) and for an empty private segment -- a spill of one
// of them before the wait would store stale bits.  The last key / value / query rows ride in three padded 1 KiB pieces.
__device__ __forceinline__ void fa_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void fa_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void fa_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

#ifdef VCLA_G2_TIMELINE   // debug build only (make -C csrc timeline): per-workgroup phase stamps of the ViT attention, 100 MHz wall clock (tools/debug/vit_attn_timeline.py)
__device__ unsigned long long* fa_timeline = nullptr;      // [workgroup][16]
extern "C" int vcla_debug_set_vit_timeline(unsigned long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(fa_timeline), &p, sizeof(p)); }
#define FA_STAMP(i_) do { if (fa_timeline && threadIdx.x == 0) fa_timeline[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * 16 + (i_)] = wall_clock64(); } while (0)
#else
#define FA_STAMP(i_) do { } while (0)
#endif

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_vit_dma_kernel(vcla_attn_args a) {
    constexpr int NWM = 4, D = 64, KST = 2, DT = 4, QT = 4, NK = NWM * 64 + 1;
    constexpr int ROWS = NWM * 64 + 8;                                 // image rows incl. the padded piece of the last key
    constexpr int IMG = ROWS * 128;                                    // one operand image
    constexpr int TILE = 64 * 128;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_all[];    // [K image][V image][last query row piece 1 KiB][NWM x 68 floats]
    unsigned char* ks_all = lds_all;
    unsigned char* vs_all = lds_all + IMG;
    unsigned char* qx_row = lds_all + 2 * IMG;
    float* part = reinterpret_cast<float*>(lds_all + 2 * IMG + 1024);
    typedef __attribute__((address_space(3))) void* lds_p;
    const unsigned lds_u = (unsigned)(uintptr_t)(lds_p)lds_all;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y;
    const bf16_t* qb = (const bf16_t*)a.q + b * a.q_bs + h * a.q_hs;
    const bf16_t* kb = (const bf16_t*)a.k + b * a.k_bs + h * a.k_hs;
    const bf16_t* vb = (const bf16_t*)a.v + b * a.v_bs + h * a.v_hs;
    bf16_t* ob = (bf16_t*)a.o + b * a.o_bs + h * a.o_hs;
    const int ql = lane & 15, g = lane >> 4;
    const int qw = wave * 64;
    const float sl2 = a.scale * 1.44269504088896340736f;
    FA_STAMP(0);

    // ---- DMA lane maps: lane = (row within the 8-row piece, 16-byte slot p); the chunk that belongs in slot p of `row`
    const int prow = lane >> 3, pslot = lane & 7;
    auto k_chunk = [&](int row) { return pslot ^ ((row >> 1) & 7); };                                   // inverse of fa_k_off (an involution)
    auto v_chunk = [&](int row) { return ((((pslot >> 1) ^ (row >> 1)) & 3) << 1) + (pslot & 1); };    // inverse of fa_v_off's 32-byte block swizzle
    // the wave's Q fragments (B port: lane (q = ql, g) holds Q[q][ks*32 + g*8 .. +8]) as register loads hipcc does not see: issued first,
    // completed by the hand-counted wait in front of tile 0 (cdna_hip_programming.md section 5, form (ii): "=v" loads, then a wait statement
    // naming every destination "+v" before the first consumer; the .s is audited for compiler moves of those registers in between)
    u32x4_t qv[QT][KST];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int s2 = 0; s2 < KST; ++s2) {
            const bf16_t* qp = qb + (int64_t)(qw + qt * 16 + ql) * a.q_rs + s2 * 32 + g * 8;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qv[qt][s2]) : "v"(qp) : "memory");
        }
    {   // the last key's K / V rows and the last query row: three padded 8-row pieces (rows clamped to the last one), one per wave (wave 3 repeats wave 2's)
        const int row = NK - 1;
        if (wave == 0) fa_dma16(kb + (int64_t)row * a.k_rs + k_chunk(NK - 1 + prow) * 8, __builtin_amdgcn_readfirstlane(lds_u + (NK - 1) * 128));
        else if (wave == 1) fa_dma16(vb + (int64_t)row * a.v_rs + v_chunk(NK - 1 + prow) * 8, __builtin_amdgcn_readfirstlane(lds_u + IMG + (NK - 1) * 128));
        else fa_dma16(qb + (int64_t)row * a.q_rs + pslot * 8, __builtin_amdgcn_readfirstlane(lds_u + 2 * IMG));
    }
    // key tiles: 16 pieces per tile (8 K + 8 V), 4 per wave, every tile requested up front in consumption order
    auto issue_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wave * 2 + i;                  // 8 rows
            const int row = t * 64 + piece * 8 + prow;
            fa_dma16(kb + (int64_t)row * a.k_rs + k_chunk(row) * 8, __builtin_amdgcn_readfirstlane(lds_u + (t * 64 + piece * 8) * 128));
            fa_dma16(vb + (int64_t)row * a.v_rs + v_chunk(row) * 8, __builtin_amdgcn_readfirstlane(lds_u + IMG + (t * 64 + piece * 8) * 128));
        }
    };
    issue_tile(0);
    issue_tile(1);
    issue_tile(2);
    issue_tile(3);
    FA_STAMP(2);
    bf16x8_t qf[QT][KST];                   // filled from qv after the counted wait below

    f32x4_t o[QT][DT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { m_run[qt] = -INFINITY; l_run[qt] = 0.f; }

    auto tile_body = [&](int tile) {
        const unsigned char* ks = ks_all + tile * TILE;
        const auto vbase = (__attribute__((address_space(3))) unsigned char*)lds_all + IMG + tile * TILE;
        f32x4_t sacc[QT][4];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int t = 0; t < 4; ++t) sacc[qt][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KST; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(ks + fa_k_off<D>(t * 16 + ql, s * 4 + g));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) sacc[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][s], sacc[qt][t], 0, 0, 0);
            }
        bf16x8_t pf[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[qt][t][r]);
            mx *= sl2;
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[qt], mx);
            const float alpha = exp2f(m_run[qt] - m_new);
            m_run[qt] = m_new;
            float ps = 0.f;
            float pv[16];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[qt][t][r], sl2, -m_new));
                    pv[t * 4 + r] = p;
                    ps += p;
                }
            l_run[qt] = l_run[qt] * alpha + ps;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[qt][dt] *= alpha;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 u;
                u.x = pack_bf2(pv[(2 * s) * 4 + 0], pv[(2 * s) * 4 + 1]);
                u.y = pack_bf2(pv[(2 * s) * 4 + 2], pv[(2 * s) * 4 + 3]);
                u.z = pack_bf2(pv[(2 * s + 1) * 4 + 0], pv[(2 * s + 1) * 4 + 1]);
                u.w = pack_bf2(pv[(2 * s + 1) * 4 + 2], pv[(2 * s + 1) * 4 + 3]);
                pf[qt][s] = __builtin_bit_cast(bf16x8_t, u);
            }
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int kr = g * 4 + (ql >> 2), dc = dt * 16 + (ql & 3) * 4;
                const fa_s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fa_lds_v4_t)(vbase + fa_v_off<D>((2 * s) * 16 + kr, dc)));
                const fa_s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fa_lds_v4_t)(vbase + fa_v_off<D>((2 * s + 1) * 16 + kr, dc)));
                const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt][s], o[qt][dt], 0, 0, 0);
            }
    };
    // VMEM queue of a wave: [Q: 8][last rows: 1][tile 0: 4][tile 1: 4][tile 2: 4][tile 3: 4]; vmcnt retires in order, so "at most 12 outstanding"
    // means Q, the last rows and this wave's pieces of tile 0 have landed; every wave says so at the barrier
    asm volatile("s_waitcnt vmcnt(12)"
                 : "+v"(qv[0][0]), "+v"(qv[0][1]), "+v"(qv[1][0]), "+v"(qv[1][1]), "+v"(qv[2][0]), "+v"(qv[2][1]), "+v"(qv[3][0]), "+v"(qv[3][1])
                 :: "memory");
    FA_STAMP(1);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int s2 = 0; s2 < KST; ++s2) qf[qt][s2] = __builtin_bit_cast(bf16x8_t, qv[qt][s2]);
#pragma unroll 1
    for (int tile = 0; tile < NWM; ++tile) {
        if (tile == 1) fa_vmcnt<8>();
        else if (tile == 2) fa_vmcnt<4>();
        else if (tile == 3) fa_vmcnt<0>();
        fa_lds_barrier();
        if (tile == 0) FA_STAMP(3);
        tile_body(tile);
        FA_STAMP(4 + tile);
    }
    // ---- the last key (NK - 1) on the VALU, from LDS
    {
        bf16x8_t kx[KST];
#pragma unroll
        for (int s = 0; s < KST; ++s) kx[s] = *reinterpret_cast<const bf16x8_t*>(ks_all + fa_k_off<D>(NK - 1, s * 4 + g));
        uint2 vx[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vx[dt] = *reinterpret_cast<const uint2*>(vs_all + fa_v_off<D>(NK - 1, dt * 16 + g * 4));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float d_ = 0.f;
#pragma unroll
            for (int s = 0; s < KST; ++s) {
                d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qf[qt][s], qf[qt][s], 0, 1), __builtin_shufflevector(kx[s], kx[s], 0, 1), d_, false);
                d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qf[qt][s], qf[qt][s], 2, 3), __builtin_shufflevector(kx[s], kx[s], 2, 3), d_, false);
                d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qf[qt][s], qf[qt][s], 4, 5), __builtin_shufflevector(kx[s], kx[s], 4, 5), d_, false);
                d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qf[qt][s], qf[qt][s], 6, 7), __builtin_shufflevector(kx[s], kx[s], 6, 7), d_, false);
            }
            d_ += __shfl_xor(d_, 16, 64);
            d_ += __shfl_xor(d_, 32, 64);
            const float sx = d_ * sl2;
            const float m_new = fmaxf(m_run[qt], sx);
            const float alpha = exp2f(m_run[qt] - m_new), px = __builtin_amdgcn_exp2f(sx - m_new);
            m_run[qt] = m_new;
            l_run[qt] = l_run[qt] * alpha + (g == 0 ? px : 0.f);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                o[qt][dt][0] = __builtin_fmaf(px, __uint_as_float(vx[dt].x << 16), o[qt][dt][0] * alpha);
                o[qt][dt][1] = __builtin_fmaf(px, __uint_as_float(vx[dt].x & 0xffff0000u), o[qt][dt][1] * alpha);
                o[qt][dt][2] = __builtin_fmaf(px, __uint_as_float(vx[dt].y << 16), o[qt][dt][2] * alpha);
                o[qt][dt][3] = __builtin_fmaf(px, __uint_as_float(vx[dt].y & 0xffff0000u), o[qt][dt][3] * alpha);
            }
        }
    }
    // normalise O and pack it to bf16 now (32 registers across the last-row phase); it leaves through LDS as whole rows after the barrier below
    uint2 opk[QT][DT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = l_run[qt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            opk[qt][dt] = make_uint2(pack_bf2(o[qt][dt][0] * inv, o[qt][dt][1] * inv), pack_bf2(o[qt][dt][2] * inv, o[qt][dt][3] * inv));
    }
    FA_STAMP(8);
    // ---- the 257th query row against this wave's key tile, on the MFMAs: one 16-row q-tile whose every row IS the last row (row 0 is read
    // out), 8 + 8 MFMAs on fragments this wave has just used -- the VALU form of the register-staged kernel above (a dot product per lane, 64
    // readlane + LDS steps for P V) took 2.4 - 3.3 us of a 17 - 24 us workgroup (profiles/r04_vit_attn_timeline.txt).  Wave 0 adds the last key.
    {
        const unsigned char* ks = ks_all + wave * TILE;
        const auto vbase = (__attribute__((address_space(3))) unsigned char*)lds_all + IMG + wave * TILE;
        bf16x8_t qfx[KST];
#pragma unroll
        for (int s2 = 0; s2 < KST; ++s2) qfx[s2] = *reinterpret_cast<const bf16x8_t*>(qx_row + (s2 * 4 + g) * 16);
        f32x4_t sx[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) sx[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < KST; ++s2)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(ks + fa_k_off<D>(t * 16 + ql, s2 * 4 + g));
                sx[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qfx[s2], sx[t], 0, 0, 0);
            }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sx[t][r]);
        mx *= sl2;
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sk = 0.f;                                   // wave 0: the last query row against the last key
        if (wave == 0) {
#pragma unroll
            for (int s2 = 0; s2 < KST; ++s2) {
                const bf16x8_t kx = *reinterpret_cast<const bf16x8_t*>(ks_all + fa_k_off<D>(NK - 1, s2 * 4 + g));
                sk = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qfx[s2], qfx[s2], 0, 1), __builtin_shufflevector(kx, kx, 0, 1), sk, false);
                sk = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qfx[s2], qfx[s2], 2, 3), __builtin_shufflevector(kx, kx, 2, 3), sk, false);
                sk = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qfx[s2], qfx[s2], 4, 5), __builtin_shufflevector(kx, kx, 4, 5), sk, false);
                sk = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qfx[s2], qfx[s2], 6, 7), __builtin_shufflevector(kx, kx, 6, 7), sk, false);
            }
            sk += __shfl_xor(sk, 16, 64);
            sk += __shfl_xor(sk, 32, 64);
            sk *= sl2;
            mx = fmaxf(mx, sk);
        }
        float ps = 0.f;
        float pv[16];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pp = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[t][r], sl2, -mx));
                pv[t * 4 + r] = pp;
                ps += pp;
            }
        ps += __shfl_xor(ps, 16, 64);
        ps += __shfl_xor(ps, 32, 64);
        bf16x8_t pfx[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            uint4 u;
            u.x = pack_bf2(pv[(2 * s2) * 4 + 0], pv[(2 * s2) * 4 + 1]);
            u.y = pack_bf2(pv[(2 * s2) * 4 + 2], pv[(2 * s2) * 4 + 3]);
            u.z = pack_bf2(pv[(2 * s2 + 1) * 4 + 0], pv[(2 * s2 + 1) * 4 + 1]);
            u.w = pack_bf2(pv[(2 * s2 + 1) * 4 + 2], pv[(2 * s2 + 1) * 4 + 3]);
            pfx[s2] = __builtin_bit_cast(bf16x8_t, u);
        }
        f32x4_t ox[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            ox[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int kr = g * 4 + (ql >> 2), dc = dt * 16 + (ql & 3) * 4;
                const fa_s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fa_lds_v4_t)(vbase + fa_v_off<D>((2 * s2) * 16 + kr, dc)));
                const fa_s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fa_lds_v4_t)(vbase + fa_v_off<D>((2 * s2 + 1) * 16 + kr, dc)));
                const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
                ox[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pfx[s2], ox[dt], 0, 0, 0);
            }
        }
        if (wave == 0) {                                  // fold the last key into wave 0's partial
            const float pk = __builtin_amdgcn_exp2f(sk - mx);
            ps += pk;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const uint2 vx = *reinterpret_cast<const uint2*>(vs_all + fa_v_off<D>(NK - 1, dt * 16 + g * 4));
                ox[dt][0] = __builtin_fmaf(pk, __uint_as_float(vx.x << 16), ox[dt][0]);
                ox[dt][1] = __builtin_fmaf(pk, __uint_as_float(vx.x & 0xffff0000u), ox[dt][1]);
                ox[dt][2] = __builtin_fmaf(pk, __uint_as_float(vx.y << 16), ox[dt][2]);
                ox[dt][3] = __builtin_fmaf(pk, __uint_as_float(vx.y & 0xffff0000u), ox[dt][3]);
            }
        }
        if (ql == 0) {                                    // lanes (0, g) hold row 0: d = dt*16 + g*4 + r
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4_t*>(part + wave * 68 + dt * 16 + g * 4) = ox[dt];
            if (g == 0) { part[wave * 68 + 64] = mx; part[wave * 68 + 65] = ps; }
        }
    }
    FA_STAMP(9);
    __syncthreads();                      // every wave is done with K / V: the images are free
    FA_STAMP(10);
    // ---- O leaves as WHOLE ROWS: a lane of the MFMA layout owns 8 bytes of 16 different rows -- 16 dwordx2 stores per lane, each wave
    // instruction touching 16 rows (the store tail of such an epilogue is issue-bound, MI355X_MICROARCH "attention epilogue store tail").
    // Staged through the wave's own 8 KiB of the K image (chunk-swizzled: conflict-free 8-byte writes up to 2-way), then 8 dwordx4 stores
    // per lane, every wave instruction = 8 full 128-byte rows.
    {
        unsigned char* os = ks_all + wave * TILE;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int row = qt * 16 + ql, c = dt * 2 + (g >> 1);
                *reinterpret_cast<uint2*>(os + row * 128 + ((c ^ (row & 7)) << 4) + (g & 1) * 8) = opk[qt][dt];
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = i * 8 + prow;
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(os + row * 128 + ((pslot ^ (row & 7)) << 4));
            *reinterpret_cast<u32x4_t*>(ob + (int64_t)(qw + row) * a.o_rs + pslot * 8) = v;
        }
    }
    if (wave == 0) {
        float m = -INFINITY;
#pragma unroll
        for (int w = 0; w < NWM; ++w) m = fmaxf(m, part[w * 68 + 64]);
        float l = 0.f, acc = 0.f;
#pragma unroll
        for (int w = 0; w < NWM; ++w) {
            const float f = exp2f(part[w * 68 + 64] - m);
            l = __builtin_fmaf(part[w * 68 + 65], f, l);
            acc = __builtin_fmaf(part[w * 68 + lane], f, acc);
        }
        ob[(int64_t)(NK - 1) * a.o_rs + lane] = f2bf(acc / l);
    }
    FA_STAMP(11);
}

// =================================================================== whole-sequence ViT attention for 336-px images: 577 = 9 x 64 + 1 tokens, d = 64
// (BASELINE configs[4]; hf clip/modeling_clip.py:259-335).  K / V of one (image, head) are 148 KB: they fit ONE workgroup's LDS (160 KiB per
// CU) but not two, so the 257-token kernel's geometry (4 waves x 64 query rows, two workgroups per CU) does not transfer.  Here ONE 8-wave
// workgroup owns the (image, head): the whole K / V sequence is parked in LDS once (all 9 key tiles requested up front with LDS-DMA in
// consumption order, hand-counted vmcnt, one bare barrier per tile as in attn_vit_dma_kernel) and the 577 query rows are 37 sixteen-row
// MFMA q-tiles -- 36 real ones plus one whose every row is the LAST row (row 0 of it is stored) -- taken in two passes over the resident
// image: pass 1 = 8 waves x 4 q-tiles (rows 0 .. 511) while the tiles land, pass 2 = waves 0 .. 4 x 1 q-tile (rows 512 .. 575 and the last
// row), no memory wait left in it.  The 577th KEY is folded in on the VALU from LDS in both passes.  The tile-by-tile kernel this replaces
// staged every key tile three times per head (three 9-wave workgroups of 192 rows) behind two barriers each: 123.7 us per layer at B = 32
// (14 % of the MFMA peak, profiles/r04_bench_fp8_336px_b32_by_grid.txt).
template <int QT>
__device__ __forceinline__ void vit_key_tile(const unsigned char* ks, __attribute__((address_space(3))) unsigned char* vbase, const bf16x8_t (&qf)[QT][2],
                                             f32x4_t (&o)[QT][4], float (&m_run)[QT], float (&l_run)[QT], float sl2, int ql, int g) {
    constexpr int D = 64, KST = 2, DT = 4;
    f32x4_t sacc[QT][4];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int t = 0; t < 4; ++t) sacc[qt][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KST; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(ks + fa_k_off<D>(t * 16 + ql, s * 4 + g));
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) sacc[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][s], sacc[qt][t], 0, 0, 0);
        }
    bf16x8_t pf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[qt][t][r]);
        mx *= sl2;
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run[qt], mx);
        const float alpha = exp2f(m_run[qt] - m_new);
        m_run[qt] = m_new;
        float ps = 0.f;
        float pv[16];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[qt][t][r], sl2, -m_new));
                pv[t * 4 + r] = p;
                ps += p;
            }
        l_run[qt] = l_run[qt] * alpha + ps;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qt][dt] *= alpha;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 u;
            u.x = pack_bf2(pv[(2 * s) * 4 + 0], pv[(2 * s) * 4 + 1]);
            u.y = pack_bf2(pv[(2 * s) * 4 + 2], pv[(2 * s) * 4 + 3]);
            u.z = pack_bf2(pv[(2 * s + 1) * 4 + 0], pv[(2 * s + 1) * 4 + 1]);
            u.w = pack_bf2(pv[(2 * s + 1) * 4 + 2], pv[(2 * s + 1) * 4 + 3]);
            pf[qt][s] = __builtin_bit_cast(bf16x8_t, u);
        }
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int kr = g * 4 + (ql >> 2), dc = dt * 16 + (ql & 3) * 4;
            const fa_s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fa_lds_v4_t)(vbase + fa_v_off<D>((2 * s) * 16 + kr, dc)));
            const fa_s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fa_lds_v4_t)(vbase + fa_v_off<D>((2 * s + 1) * 16 + kr, dc)));
            const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt][s], o[qt][dt], 0, 0, 0);
        }
}

// the last key (one more than a whole number of tiles) on the VALU, from LDS row `key` of the K / V images
template <int QT>
__device__ __forceinline__ void vit_last_key(const unsigned char* ks_all, const unsigned char* vs_all, int key, const bf16x8_t (&qf)[QT][2], f32x4_t (&o)[QT][4],
                                             float (&m_run)[QT], float (&l_run)[QT], float sl2, int g) {
    constexpr int D = 64, KST = 2, DT = 4;
    bf16x8_t kx[KST];
#pragma unroll
    for (int s = 0; s < KST; ++s) kx[s] = *reinterpret_cast<const bf16x8_t*>(ks_all + fa_k_off<D>(key, s * 4 + g));
    uint2 vx[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vx[dt] = *reinterpret_cast<const uint2*>(vs_all + fa_v_off<D>(key, dt * 16 + g * 4));
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float d_ = 0.f;
#pragma unroll
        for (int s = 0; s < KST; ++s) {
            d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qf[qt][s], qf[qt][s], 0, 1), __builtin_shufflevector(kx[s], kx[s], 0, 1), d_, false);
            d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qf[qt][s], qf[qt][s], 2, 3), __builtin_shufflevector(kx[s], kx[s], 2, 3), d_, false);
            d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qf[qt][s], qf[qt][s], 4, 5), __builtin_shufflevector(kx[s], kx[s], 4, 5), d_, false);
            d_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(qf[qt][s], qf[qt][s], 6, 7), __builtin_shufflevector(kx[s], kx[s], 6, 7), d_, false);
        }
        d_ += __shfl_xor(d_, 16, 64);
        d_ += __shfl_xor(d_, 32, 64);
        const float sx = d_ * sl2;
        const float m_new = fmaxf(m_run[qt], sx);
        const float alpha = exp2f(m_run[qt] - m_new), px = __builtin_amdgcn_exp2f(sx - m_new);
        m_run[qt] = m_new;
        l_run[qt] = l_run[qt] * alpha + (g == 0 ? px : 0.f);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            o[qt][dt][0] = __builtin_fmaf(px, __uint_as_float(vx[dt].x << 16), o[qt][dt][0] * alpha);
            o[qt][dt][1] = __builtin_fmaf(px, __uint_as_float(vx[dt].x & 0xffff0000u), o[qt][dt][1] * alpha);
            o[qt][dt][2] = __builtin_fmaf(px, __uint_as_float(vx[dt].y << 16), o[qt][dt][2] * alpha);
            o[qt][dt][3] = __builtin_fmaf(px, __uint_as_float(vx[dt].y & 0xffff0000u), o[qt][dt][3] * alpha);
        }
    }
}

template <int NT>      // key tiles of 64; the sequence is NT * 64 + 1 tokens (NT = 9: 577)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_vit_long_kernel(vcla_attn_args a) {
    constexpr int KST = 2, DT = 4, QT = 4, NW = 8, NK = NT * 64 + 1;
    constexpr int ROWS = NT * 64 + 8;                                   // image rows incl. the padded piece of the last key
    constexpr int IMG = ROWS * 128;                                     // one operand image
    constexpr int TILE = 64 * 128;
    static_assert(NT * 64 >= NW * 64 && 2 * IMG <= 160 * 1024, "geometry");
    static_assert(2 * (NT - 1) + 1 + 8 <= 63, "vmcnt range");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_all[];    // [K image][V image]
    unsigned char* ks_all = lds_all;
    unsigned char* vs_all = lds_all + IMG;
    typedef __attribute__((address_space(3))) void* lds_p;
    const unsigned lds_u = (unsigned)(uintptr_t)(lds_p)lds_all;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y;
    const bf16_t* qb = (const bf16_t*)a.q + b * a.q_bs + h * a.q_hs;
    const bf16_t* kb = (const bf16_t*)a.k + b * a.k_bs + h * a.k_hs;
    const bf16_t* vb = (const bf16_t*)a.v + b * a.v_bs + h * a.v_hs;
    bf16_t* ob = (bf16_t*)a.o + b * a.o_bs + h * a.o_hs;
    const int ql = lane & 15, g = lane >> 4;
    const int qw = wave * 64;                                            // pass 1: this wave's 64 query rows
    const float sl2 = a.scale * 1.44269504088896340736f;

    // ---- DMA lane maps (as attn_vit_dma_kernel): lane = (row within the 8-row piece, 16-byte slot p); the chunk that belongs in slot p of `row`
    const int prow = lane >> 3, pslot = lane & 7;
    auto k_chunk = [&](int row) { return pslot ^ ((row >> 1) & 7); };
    auto v_chunk = [&](int row) { return ((((pslot >> 1) ^ (row >> 1)) & 3) << 1) + (pslot & 1); };
    // pass-1 Q fragments: register loads hipcc does not see, completed by the counted wait in front of key tile 0 (form (ii) of the guide)
    u32x4_t qv[QT][KST];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int s2 = 0; s2 < KST; ++s2) {
            const bf16_t* qp = qb + (int64_t)(qw + qt * 16 + ql) * a.q_rs + s2 * 32 + g * 8;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qv[qt][s2]) : "v"(qp) : "memory");
        }
    // key tiles in consumption order: 16 pieces per tile (8 K + 8 V), one K and one V piece per wave
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int row = t * 64 + wave * 8 + prow;
        fa_dma16(kb + (int64_t)row * a.k_rs + k_chunk(row) * 8, __builtin_amdgcn_readfirstlane(lds_u + (t * 64 + wave * 8) * 128));
        fa_dma16(vb + (int64_t)row * a.v_rs + v_chunk(row) * 8, __builtin_amdgcn_readfirstlane(lds_u + IMG + (t * 64 + wave * 8) * 128));
    }
    // the last key's K / V rows: one padded 8-row piece each (rows clamped to the last one); EVERY wave issues one (even waves K, odd waves V:
    // identical bytes to the same place) so that the VMEM queue has the same length in every wave: [Q: 8][tile 0: 2] ... [tile NT-1: 2][last: 1]
    if ((wave & 1) == 0) fa_dma16(kb + (int64_t)(NK - 1) * a.k_rs + k_chunk(NK - 1 + prow) * 8, __builtin_amdgcn_readfirstlane(lds_u + (NK - 1) * 128));
    else fa_dma16(vb + (int64_t)(NK - 1) * a.v_rs + v_chunk(NK - 1 + prow) * 8, __builtin_amdgcn_readfirstlane(lds_u + IMG + (NK - 1) * 128));

    bf16x8_t qf[QT][KST];
    f32x4_t o[QT][DT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { m_run[qt] = -INFINITY; l_run[qt] = 0.f; }

    // vmcnt retires in order: "at most 2 (NT - 1) + 1 outstanding" = Q and this wave's pieces of tile 0 have landed; every wave says so at the barrier
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(qv[0][0]), "+v"(qv[0][1]), "+v"(qv[1][0]), "+v"(qv[1][1]), "+v"(qv[2][0]), "+v"(qv[2][1]), "+v"(qv[3][0]), "+v"(qv[3][1])
                 : "n"(2 * (NT - 1) + 1) : "memory");
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int s2 = 0; s2 < KST; ++s2) qf[qt][s2] = __builtin_bit_cast(bf16x8_t, qv[qt][s2]);
    // ---- pass 1: rows wave * 64 .. + 63 over the key tiles as they land
#pragma unroll 1
    for (int tile = 0; tile < NT; ++tile) {
        // tile t is ready once at most 2 (NT - 1 - t) + 1 younger instructions are outstanding; the wait for tile 0 was taken above
        switch (NT - 1 - tile) {
            case 0: fa_vmcnt<1>(); break;
            case 1: fa_vmcnt<3>(); break;
            case 2: fa_vmcnt<5>(); break;
            case 3: fa_vmcnt<7>(); break;
            case 4: fa_vmcnt<9>(); break;
            case 5: fa_vmcnt<11>(); break;
            case 6: fa_vmcnt<13>(); break;
            case 7: fa_vmcnt<15>(); break;
            default: fa_vmcnt<2 * (NT - 1) + 1>(); break;
        }
        fa_lds_barrier();
        vit_key_tile<QT>(ks_all + tile * TILE, (__attribute__((address_space(3))) unsigned char*)lds_all + IMG + tile * TILE, qf, o, m_run, l_run, sl2, ql, g);
    }
    fa_vmcnt<0>();
    fa_lds_barrier();                                                    // the last key's rows have landed for everyone
    vit_last_key<QT>(ks_all, vs_all, NK - 1, qf, o, m_run, l_run, sl2, g);
    // normalise O and pack it to bf16 (32 registers across pass 2); it leaves through LDS as whole rows once K / V are no longer needed
    uint2 opk[QT][DT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = l_run[qt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            opk[qt][dt] = make_uint2(pack_bf2(o[qt][dt][0] * inv, o[qt][dt][1] * inv), pack_bf2(o[qt][dt][2] * inv, o[qt][dt][3] * inv));
    }
    // ---- pass 2: the remaining (NT - NW) * 64 rows as 16-row q-tiles on waves 0 .., then ONE q-tile whose every row is the last row.
    // K / V are resident: no wait, no barrier.  (NT = 9: waves 0 - 3 take rows 512 .. 575, wave 4 the last row; waves 5 - 7 are done.)
    constexpr int Q2 = (NT - NW) * 4;                                     // 16-row q-tiles left after pass 1
    static_assert(Q2 + 1 <= NW, "pass 2 is one q-tile per wave");
    if (wave <= Q2) {
        const int row2 = wave < Q2 ? NW * 64 + wave * 16 + ql : NK - 1;
        bf16x8_t qf2[1][KST];
#pragma unroll
        for (int s2 = 0; s2 < KST; ++s2) qf2[0][s2] = *reinterpret_cast<const bf16x8_t*>(qb + (int64_t)row2 * a.q_rs + s2 * 32 + g * 8);
        f32x4_t o2[1][DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o2[0][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        float m2[1] = {-INFINITY}, l2[1] = {0.f};
#pragma unroll 1
        for (int tile = 0; tile < NT; ++tile)
            vit_key_tile<1>(ks_all + tile * TILE, (__attribute__((address_space(3))) unsigned char*)lds_all + IMG + tile * TILE, qf2, o2, m2, l2, sl2, ql, g);
        vit_last_key<1>(ks_all, vs_all, NK - 1, qf2, o2, m2, l2, sl2, g);
        float l = l2[0];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        if (wave < Q2 || ql == 0) {                                      // the last-row tile: lanes (0, g) hold row 0 = the row itself
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                *reinterpret_cast<uint2*>(ob + (int64_t)row2 * a.o_rs + dt * 16 + g * 4) =
                    make_uint2(pack_bf2(o2[0][dt][0] * inv, o2[0][dt][1] * inv), pack_bf2(o2[0][dt][2] * inv, o2[0][dt][3] * inv));
        }
    }
    __syncthreads();                      // every wave is done with K / V: the images are free
    // ---- pass-1 O leaves as WHOLE ROWS through the wave's own 8 KiB of the K image (as attn_vit_dma_kernel): 8 dwordx4 stores per lane
    {
        unsigned char* os = ks_all + wave * TILE;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int row = qt * 16 + ql, c = dt * 2 + (g >> 1);
                *reinterpret_cast<uint2*>(os + row * 128 + ((c ^ (row & 7)) << 4) + (g & 1) * 8) = opk[qt][dt];
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = i * 8 + prow;
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(os + row * 128 + ((pslot ^ (row & 7)) << 4));
            *reinterpret_cast<u32x4_t*>(ob + (int64_t)(qw + row) * a.o_rs + pslot * 8) = v;
        }
    }
}

static bool attn_vit_shape(const vcla_attn_args* a) {
    if (a->causal || a->key_mask || a->tk_dev || a->D != 64 || a->Tq != a->Tk) return false;
    if (a->Tq == 577) return vcla_aligned(a->o, 16) && a->o_bs % 8 == 0 && a->o_hs % 8 == 0 && a->o_rs % 8 == 0;     // 336 px: attn_vit_long_kernel (16-byte O pieces)
    return a->Tq == 65 || a->Tq == 257;      // NWM = 1, 4 (a 2-wave instance spills its 72 staging registers; 129 tokens is no ViT geometry here)
}

int vcla_attention_vit(const vcla_attn_args* a, void* stream) {
    VCLA_REQUIRE(attn_vit_shape(a) && vcla_attention_mfma_supported(a) && vcla_aligned(a->o, 2), VCLA_ERR_BAD_ARG,
                 "attention: the whole-sequence ViT kernel needs an unmasked bidirectional d = 64 self-attention over 65 / 257 / 577 tokens");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(1, a->H, a->B);
    if (a->Tq == 577) {      // 336 px (BASELINE configs[4]): one 8-wave workgroup per (image, head), the whole 148 KB of K / V in LDS
        auto kern = attn_vit_long_kernel<9>;
        const size_t lds = (size_t)2 * (9 * 64 + 8) * 128;
        static bool attr_long[VCLA_MAX_DEVICES] = {};
        const int rc_ = vcla_raise_dyn_lds((const void*)kern, lds, attr_long);
        if (rc_) return rc_;
        kern<<<grid, 512, lds, s>>>(*a);
        VCLA_CHECK_LAUNCH("attn_vit_long_kernel");
        return VCLA_OK;
    }
    static const int abl_env = getenv("VCLA_ATTN_VIT_ABL") ? atoi(getenv("VCLA_ATTN_VIT_ABL")) : 0;
#define VIT_GO(NWM_)                                                                                                 \
    {                                                                                                                \
        auto kern = attn_vit_kernel<NWM_>;                                                                           \
        if ((NWM_) == 4 && abl_env == 1) kern = attn_vit_kernel<4, 1>;                                              \
        if ((NWM_) == 4 && abl_env == 2) kern = attn_vit_kernel<4, 2>;                                              \
        if ((NWM_) == 4 && abl_env == 3) kern = attn_vit_kernel<4, 3>;                                              \
        if ((NWM_) == 4 && abl_env == 4) kern = attn_vit_kernel<4, 4>;                                              \
        const size_t lds = (size_t)2 * ((NWM_) * 64 + 1) * 128 + (size_t)(NWM_) * 66 * 4;                             \
        static bool attr_set[VCLA_MAX_DEVICES] = {};                                                                 \
        if (lds > 64 * 1024) { const int rc_ = vcla_raise_dyn_lds((const void*)kern, lds, attr_set); if (rc_) return rc_; } \
        kern<<<grid, (NWM_) * 64, lds, s>>>(*a);                                                                     \
    }
    static const int vit_form = getenv("VCLA_ATTN_VIT") ? atoi(getenv("VCLA_ATTN_VIT")) : 2;   // 1 = register-staged form, 2 = direct-to-LDS pipelined form (257 tokens)
    const bool o16 = vcla_aligned(a->o, 16) && a->o_bs % 8 == 0 && a->o_hs % 8 == 0 && a->o_rs % 8 == 0;     // the DMA form stores O in 16-byte pieces
    if (a->Tq == 257 && vit_form != 1 && abl_env == 0 && o16) {
        auto kern = attn_vit_dma_kernel;
        const size_t lds = (size_t)2 * 264 * 128 + 1024 + 4 * 68 * 4;
        static bool attr_set[VCLA_MAX_DEVICES] = {};
        const int rc_ = vcla_raise_dyn_lds((const void*)kern, lds, attr_set);
        if (rc_) return rc_;
        kern<<<grid, 256, lds, s>>>(*a);
    } else if (a->Tq == 257) VIT_GO(4)
    else VIT_GO(1)
#undef VIT_GO
    VCLA_CHECK_LAUNCH("attn_vit_kernel");
    return VCLA_OK;
}

bool vcla_attention_mfma_supported(const vcla_attn_args* a) {
    if (a->D != 64 && a->D != 128) return false;
    if (a->tk_dev) return false;           // decode (Tq = 1) stays on the generic kernel
    if (a->Tq < 16) return false;
    // 16-byte row alignment for Q/K/V fragment + staging loads, 8-byte for O stores
    auto ok16 = [](const void* p, int64_t bs, int64_t hs, int64_t rs) {
        return vcla_aligned(p, 16) && bs % 8 == 0 && hs % 8 == 0 && rs % 8 == 0;
    };
    if (!ok16(a->q, a->q_bs, a->q_hs, a->q_rs) || !ok16(a->k, a->k_bs, a->k_hs, a->k_rs) || !ok16(a->v, a->v_bs, a->v_hs, a->v_rs)) return false;
    if (!vcla_aligned(a->o, 8) || a->o_bs % 4 || a->o_hs % 4 || a->o_rs % 4) return false;
    return true;
}

int vcla_attention_mfma(const vcla_attn_args* a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    static const int nw_env = getenv("VCLA_ATTN_MFMA_NW") ? atoi(getenv("VCLA_ATTN_MFMA_NW")) : 0;   // A/B runs: force 4
    // ViT self-attention (65 / 257 / 577 tokens, d = 64) with at least half a round of workgroups: the whole-sequence kernels
    static const int vit_env = getenv("VCLA_ATTN_VIT") ? atoi(getenv("VCLA_ATTN_VIT")) : 1;           // A/B runs: 0 = the tile-by-tile kernel
    if (vit_env && a->force_kernel != 2 && attn_vit_shape(a) && (int64_t)a->B * a->H >= 128) return vcla_attention_vit(a, stream);
    // waves per workgroup: bidirectional sequences longer than one 128-row block go 288 rows at a time (ViT-L/14 224 px: the
    // whole sequence), 64-query cross attention (resampler) needs only 2 waves
    int nw = 4;
    // (the 9-wave form needs B * H >= one workgroup per CU to pay: a single image is 16 heads = 16 workgroups, 48 with 4 waves)
    if (!a->causal && a->D == 64) nw = a->Tq <= 64 ? 2 : ((a->Tq > 128 && (int64_t)a->B * a->H >= 256) ? 9 : 4);
    if (nw_env == 4) nw = 4;
    dim3 grid((a->Tq + nw * 32 - 1) / (nw * 32), a->H, a->B);
    // key tiles the tile loop will walk (the kernel peels 1 - 4 remainder keys of an unmasked bidirectional sequence)
    int kv = a->Tk;
    if (!a->causal && !a->key_mask && a->Tk > FA_KV && a->Tk % FA_KV >= 1 && a->Tk % FA_KV <= 4) kv -= a->Tk % FA_KV;
    const int ntiles = (kv + FA_KV - 1) / FA_KV;
    static const int whole_env = getenv("VCLA_ATTN_MFMA_WHOLE") ? atoi(getenv("VCLA_ATTN_MFMA_WHOLE")) : 1;   // A/B runs: 0 = always tile by tile
#define FA_GO(D_, NW_, MAXT_)                                                                                        \
    {                                                                                                                \
        auto kern = attn_mfma_kernel<D_, NW_, MAXT_>;                                                                \
        const size_t lds = (size_t)(2 * FA_KV * D_ * 2) * ((MAXT_) > 0 ? (MAXT_) : 1);                               \
        static bool attr_set[VCLA_MAX_DEVICES] = {};                                                                 \
        if (lds > 64 * 1024) { const int rc_ = vcla_raise_dyn_lds((const void*)kern, lds, attr_set); if (rc_) return rc_; } \
        kern<<<grid, (NW_) * 64, lds, s>>>(*a);                                                                      \
    }
    if (a->D == 128) {
        // causal prefill: a 128-token prompt is 2 tiles = 64 KiB (two 4-wave workgroups per CU): 75.0 -> 54.6 us per layer at B = 64.
        // (The d = 64 shapes LOSE in this form -- ViT 57.7 -> 62.1 us with 4 tiles / 64 KiB, resampler 21.3 -> 35.0 us with 5 tiles /
        // 80 KiB: three co-resident 9-wave workgroups hide more than one barrier-free workgroup pair does -- and stay tile by tile.)
        if (whole_env && ntiles <= 2 && (int64_t)a->B * a->H * grid.x >= 512) FA_GO(128, 4, 2)
        else FA_GO(128, 4, 0)
    } else if (nw == 9) {
        FA_GO(64, 9, 0)
    } else if (nw == 2) {
        FA_GO(64, 2, 0)
    } else {
        FA_GO(64, 4, 0)
    }
#undef FA_GO
    VCLA_CHECK_LAUNCH("attn_mfma_kernel");
    return VCLA_OK;
}
