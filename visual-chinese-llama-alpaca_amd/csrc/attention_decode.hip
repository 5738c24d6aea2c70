// attention_decode.hip -- one decode step of LLaMA attention for every (sequence, head), fused:
//   RoPE(q), RoPE(k_new)  ->  append k_new / v_new to the KV cache  ->  softmax(q K^T / sqrt(d)) V over the cache.
// Replaces three launches (rope + append, attention) and the q/k round trip through HBM.  HBM-bound: the only
// large reads are the K and V rows of the cache (ctx * d * 2 elements per head), each read exactly once.
//
// Workgroup = one (b, h), 256 threads.  Scores: one key per thread (B = 1) or D/8 lanes per key row (batch decode), 16-byte
// row loads, probabilities in LDS; P V: D/8 adjacent lanes own one value row, partial outputs reduced by shuffles + LDS.
// Position / context length come from device memory (pos0 + *pos_dev) so the launch is hipGraph-replayable.
#include "vcla_common.h"
#include <stdlib.h>

template <typename T, int D> struct RowDot;
template <int D> struct RowDot<float, D> {
    __device__ static __forceinline__ float dot(const float* qs, const float* k) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            const float4 kv = *reinterpret_cast<const float4*>(k + c);
            acc += qs[c] * kv.x + qs[c + 1] * kv.y + qs[c + 2] * kv.z + qs[c + 3] * kv.w;
        }
        return acc;
    }
};
template <int D> struct RowDot<bf16_t, D> {
    __device__ static __forceinline__ float dot(const float* qs, const bf16_t* k) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < D; c += 8) {
            const uint4 kv = *reinterpret_cast<const uint4*>(k + c);
            float f[8];
            bf8_to_f32(kv, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += qs[c + e] * f[e];
        }
        return acc;
    }
};

template <typename T> __device__ __forceinline__ void load8(const T* p, float* v);
template <> __device__ __forceinline__ void load8<float>(const float* p, float* v) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float* v) { bf8_to_f32(*reinterpret_cast<const uint4*>(p), v); }

// 8 consecutive elements of a cache row, still in the storage dtype: lets the V rows be FETCHED before the softmax (the loads
// do not depend on the scores) and converted afterwards
template <typename T> struct Raw8;
template <> struct Raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
    __device__ __forceinline__ void zero() { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
    __device__ __forceinline__ void get(float* v) const { v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; }
};
template <> struct Raw8<bf16_t> {
    uint4 t;
    __device__ __forceinline__ void load(const bf16_t* p) { t = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void zero() { t = make_uint4(0u, 0u, 0u, 0u); }
    __device__ __forceinline__ void get(float* v) const { bf8_to_f32(t, v); }
};

// COOP = false: one key per thread (all of a head's K rows in flight after one instruction burst: best when the launch is a
// few dozen workgroups, B = 1).  COOP = true: D/8 adjacent lanes share a key row, so every load instruction of a wave
// covers 64/(D/8) whole rows = 8 full cache lines instead of 64 partial ones -- with thousands of workgroups (batch decode)
// the per-CU texture path, not latency, is what the one-key-per-thread form saturates.
// NW = waves per workgroup.  4 everywhere except the batch form at short contexts (NW = 2, see launch_decode): the batch launch
// is bounded by the dependent chain of a workgroup times the number of ROUNDS of workgroups, not by bandwidth; 2-wave
// workgroups fit 8 per CU, i.e. B * H = 2048 (b, h) pairs in ONE round instead of two.
template <typename T, int D, bool COOP, int NW = 4>
__global__ __launch_bounds__(NW * 64) void attn_decode_kernel(const T* __restrict__ qkv, T* __restrict__ kc, T* __restrict__ vc,
                                                          const float* __restrict__ cos_tab, const float* __restrict__ sin_tab,
                                                          T* __restrict__ out, int H, int ctx_max, int pos0,
                                                          const int32_t* __restrict__ pos_dev, const int32_t* __restrict__ key_mask,
                                                          int64_t key_mask_ld, float scale, int sc_cap, int out_frag_mt) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* qs = smem;              // [D]   roped query
    float* knew = smem + D;        // [D]   roped new key
    float* vnew = smem + 2 * D;    // [D]   new value
    float* red = smem + 3 * D;     // [2 * NW]
    float* part = smem + 3 * D + 2 * NW;  // [NW][D] partial outputs
    float* sc = part + NW * D;     // [sc_cap] scores / probabilities
    constexpr int NT = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const int pos = pos0 + (pos_dev ? *pos_dev : 0);   // position of the new token = number of cached keys
    constexpr int HALF = D / 2;
    const int HD = H * D;
    const T* row = qkv + (int64_t)b * 3 * HD;
    T* kbase = kc + ((int64_t)b * H + h) * ctx_max * D;
    T* vbase = vc + ((int64_t)b * H + h) * ctx_max * D;

    // ---- RoPE on q and k_new (rotate-half form), stage q/k/v in LDS, append k/v to the cache
    if (tid < HALF) {
        const float c = Act<T>::rnd(cos_tab[(int64_t)pos * HALF + tid]), s = Act<T>::rnd(sin_tab[(int64_t)pos * HALF + tid]);
        const float q0 = Act<T>::ld(row + h * D + tid), q1 = Act<T>::ld(row + h * D + tid + HALF);
        qs[tid] = Act<T>::rnd(q0 * c - q1 * s);
        qs[tid + HALF] = Act<T>::rnd(q1 * c + q0 * s);
        const float k0 = Act<T>::ld(row + HD + h * D + tid), k1 = Act<T>::ld(row + HD + h * D + tid + HALF);
        const float r0 = Act<T>::rnd(k0 * c - k1 * s), r1 = Act<T>::rnd(k1 * c + k0 * s);
        knew[tid] = r0; knew[tid + HALF] = r1;
        Act<T>::st(kbase + (int64_t)pos * D + tid, r0);
        Act<T>::st(kbase + (int64_t)pos * D + tid + HALF, r1);
    } else if (NT >= 128 + D) {
        if (tid >= 128 && tid < 128 + D) {
            const int i = tid - 128;
            const float v = Act<T>::ld(row + 2 * HD + h * D + i);
            vnew[i] = v;
            Act<T>::st(vbase + (int64_t)pos * D + i, v);
        }
    } else {
        for (int i = tid - HALF; i < D; i += NT - HALF) {   // fewer threads than values: the threads past the RoPE lanes share them
            const float v = Act<T>::ld(row + 2 * HD + h * D + i);
            vnew[i] = v;
            Act<T>::st(vbase + (int64_t)pos * D + i, v);
        }
    }
    __syncthreads();

    // ---- V rows of the first P V pass are requested NOW: their latency overlaps the score / softmax phase below.  (Requesting
    // them -- or the K rows -- even earlier, before the RoPE phase, is slower: vmcnt retires in order, so the few small RoPE
    // loads would then wait behind ~24 row loads.  Measured 329 vs 332 tok/s at B = 1.)
    constexpr int LPK = D / 8, KPW = 64 / LPK, KPB = NW * KPW, UV = 8;
    const int vc_ = lane % LPK, vsub = wave * KPW + lane / LPK;   // chunk of the row, key slot within a block pass
    Raw8<T> vpre[UV];
#pragma unroll
    for (int u = 0; u < UV; ++u) {
        const int j = vsub + u * KPB;
        if (j < pos) vpre[u].load(vbase + (int64_t)j * D + vc_ * 8);   // predicated: no traffic for slots past the context
        else vpre[u].zero();
    }

    // ---- scores over cached keys 0..pos-1 (from HBM) and the new key (from LDS)
    const int Tk = pos + 1;
    const int32_t* km = key_mask ? key_mask + b * key_mask_ld : nullptr;
    float mx = -INFINITY;
    if constexpr (!COOP) {
        for (int j = tid; j < Tk; j += NT) {
            float sv;
            if (km && km[j] == 0) sv = -INFINITY;
            else if (j < pos) sv = scale * RowDot<T, D>::dot(qs, kbase + (int64_t)j * D);
            else {
                float acc = 0.f;
#pragma unroll 8
                for (int c = 0; c < D; ++c) acc += qs[c] * knew[c];
                sv = scale * acc;
            }
            sc[j] = sv;
            mx = fmaxf(mx, sv);
        }
    } else {
        constexpr int UK = 8;   // 8 key rows per lane group in flight (raw, converted at use)
        const int kc_ = vc_, ksub = vsub;
        float qr[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) qr[e] = qs[kc_ * 8 + e];
        for (int j0 = ksub; j0 < pos; j0 += KPB * UK) {
            Raw8<T> kr[UK];
#pragma unroll
            for (int u = 0; u < UK; ++u) {
                const int j = j0 + u * KPB;
                if (j < pos) kr[u].load(kbase + (int64_t)j * D + kc_ * 8);
                else kr[u].zero();
            }
#pragma unroll
            for (int u = 0; u < UK; ++u) {
                float kk[8];
                kr[u].get(kk);
                float acc = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += qr[e] * kk[e];
#pragma unroll
                for (int off = 1; off < LPK; off <<= 1) acc += __shfl_xor(acc, off, 64);
                const int j = j0 + u * KPB;
                if (kc_ == 0 && j < pos) {
                    const float sv = (km && km[j] == 0) ? -INFINITY : scale * acc;
                    sc[j] = sv;
                    mx = fmaxf(mx, sv);
                }
            }
        }
        if (tid == 0) {   // the new key
            float acc = 0.f;
#pragma unroll 8
            for (int c = 0; c < D; ++c) acc += qs[c] * knew[c];
            const float sv = (km && km[pos] == 0) ? -INFINITY : scale * acc;
            sc[pos] = sv;
            mx = fmaxf(mx, sv);
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w]);
    float l = 0.f;
    if (mx > -INFINITY)
        for (int j = tid; j < Tk; j += NT) {
            const float e = __expf(sc[j] - mx);
            sc[j] = e;
            l += e;
        }
    l = wave_sum(l);
    __syncthreads();           // everyone has read red[] (max) before it is reused
    if (lane == 0) red[NW + wave] = l;
    __syncthreads();
    l = red[NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) l += red[NW + w];
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    // probabilities rounded to the activation dtype before P V (HF: softmax(...).to(query.dtype))
    for (int j = tid; j < Tk; j += NT) sc[j] = (mx > -INFINITY) ? Act<T>::rnd(sc[j] * inv) : 0.f;
    __syncthreads();

    // ---- P V.  LPK = D/8 adjacent lanes own the 8-dim chunks of ONE value row (16-byte loads, a full row per lane group),
    // so a wave covers 64/LPK keys per load instruction and the workgroup 4x that; every thread just accumulates its 8 dims
    // over its keys -- all loads of a pass are independent (one HBM latency per 8 keys in flight per thread).
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
    for (int u = 0; u < UV; ++u) {   // pass 0: the rows fetched before the softmax
        const int j = vsub + u * KPB;
        const float pj = j < pos ? sc[j] : 0.f;
        float vv[8];
        vpre[u].get(vv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += pj * vv[e];
    }
    for (int j0 = vsub + KPB * UV; j0 < pos; j0 += KPB * UV) {
        Raw8<T> vr[UV];
        float pj[UV];
#pragma unroll
        for (int u = 0; u < UV; ++u) {
            const int j = j0 + u * KPB;
            const bool ok = j < pos;
            pj[u] = ok ? sc[j] : 0.f;
            if (ok) vr[u].load(vbase + (int64_t)j * D + vc_ * 8);
            else vr[u].zero();
        }
#pragma unroll
        for (int u = 0; u < UV; ++u) {
            float vv[8];
            vr[u].get(vv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += pj[u] * vv[e];
        }
    }
    if (vsub == 0) {  // the new token's value comes from LDS
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += sc[pos] * vnew[vc_ * 8 + e];
    }
    // fold the KPW key slots of the wave (lanes LPK apart), then the NW waves through LDS
#pragma unroll
    for (int off = LPK; off < 64; off <<= 1)
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += __shfl_xor(o[e], off, 64);
    if (lane < LPK) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part[wave * D + lane * 8 + e] = o[e];
    }
    __syncthreads();
    if (out_frag_mt > 0) {
        // fragment-major store for the streaming o_proj GEMM (vcla_gemm_args.A_frag): row b, columns k = h*D + 8 t .. + 7 are
        // one 16-byte fragment slot -> D/8 threads, one 16-byte store each
        if (tid < D / 8) {
            float o8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = tid * 8 + e;
                float t_ = part[c];
#pragma unroll
                for (int w = 1; w < NW; ++w) t_ += part[w * D + c];
                o8[e] = t_;
            }
            const int k = h * D + tid * 8;
            T* dst = out + ((((int64_t)(k >> 5) * out_frag_mt + (b >> 4)) * 64 + ((k & 31) >> 3) * 16 + (b & 15)) << 3);
            if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf2(o8[0], o8[1]), pack_bf2(o8[2], o8[3]), pack_bf2(o8[4], o8[5]), pack_bf2(o8[6], o8[7]));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) Act<T>::st(dst + e, o8[e]);
            }
        }
    } else if (tid < D) {
        float t_ = part[tid];
#pragma unroll
        for (int w = 1; w < NW; ++w) t_ += part[w * D + tid];
        Act<T>::st(out + (int64_t)b * HD + h * D + tid, t_);
    }
}

// =================================================================== single-pass ("flash") form, bf16 (round 3)
// The kernel above walks the cache in dependent phases: K pass(es) -> block-wide max -> exp pass -> sum -> normalise -> V pass(es),
// four barriers and as many exposed HBM round trips as there are passes (context 192 at B = 64: 3 + 2).  Here every lane group
// (D/8 adjacent lanes = one 16-byte chunk each of a key row AND of the same key's value row) keeps its own online-softmax state
// (m, l, o[8]) over the keys it owns, so K and V rows are requested TOGETHER, a batch of U keys ahead of the batch being
// consumed (register double buffer), and nothing in the loop waits on a barrier or on another lane group.  The groups meet once:
// butterfly over the wave, LDS over the waves.  The score reduction over the group's lanes uses DPP row operations (quad_perm /
// row_half_mirror / row_mirror folded into v_add_f32) instead of ds_bpermute, q . k runs on v_dot2c_f32_bf16 against the bf16
// q fragment (q is bf16-rounded after RoPE, as in HF: nothing is lost), and the probabilities stay fp32 (HF rounds the
// normalised p to bf16 before P V; the un-rounded form is closer to the fp32 oracle and inside the stated bf16 bounds).
// K / V rows are read with the non-temporal policy: each row is read by exactly one workgroup per step.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the LPK (4 / 8 / 16) adjacent lanes of a lane group; every lane of the group ends up with the total
template <int LPK> __device__ __forceinline__ float group_sum(float v) {
    v += dpp_mov<0xB1>(v);                        // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);                        // quad_perm [2,3,0,1]
    if (LPK >= 8) v += dpp_mov<0x141>(v);         // row_half_mirror
    if (LPK >= 16) v += dpp_mov<0x140>(v);        // row_mirror
    return v;
}
__device__ __forceinline__ u32x4_t ld_nt16(const bf16_t* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)); }
// 8 bf16 x 8 bf16 -> fp32 on v_dot2c_f32_bf16.  (The pairs are taken with shufflevector: __builtin_bit_cast of an ext-vector
// ELEMENT expression, e.g. bit_cast<bf16x2>(a.y), is miscompiled by this clang -- every element collapses to .x.)
__device__ __forceinline__ float dot8_bf16(const u32x4_t& a, const u32x4_t& b) {
    const bf16x8_t x = __builtin_bit_cast(bf16x8_t, a), y = __builtin_bit_cast(bf16x8_t, b);
    float acc = 0.f;
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(x, x, 0, 1), __builtin_shufflevector(y, y, 0, 1), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(x, x, 2, 3), __builtin_shufflevector(y, y, 2, 3), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(x, x, 4, 5), __builtin_shufflevector(y, y, 4, 5), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(x, x, 6, 7), __builtin_shufflevector(y, y, 6, 7), acc, false);
    return acc;
}

// ---- fp8 (e4m3fn, unit scale) cache rows: 16 elements per 16-byte lane chunk
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two fp32 values that are exactly representable in bf16 (every e4m3 value is) -> packed bf16 pair: the high halves, one v_perm_b32
__device__ __forceinline__ uint32_t hi16_pair(float lo, float hi) {
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
// 16 bf16 (q: two packed quads of dwords) x 16 e4m3 (k) -> fp32: cvt_pk_f32_fp8 + v_perm_b32 + v_dot2c_f32_bf16 per pair
__device__ __forceinline__ float dot16_fp8(const u32x4_t& q0, const u32x4_t& q1, const u32x4_t& k) {
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(k[w], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(k[w], true);
        const uint32_t ka = hi16_pair(lo.x, lo.y), kb = hi16_pair(hi.x, hi.y);
        const uint32_t qa = w < 2 ? q0[2 * w] : q1[2 * w - 4], qb = w < 2 ? q0[2 * w + 1] : q1[2 * w - 3];
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, qa), __builtin_bit_cast(bf16x2_t, ka), acc, false);
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, qb), __builtin_bit_cast(bf16x2_t, kb), acc, false);
    }
    return acc;
}
__device__ __forceinline__ void fp8x16_to_f32(const u32x4_t& t, float* v) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(t[w], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(t[w], true);
        v[4 * w] = lo.x; v[4 * w + 1] = lo.y; v[4 * w + 2] = hi.x; v[4 * w + 3] = hi.y;
    }
}
// e4m3 rounding of one value: the byte, and the value it decodes to
__device__ __forceinline__ unsigned char f32_to_fp8(float x) { return (unsigned char)(VCLA_CVT_PK_FP8_SAT(x, x, 0, false) & 0xff); }
__device__ __forceinline__ float fp8_round(float x) {
    const f32x2_t r = __builtin_amdgcn_cvt_pk_f32_fp8(VCLA_CVT_PK_FP8_SAT(x, x, 0, false), false);
    return r.x;
}

// QP: the qkv row of the step does not exist yet -- the qkv projection ran split in two K slices and left RAW fp32 partial rows
// (vcla_gemm_args.ds_raw_partials).  `qkv` is then the first slice (fp32 [B][3 H D]), `qp.slice` the distance to the second; the lanes
// that read q / k / v sum the slices, apply what the GEMM epilogue would have (the deferred-RMSNorm rstd of the row from its 16 partial
// sums of squares, the fp8 weight scale of the column) and round to bf16: the values the reduce launch would have stored.
struct QkvParts { int64_t slice; const float* ssq; const float* w_scale; float inv_hidden, eps; };
template <int D, int NW, bool MASK, bool KV8 = false, bool QP = false>
__global__ __launch_bounds__(NW * 64, NW == 2 ? 4 : 2) void attn_decode_flash_kernel(const void* __restrict__ qkv_, void* __restrict__ kc_, void* __restrict__ vc_,
                                                                const float* __restrict__ cos_tab, const float* __restrict__ sin_tab,
                                                                bf16_t* __restrict__ out, int H, int ctx_max, int pos0,
                                                                const int32_t* __restrict__ pos_dev, const int32_t* __restrict__ key_mask,
                                                                int64_t key_mask_ld, float scale, int out_frag_mt, QkvParts qp = QkvParts{}) {
    // U keys per lane group and batch.  The 2-wave form (batch decode: thousands of workgroups, 4 waves per SIMD) keeps 2 x 4 rows of K
    // and V per lane in flight; the 4-wave form (a few dozen latency-bound workgroups: B = 1) 2 x 8 -- with D = 128 its 16 groups
    // then have the first 256 keys of the context requested before the RoPE phase ends
    // KV8: the cache holds e4m3 bytes -- E = 16 elements per 16-byte chunk, half the lanes per row, twice the keys per wave load
    // (half the lane loads per batch: a wave load then covers the same number of KEYS as in the bf16 form, and the 16 output / 16 value
    // registers per lane fit next to the rings without spilling -- with U = 4 the 2-wave form spilled 55 registers and ran 64 us
    // instead of 34 at B = 64, context 192)
    constexpr int E = KV8 ? 16 : 8, ESZ = KV8 ? 1 : 2, LPK = D / E, KPW = 64 / LPK, KPB = NW * KPW, HALF = D / 2, NT = NW * 64;
    constexpr int U = (NW == 2 ? 4 : 8) / (KV8 ? 2 : 1);
    static_assert(LPK >= 4, "a lane group is at least a quad (group_sum)");
    // one LDS object (16-byte aligned carve): roped q as packed bf16 [D/2 dwords], new key / value fp32 [D] each, wave partials
    __shared__ __attribute__((aligned(16))) float smem[D / 2 + 2 * D + NW * (D + 2)];
    uint32_t* qpk = reinterpret_cast<uint32_t*>(smem);       // [D/2]  bf16 pairs (2e, 2e+1)
    float* knew = smem + D / 2;                              // [D]
    float* vnew = knew + D;                                  // [D]
    float* part = vnew + D;                                  // [NW][D + 2]: o[D], m, l
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const int pos = pos0 + (pos_dev ? *pos_dev : 0);         // position of the new token = number of cached keys
    const int HD = H * D;
    const bf16_t* row = (const bf16_t*)qkv_ + (int64_t)b * 3 * HD;
    const float* prow = (const float*)qkv_ + (int64_t)b * 3 * HD;      // QP: slice 0 of this sequence's row
    unsigned char* kbase = (unsigned char*)kc_ + ((int64_t)b * H + h) * ctx_max * D * ESZ;
    unsigned char* vbase = (unsigned char*)vc_ + ((int64_t)b * H + h) * ctx_max * D * ESZ;
    const int32_t* km = key_mask ? key_mask + b * key_mask_ld : nullptr;
    const int c = lane % LPK, grp = wave * KPW + lane / LPK;   // chunk of the row, lane group within the workgroup

    // ---- RoPE inputs first (small, needed first), the first batch of cache rows right behind them: vmcnt retires in order, so
    // this order lets the RoPE phase run while the rows are still in flight
    float rc = 0.f, rs = 0.f, q0 = 0.f, q1 = 0.f, k0 = 0.f, k1 = 0.f;
    float qp_rstd = 1.f;
    float pq[2][4];           // QP: the two slices of (q0, q1, k0, k1)
    if constexpr (QP) {
        // rstd of the row (deferred RMSNorm): 16 partial sums of squares, one per lane of every 16-lane row of the wave
        float ss = qp.ssq ? qp.ssq[(int64_t)b * 16 + (lane & 15)] : 0.f;
        ss = group_sum<16>(ss);
        qp_rstd = qp.ssq ? rsqrtf(ss * qp.inv_hidden + qp.eps) : 1.f;
    }
    if (tid < HALF) {
        rc = cos_tab[(int64_t)pos * HALF + tid]; rs = sin_tab[(int64_t)pos * HALF + tid];
        if constexpr (QP) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const float* pr = prow + sl * qp.slice;
                pq[sl][0] = pr[h * D + tid]; pq[sl][1] = pr[h * D + tid + HALF];
                pq[sl][2] = pr[HD + h * D + tid]; pq[sl][3] = pr[HD + h * D + tid + HALF];
            }
        } else {
            q0 = bf2f(row[h * D + tid]); q1 = bf2f(row[h * D + tid + HALF]);
            k0 = bf2f(row[HD + h * D + tid]); k1 = bf2f(row[HD + h * D + tid + HALF]);
        }
    }
    u32x4_t kA[U], vA[U], kB[U], vB[U];
    int mA[U], mB[U];
    const int last = pos > 0 ? pos - 1 : 0;
    // rows past the context are clamped to the last cached row (a line the group reads anyway) and masked at use: the loads
    // are unconditional, so the loop carries no branch around a load and the compiler emits counted waits
    // buffer loads (uniform descriptor per (b, h) slab + one 32-bit offset per row, non-temporal): no 64-bit address arithmetic in
    // the loop, and the policy bit survives (the plain nontemporal load builtin lost it once fences were nearby)
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(kbase, 0, ctx_max * D * ESZ, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(vbase, 0, ctx_max * D * ESZ, 0x00020000);
#define FD_LOAD(KB_, VB_, MB_, t_)                                                                                   \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                                  \
        int j_ = grp + ((t_) * U + u) * KPB;                                                                         \
        j_ = j_ < pos ? j_ : last;                                                                                   \
        const unsigned off_ = (unsigned)(j_ * D * ESZ + c * 16);                                                     \
        KB_[u] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rK, off_, 0, 2 /* nt */));         \
        VB_[u] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rV, off_, 0, 2 /* nt */));         \
        MB_[u] = MASK ? km[j_] : 1;                                                                                  \
    }                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);
    const int nb = (pos + U * KPB - 1) / (U * KPB);          // batches of U keys per lane group (workgroup-uniform)
    if (nb > 0) {            // the first TWO batches go out ahead of the RoPE arithmetic
        FD_LOAD(kA, vA, mA, 0)
        FD_LOAD(kB, vB, mB, 1)
    }
    if constexpr (QP) {
        if (tid < HALF) {     // what the GEMM epilogue + reduce would have stored: (slice 0 + slice 1) * rstd * w_scale, rounded to bf16
            const float w0 = qp.w_scale ? qp.w_scale[h * D + tid] : 1.f, w1 = qp.w_scale ? qp.w_scale[h * D + tid + HALF] : 1.f;
            const float w2 = qp.w_scale ? qp.w_scale[HD + h * D + tid] : 1.f, w3 = qp.w_scale ? qp.w_scale[HD + h * D + tid + HALF] : 1.f;
            q0 = Act<bf16_t>::rnd((pq[0][0] + pq[1][0]) * qp_rstd * w0); q1 = Act<bf16_t>::rnd((pq[0][1] + pq[1][1]) * qp_rstd * w1);
            k0 = Act<bf16_t>::rnd((pq[0][2] + pq[1][2]) * qp_rstd * w2); k1 = Act<bf16_t>::rnd((pq[0][3] + pq[1][3]) * qp_rstd * w3);
        }
    }
    if (tid < HALF) {
        const float cr = Act<bf16_t>::rnd(rc), sr = Act<bf16_t>::rnd(rs);
        const float a0 = Act<bf16_t>::rnd(q0 * cr - q1 * sr), a1 = Act<bf16_t>::rnd(q1 * cr + q0 * sr);
        const float r0 = Act<bf16_t>::rnd(k0 * cr - k1 * sr), r1 = Act<bf16_t>::rnd(k1 * cr + k0 * sr);
        reinterpret_cast<bf16_t*>(qpk)[tid] = f2bf(a0);
        reinterpret_cast<bf16_t*>(qpk)[tid + HALF] = f2bf(a1);
        if constexpr (KV8) {     // the cache -- and this step's own score -- see the e4m3 rounding of the new key
            knew[tid] = fp8_round(r0); knew[tid + HALF] = fp8_round(r1);
            kbase[(int64_t)pos * D + tid] = f32_to_fp8(r0);
            kbase[(int64_t)pos * D + tid + HALF] = f32_to_fp8(r1);
        } else {
            knew[tid] = r0; knew[tid + HALF] = r1;
            reinterpret_cast<bf16_t*>(kbase)[(int64_t)pos * D + tid] = f2bf(r0);
            reinterpret_cast<bf16_t*>(kbase)[(int64_t)pos * D + tid + HALF] = f2bf(r1);
        }
    } else {
        for (int i = tid - HALF; i < D; i += NT - HALF) {    // the threads past the RoPE lanes move the value row
            bf16_t v;
            if constexpr (QP) {
                const float ws_ = qp.w_scale ? qp.w_scale[2 * HD + h * D + i] : 1.f;
                v = f2bf((prow[2 * HD + h * D + i] + prow[qp.slice + 2 * HD + h * D + i]) * qp_rstd * ws_);
            } else {
                v = row[2 * HD + h * D + i];
            }
            if constexpr (KV8) {
                vnew[i] = fp8_round(bf2f(v));
                vbase[(int64_t)pos * D + i] = f32_to_fp8(bf2f(v));
            } else {
                vnew[i] = bf2f(v);
                reinterpret_cast<bf16_t*>(vbase)[(int64_t)pos * D + i] = v;
            }
        }
    }
    // LDS-only release / acquire around a bare s_barrier: __syncthreads() would also drain vmcnt (the cache-append stores AND,
    // in order behind them, the first batch of row loads) in every wave before any of them may start on the scores
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    const u32x4_t qv = *reinterpret_cast<const u32x4_t*>(qpk + c * (E / 2));      // this lane's first 8 q values, packed bf16
    u32x4_t qv1 = qv;                                                             // KV8: values 8 .. 15 of the chunk
    if constexpr (KV8) qv1 = *reinterpret_cast<const u32x4_t*>(qpk + c * (E / 2) + 4);
    const float sl2 = scale * 1.44269504088896340736f;                     // scores live in the log2 domain

    float m_run = -INFINITY, l_run = 0.f, o[E];
#pragma unroll
    for (int e = 0; e < E; ++e) o[e] = 0.f;
#define FD_COMPUTE(KB_, VB_, MB_, t_)                                                                                \
    {                                                                                                                \
        float s_[U];                                                                                                 \
        float mb_ = -INFINITY;                                                                                       \
        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                              \
            const int j_ = grp + ((t_) * U + u) * KPB;                                                               \
            const float d_ = group_sum<LPK>(KV8 ? dot16_fp8(qv, qv1, KB_[u]) : dot8_bf16(qv, KB_[u]));               \
            s_[u] = (j_ < pos && MB_[u] != 0) ? d_ * sl2 : -INFINITY;                                                \
            mb_ = fmaxf(mb_, s_[u]);                                                                                 \
        }                                                                                                            \
        const float mn_ = fmaxf(m_run, mb_);                                                                         \
        const float mu_ = mn_ == -INFINITY ? 0.f : mn_;        /* nothing visible so far: keep everything at 0 */    \
        const float al_ = __builtin_amdgcn_exp2f(m_run - mu_);                                                       \
        m_run = mn_;                                                                                                 \
        float ps_ = 0.f;                                                                                             \
        _Pragma("unroll") for (int e = 0; e < E; ++e) o[e] *= al_;                                                   \
        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                              \
            const float p_ = __builtin_amdgcn_exp2f(s_[u] - mu_);                                                    \
            ps_ += p_;                                                                                               \
            float vv_[E];                                                                                            \
            if constexpr (KV8) fp8x16_to_f32(VB_[u], vv_);                                                           \
            else bf8_to_f32(make_uint4(VB_[u].x, VB_[u].y, VB_[u].z, VB_[u].w), vv_);                                \
            _Pragma("unroll") for (int e = 0; e < E; ++e) o[e] = __builtin_fmaf(p_, vv_[e], o[e]);                   \
        }                                                                                                            \
        l_run = l_run * al_ + ps_;                                                                                   \
    }                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);
    // two batches per trip, both computed unconditionally (a batch past the context is all masked: p = 0, alpha = 1): a branch
    // around the second compute lets the compiler sink the B loads into it, right in front of their use -- no prefetch left
    for (int t = 0; t < nb; t += 2) {
        FD_COMPUTE(kA, vA, mA, t)
        FD_LOAD(kA, vA, mA, t + 2)                 // (rows past the context: a harmless re-load of the last row)
        FD_COMPUTE(kB, vB, mB, t + 1)
        FD_LOAD(kB, vB, mB, t + 3)
    }
#undef FD_LOAD
#undef FD_COMPUTE
    {   // the new token (key / value still in LDS): every group computes it, only group 0 of the workgroup folds it in
        float d_ = 0.f;
#pragma unroll
        for (int e = 0; e < E; e += 2) {
            const uint32_t qq = qpk[c * (E / 2) + e / 2];
            d_ = __builtin_fmaf(__uint_as_float(qq << 16), knew[c * E + e], d_);
            d_ = __builtin_fmaf(__uint_as_float(qq & 0xffff0000u), knew[c * E + e + 1], d_);
        }
        d_ = group_sum<LPK>(d_);
        const float s_ = (grp == 0 && (!MASK || km[pos] != 0)) ? d_ * sl2 : -INFINITY;
        const float mn_ = fmaxf(m_run, s_);
        const float mu_ = mn_ == -INFINITY ? 0.f : mn_;
        const float al_ = __builtin_amdgcn_exp2f(m_run - mu_), p_ = __builtin_amdgcn_exp2f(s_ - mu_);
        m_run = mn_;
        l_run = l_run * al_ + p_;
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = __builtin_fmaf(p_, vnew[c * E + e], o[e] * al_);
    }
    // ---- merge the KPW lane groups of the wave (butterfly over lanes LPK, 2 LPK, ... apart), then the waves through LDS
#pragma unroll
    for (int off = LPK; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(m_run, off, 64), l2 = __shfl_xor(l_run, off, 64);
        const float mn_ = fmaxf(m_run, m2);
        const float mu_ = mn_ == -INFINITY ? 0.f : mn_;
        const float a1 = __builtin_amdgcn_exp2f(m_run - mu_), a2 = __builtin_amdgcn_exp2f(m2 - mu_);
        l_run = l_run * a1 + l2 * a2;
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = o[e] * a1 + __shfl_xor(o[e], off, 64) * a2;
        m_run = mn_;
    }
    if (lane < LPK) {
        float* pw = part + wave * (D + 2);
#pragma unroll
        for (int e = 0; e < E; ++e) pw[lane * E + e] = o[e];
        if (lane == 0) { pw[D] = m_run; pw[D + 1] = l_run; }
    }
    __syncthreads();
    if (tid < D / 8) {     // D/8 threads, 8 output dims each
        float mf = part[D];
#pragma unroll
        for (int w = 1; w < NW; ++w) mf = fmaxf(mf, part[w * (D + 2) + D]);
        const float mu_ = mf == -INFINITY ? 0.f : mf;
        float lf = 0.f, o8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float* pw = part + w * (D + 2);
            const float a_ = __builtin_amdgcn_exp2f(pw[D] - mu_);
            lf += pw[D + 1] * a_;
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] += pw[tid * 8 + e] * a_;
        }
        const float inv = lf > 0.f ? 1.0f / lf : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] *= inv;
        const int k = h * D + tid * 8;
        bf16_t* dst = out_frag_mt > 0 ? out + ((((int64_t)(k >> 5) * out_frag_mt + (b >> 4)) * 64 + ((k & 31) >> 3) * 16 + (b & 15)) << 3)
                                      : out + (int64_t)b * HD + k;
        *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf2(o8[0], o8[1]), pack_bf2(o8[2], o8[3]), pack_bf2(o8[4], o8[5]), pack_bf2(o8[6], o8[7]));
    }
}

template <typename T, int D>
static int launch_decode(const void* qkv, void* kc, void* vc, const float* cos_tab, const float* sin_tab, void* out, int B,
                         int H, int ctx_max, int pos0, const int32_t* pos_dev, const int32_t* key_mask, int64_t key_mask_ld,
                         float scale, int out_frag, hipStream_t s, bool kv8 = false, const QkvParts* qp = nullptr) {
    const int sc_cap = (ctx_max + 63) & ~63;
    // batch form: 2-wave workgroups (8 per CU: one round for B * H <= 2048) while the context is short enough that the extra
    // passes over K and V (64 instead of 128 rows per pass) cost less than the second round of workgroups saves
    // (measured at B = 64, H = 32, context 192: 46.0 -> 42.4 us per launch; VCLA_ATTN_NW=4 restores the 4-wave form)
    static const int nw_env = getenv("VCLA_ATTN_NW") ? atoi(getenv("VCLA_ATTN_NW")) : 0;
    const int NWs = (D == 128 && (int64_t)B * H >= 1024 && nw_env != 4) ? 2 : 4;
    const size_t lds = (size_t)(3 * D + 2 * NWs + NWs * D + sc_cap) * sizeof(float);
    VCLA_REQUIRE(lds <= 64 * 1024, VCLA_ERR_BAD_SHAPE, "attn_decode: ctx_max=%d needs %zu B of LDS (max 64 KiB)", ctx_max, lds);
    dim3 grid(H, B);
    const int out_frag_mt = out_frag ? (B + 15) / 16 : 0;
    if constexpr (sizeof(T) == 2) {
        // bf16: the single-pass kernel (VCLA_ATTN_FLASH=0 restores the phased kernel below for A/B runs).  2-wave workgroups once
        // B * H fills the chip that way (16 per CU by waves), 4 waves otherwise; the row-major output needs 16-byte rows.
        static const int flash_env = getenv("VCLA_ATTN_FLASH") ? atoi(getenv("VCLA_ATTN_FLASH")) : 1;
        const bool flash_ok = out_frag || (((int64_t)H * D) % 8 == 0 && vcla_aligned(out, 16));
        const bool small_wg = (int64_t)B * H >= 1024 && nw_env != 4;
#define FD_GO(NW_, MASK_, KV8_) attn_decode_flash_kernel<D, NW_, MASK_, KV8_><<<grid, NW_ * 64, 0, s>>>(qkv, kc, vc, cos_tab, sin_tab, \
                                                    (bf16_t*)out, H, ctx_max, pos0, pos_dev, key_mask, key_mask_ld, scale, out_frag_mt)
        if (qp) {    // q / k / v arrive as two raw fp32 K slices of the qkv projection: 2-wave form only (batch decode, B * H >= 1024)
            if constexpr (D >= 64) {
                VCLA_REQUIRE(flash_ok && small_wg, VCLA_ERR_BAD_ARG, "attn_decode: the split-qkv form needs B * H >= 1024 and a 16-byte aligned output");
#define FD_GOP(MASK_, KV8_) attn_decode_flash_kernel<D, 2, MASK_, KV8_, true><<<grid, 128, 0, s>>>(qkv, kc, vc, cos_tab, sin_tab, (bf16_t*)out, H, ctx_max, pos0, pos_dev, \
                                                                                                 key_mask, key_mask_ld, scale, out_frag_mt, *qp)
                if (kv8) { if (key_mask) FD_GOP(true, true); else FD_GOP(false, true); }
                else { if (key_mask) FD_GOP(true, false); else FD_GOP(false, false); }
#undef FD_GOP
                VCLA_CHECK_LAUNCH("attn_decode_flash_kernel<split qkv>");
                return VCLA_OK;
            } else {
                return vcla_fail(VCLA_ERR_BAD_SHAPE, "attn_decode: the split-qkv form needs head dim 64 or 128 (got %d)", D);
            }
        }
        if (kv8) {   // e4m3 cache rows: the single-pass kernel is the only reader
            if constexpr (D >= 64) {
                VCLA_REQUIRE(flash_ok, VCLA_ERR_BAD_ARG, "attn_decode: the fp8 cache needs a 16-byte aligned output with H * d %% 8 == 0");
                if (small_wg) { if (key_mask) FD_GO(2, true, true); else FD_GO(2, false, true); }
                else { if (key_mask) FD_GO(4, true, true); else FD_GO(4, false, true); }
                VCLA_CHECK_LAUNCH("attn_decode_flash_kernel<fp8 cache>");
                return VCLA_OK;
            } else {
                return vcla_fail(VCLA_ERR_BAD_SHAPE, "attn_decode: the fp8 cache needs head dim 64 or 128 (got %d)", D);
            }
        }
        if (flash_env && flash_ok) {
            if (small_wg) { if (key_mask) FD_GO(2, true, false); else FD_GO(2, false, false); }
            else { if (key_mask) FD_GO(4, true, false); else FD_GO(4, false, false); }
            VCLA_CHECK_LAUNCH("attn_decode_flash_kernel");
            return VCLA_OK;
        }
#undef FD_GO
    }
    static const int coop_env = getenv("VCLA_ATTN_COOP") ? atoi(getenv("VCLA_ATTN_COOP")) : -1;   // -1 auto, 0 / 1 force (A/B runs)
    const bool coop = coop_env >= 0 ? coop_env != 0 : (int64_t)B * H >= 512;
    if (coop && NWs == 2)
        attn_decode_kernel<T, D, true, 2><<<grid, 128, lds, s>>>((const T*)qkv, (T*)kc, (T*)vc, cos_tab, sin_tab, (T*)out, H, ctx_max, pos0,
                                                                 pos_dev, key_mask, key_mask_ld, scale, sc_cap, out_frag_mt);
    else if (coop)
        attn_decode_kernel<T, D, true><<<grid, 256, lds, s>>>((const T*)qkv, (T*)kc, (T*)vc, cos_tab, sin_tab, (T*)out, H, ctx_max, pos0,
                                                              pos_dev, key_mask, key_mask_ld, scale, sc_cap, out_frag_mt);
    else
        attn_decode_kernel<T, D, false><<<grid, 256, lds, s>>>((const T*)qkv, (T*)kc, (T*)vc, cos_tab, sin_tab, (T*)out, H, ctx_max, pos0,
                                                               pos_dev, key_mask, key_mask_ld, scale, sc_cap, out_frag_mt);
    VCLA_CHECK_LAUNCH("attn_decode_kernel");
    return VCLA_OK;
}

static int attn_decode_entry(const void* qkv, void* k_cache, void* v_cache, const float* cos_tab, const float* sin_tab, void* out, int B, int H, int d,
                             int ctx_max, int pos0, const int32_t* pos_dev, const int32_t* key_mask, int64_t key_mask_ld, float scale, int dtype, int out_frag,
                             void* stream, const QkvParts* qp);

extern "C" int vcla_attn_decode_fused_parts(const float* qkv_parts, int64_t slice_stride, const float* row_ssq, const float* w_scale, float norm_eps,
                                            void* k_cache, void* v_cache, const float* cos_tab, const float* sin_tab, void* out, int B, int H, int d,
                                            int ctx_max, int pos0, const int32_t* pos_dev, const int32_t* key_mask, int64_t key_mask_ld, float scale,
                                            int dtype, int out_frag, void* stream) {
    VCLA_REQUIRE(qkv_parts && slice_stride >= (int64_t)B * 3 * H * d && vcla_aligned(qkv_parts, 16), VCLA_ERR_BAD_ARG,
                 "attn_decode_parts: qkv_parts = two fp32 slices [B][3 H d], slice_stride elements apart");
    VCLA_REQUIRE((dtype & ~VCLA_KV_FP8) == VCLA_BF16, VCLA_ERR_BAD_DTYPE, "attn_decode_parts: bf16 activations only");
    QkvParts qp{slice_stride, row_ssq, w_scale, 1.0f / (float)(H * d), norm_eps};
    return attn_decode_entry(qkv_parts, k_cache, v_cache, cos_tab, sin_tab, out, B, H, d, ctx_max, pos0, pos_dev, key_mask, key_mask_ld, scale, dtype, out_frag,
                             stream, &qp);
}

extern "C" int vcla_attn_decode_fused(const void* qkv, void* k_cache, void* v_cache, const float* cos_tab,
                                      const float* sin_tab, void* out, int B, int H, int d, int ctx_max, int pos0,
                                      const int32_t* pos_dev, const int32_t* key_mask, int64_t key_mask_ld, float scale,
                                      int dtype, int out_frag, void* stream) {
    return attn_decode_entry(qkv, k_cache, v_cache, cos_tab, sin_tab, out, B, H, d, ctx_max, pos0, pos_dev, key_mask, key_mask_ld, scale, dtype, out_frag, stream,
                             nullptr);
}

static int attn_decode_entry(const void* qkv, void* k_cache, void* v_cache, const float* cos_tab,
                             const float* sin_tab, void* out, int B, int H, int d, int ctx_max, int pos0,
                             const int32_t* pos_dev, const int32_t* key_mask, int64_t key_mask_ld, float scale,
                             int dtype, int out_frag, void* stream, const QkvParts* qp) {
    const bool kv8 = (dtype & VCLA_KV_FP8) != 0;
    dtype &= ~VCLA_KV_FP8;
    VCLA_REQUIRE(dtype == VCLA_F32 || dtype == VCLA_BF16, VCLA_ERR_BAD_DTYPE, "attn_decode: bad dtype %d", dtype);
    VCLA_REQUIRE(!kv8 || dtype == VCLA_BF16, VCLA_ERR_BAD_DTYPE, "attn_decode: VCLA_KV_FP8 goes with VCLA_BF16 activations only");
    VCLA_REQUIRE(d == 32 || d == 64 || d == 128, VCLA_ERR_BAD_SHAPE, "attn_decode: head dim %d not in {32,64,128}", d);
    VCLA_REQUIRE(B >= 0 && H > 0 && ctx_max > 0 && pos0 >= 0 && (pos_dev || pos0 < ctx_max), VCLA_ERR_BAD_SHAPE,
                 "attn_decode: B=%d H=%d ctx_max=%d pos0=%d", B, H, ctx_max, pos0);
    VCLA_REQUIRE(qkv && k_cache && v_cache && cos_tab && sin_tab && out, VCLA_ERR_BAD_ARG, "attn_decode: null pointer");
    VCLA_REQUIRE(vcla_aligned(k_cache, 16) && vcla_aligned(v_cache, 16) && vcla_aligned(qkv, 16), VCLA_ERR_BAD_ARG,
                 "attn_decode: buffers must be 16-byte aligned");
    VCLA_REQUIRE(!out_frag || (dtype == VCLA_BF16 && B <= 64 && (H * d) % 32 == 0), VCLA_ERR_BAD_ARG,
                 "attn_decode: out_frag needs bf16, B <= 64 (got %d) and H*d %% 32 == 0", B);
    if (B == 0) return VCLA_OK;
    hipStream_t s = (hipStream_t)stream;
#define DEC_CASE(TT, DD) return launch_decode<TT, DD>(qkv, k_cache, v_cache, cos_tab, sin_tab, out, B, H, ctx_max, pos0, pos_dev, key_mask, key_mask_ld, scale, out_frag, s, kv8, qp)
    if (dtype == VCLA_F32) {
        if (d == 32) DEC_CASE(float, 32);
        if (d == 64) DEC_CASE(float, 64);
        DEC_CASE(float, 128);
    } else {
        if (d == 32) DEC_CASE(bf16_t, 32);
        if (d == 64) DEC_CASE(bf16_t, 64);
        DEC_CASE(bf16_t, 128);
    }
#undef DEC_CASE
}
