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

template <typename T, int D>
static int launch_decode(const void* qkv, void* kc, void* vc, const float* cos_tab, const float* sin_tab, void* out, int B,
                         int H, int ctx_max, int pos0, const int32_t* pos_dev, const int32_t* key_mask, int64_t key_mask_ld,
                         float scale, int out_frag, hipStream_t s) {
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

extern "C" int vcla_attn_decode_fused(const void* qkv, void* k_cache, void* v_cache, const float* cos_tab,
                                      const float* sin_tab, void* out, int B, int H, int d, int ctx_max, int pos0,
                                      const int32_t* pos_dev, const int32_t* key_mask, int64_t key_mask_ld, float scale,
                                      int dtype, int out_frag, void* stream) {
    VCLA_REQUIRE(dtype == VCLA_F32 || dtype == VCLA_BF16, VCLA_ERR_BAD_DTYPE, "attn_decode: bad dtype %d", dtype);
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
#define DEC_CASE(TT, DD) return launch_decode<TT, DD>(qkv, k_cache, v_cache, cos_tab, sin_tab, out, B, H, ctx_max, pos0, pos_dev, key_mask, key_mask_ld, scale, out_frag, s)
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
