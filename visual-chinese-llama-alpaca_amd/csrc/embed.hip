// embed.hip -- data-movement kernels either side of the GEMMs: patch gather (im2col), token
// embedding gather + image splice, RoPE + KV-cache append, greedy argmax.  All HBM-bound.
#include "vcla_common.h"

// ------------------------------------------------------------------ im2col for the patch conv
// out[(b*gh + py)*gw + px][c*P*P + ky*P + kx] = pix[b][c][py*P+ky][px*P+kx]; columns >= C*P*P are zero.
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const T* __restrict__ pix, T* __restrict__ out, int C, int H,
                                                     int W, int P, int k_pad) {
    const int gw = W / P, gh = H / P;
    const int64_t prow = blockIdx.x;  // patch row index
    const int b = (int)(prow / (gh * gw));
    const int py = (int)((prow / gw) % gh), px = (int)(prow % gw);
    const int kreal = C * P * P;
    T* o = out + prow * k_pad;
    for (int k = threadIdx.x; k < k_pad; k += 256) {
        if (k < kreal) {
            const int c = k / (P * P), ky = (k / P) % P, kx = k % P;
            o[k] = pix[(((int64_t)b * C + c) * H + (py * P + ky)) * W + (px * P + kx)];
        } else {
            Act<T>::st(o + k, 0.f);
        }
    }
}

extern "C" int vcla_im2col(const void* pixels, void* patches, int B, int C, int H, int W, int P, int k_pad,
                           int dtype, void* stream) {
    VCLA_REQUIRE(dtype == VCLA_F32 || dtype == VCLA_BF16, VCLA_ERR_BAD_DTYPE, "im2col: bad dtype %d", dtype);
    VCLA_REQUIRE(B >= 0 && C > 0 && P > 0 && H % P == 0 && W % P == 0 && k_pad >= C * P * P, VCLA_ERR_BAD_SHAPE,
                 "im2col: B=%d C=%d H=%d W=%d P=%d k_pad=%d", B, C, H, W, P, k_pad);
    VCLA_REQUIRE(pixels && patches, VCLA_ERR_BAD_ARG, "im2col: null pointer");
    const int64_t rows = (int64_t)B * (H / P) * (W / P);
    if (rows == 0) return VCLA_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VCLA_F32)
        im2col_kernel<float><<<(unsigned)rows, 256, 0, s>>>((const float*)pixels, (float*)patches, C, H, W, P, k_pad);
    else
        im2col_kernel<bf16_t><<<(unsigned)rows, 256, 0, s>>>((const bf16_t*)pixels, (bf16_t*)patches, C, H, W, P, k_pad);
    VCLA_CHECK_LAUNCH("im2col_kernel");
    return VCLA_OK;
}

// ------------------------------------------------------------------ request validation (every data-dependent check of a request in ONE launch)
// One wave per prompt row; lane l reads positions l, l + 64, ...  The reference raises from Python after `.nonzero()` / `.any()` round trips
// (models/visualcla/modeling_visualcla.py:296-302 forward, :362-367 generate); round 3 - 5 here ran ~25 elementwise / reduce launches of torch for the
// same answers (0.4 ms in front of every forward / generate).  flags (int32[5], zeroed by the launcher): [0] an id outside [0, vocab), [1] an image slot
// whose <img> is not followed by q_tokens fillers and </img>, [2] a masked position anywhere, [3] a visible position AFTER a masked one that itself follows
// a visible one (a hole, not padding at either end), [4] a label outside [0, vocab) other than -100.  img_pos[b] = first <img> of row b, -1: none.
__global__ __launch_bounds__(64) void check_request_kernel(const int64_t* __restrict__ ids, int T, int vocab, int q_tokens, int64_t start_id, int64_t end_id,
                                                           int64_t tok_id, int need_tok, const int64_t* __restrict__ mask, int Tm, int mask_prefix_visible,
                                                           const int64_t* __restrict__ labels, int Tl, int32_t* __restrict__ img_pos,
                                                           int32_t* __restrict__ flags) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int64_t* row = ids + (int64_t)b * T;
    bool bad = false, tok = false;
    int first = T;                                                   // first <img> seen by this lane
    for (int t = lane; t < T; t += 64) {
        const int64_t id = row[t];
        bad |= id < 0 || id >= vocab;
        tok |= id == tok_id;
        if (id == start_id && t < first) first = t;
    }
    if (__ballot(bad)) { if (lane == 0) atomicOr(flags + 0, 1); }
    if (q_tokens > 0) {
        for (int o = 32; o; o >>= 1) first = min(first, __shfl_xor(first, o));
        bool has = first < T;
        if (need_tok) has = has && __ballot(tok) != 0;
        if (lane == 0) {
            const int endpos = first + q_tokens + 1;
            const bool ok = endpos < T && row[endpos < T ? endpos : 0] == end_id;
            if (has && !ok) atomicOr(flags + 1, 1);
            if (img_pos) img_pos[b] = has ? first : -1;
        }
    }
    if (mask) {
        const int64_t* mrow = mask + (int64_t)b * Tm;
        bool seen_vis = mask_prefix_visible != 0, seen_hole = false, any_masked = false, gap = false;      // wave-uniform running state
        for (int t0 = 0; t0 < Tm; t0 += 64) {
            const int t = t0 + lane;
            const bool in = t < Tm;
            const bool vis = in && mrow[t] != 0;
            const unsigned long long V = __ballot(vis), Z = __ballot(in && !vis);
            any_masked |= Z != 0;
            if (seen_hole) gap |= V != 0;
            else {
                unsigned long long holes = Z;                        // masked positions that follow a visible one
                if (!seen_vis) holes = V ? (Z & ~((2ull << __builtin_ctzll(V)) - 1ull)) : 0ull;
                if (holes) {
                    const int h0 = __builtin_ctzll(holes);
                    gap |= h0 < 63 && (V >> (h0 + 1)) != 0;
                    seen_hole = true;
                }
            }
            seen_vis |= V != 0;
        }
        if (lane == 0) {
            if (any_masked) atomicOr(flags + 2, 1);
            if (gap) atomicOr(flags + 3, 1);
        }
    }
    if (labels) {
        const int64_t* lrow = labels + (int64_t)b * Tl;
        bool badl = false;
        for (int t = lane; t < Tl; t += 64) {
            const int64_t l = lrow[t];
            badl |= l != -100 && (l < 0 || l >= vocab);
        }
        if (__ballot(badl)) { if (lane == 0) atomicOr(flags + 4, 1); }
    }
}

extern "C" int vcla_check_request(const int64_t* ids, int B, int T, int vocab, int q_tokens, int64_t start_id, int64_t end_id, int64_t tok_id, int need_tok,
                                  const int64_t* mask, int Tm, int mask_prefix_visible, const int64_t* labels, int Tl, int32_t* img_pos, int32_t* flags,
                                  void* stream) {
    VCLA_REQUIRE(ids && flags && B > 0 && T > 0 && vocab > 0 && q_tokens >= 0, VCLA_ERR_BAD_ARG, "check_request: B=%d T=%d vocab=%d q_tokens=%d", B, T, vocab, q_tokens);
    VCLA_REQUIRE((!mask || Tm > 0) && (!labels || Tl > 0), VCLA_ERR_BAD_SHAPE, "check_request: mask / labels need a positive length (Tm=%d Tl=%d)", Tm, Tl);
    hipStream_t s = (hipStream_t)stream;
    VCLA_CHECK_HIP(hipMemsetAsync(flags, 0, 5 * sizeof(int32_t), s));
    check_request_kernel<<<B, 64, 0, s>>>(ids, T, vocab, q_tokens, start_id, end_id, tok_id, need_tok, mask, Tm, mask_prefix_visible, labels, Tl, img_pos, flags);
    VCLA_CHECK_LAUNCH("check_request_kernel");
    return VCLA_OK;
}

// ------------------------------------------------------------------ embedding gather + image splice
template <typename T>
__global__ __launch_bounds__(256) void embed_splice_kernel(const int64_t* __restrict__ ids,
                                                           const bf16_t* __restrict__ table,
                                                           const T* __restrict__ img,
                                                           const int32_t* __restrict__ img_pos, T* __restrict__ out,
                                                           int T_, int Q, int D, int V) {
    const int64_t row = blockIdx.x;
    const int b = (int)(row / T_), t = (int)(row % T_);
    T* o = out + row * D;
    const int p0 = (img && img_pos) ? img_pos[b] : -1;
    if (p0 >= 0 && t > p0 && t <= p0 + Q) {
        const T* src = img + ((int64_t)b * Q + (t - p0 - 1)) * D;
        for (int c = threadIdx.x; c < D; c += 256) o[c] = src[c];
    } else {
        int64_t id = ids[row];
        if (id < 0 || id >= V) id = 0;  // out-of-range ids are rejected on the host; stay in bounds here
        const bf16_t* src = table + id * D;
        for (int c = threadIdx.x; c < D; c += 256) Act<T>::st(o + c, bf2f(src[c]));
    }
}

extern "C" int vcla_embed_splice(const int64_t* ids, const void* table, const void* image_embeds,
                                 const int32_t* img_pos, void* out, int B, int T, int Q, int D, int V, int dtype,
                                 void* stream) {
    VCLA_REQUIRE(dtype == VCLA_F32 || dtype == VCLA_BF16, VCLA_ERR_BAD_DTYPE, "embed_splice: bad dtype %d", dtype);
    VCLA_REQUIRE(B >= 0 && T >= 0 && D > 0 && V > 0 && Q >= 0, VCLA_ERR_BAD_SHAPE, "embed_splice: B=%d T=%d Q=%d D=%d V=%d",
                 B, T, Q, D, V);
    VCLA_REQUIRE(ids && table && out, VCLA_ERR_BAD_ARG, "embed_splice: null pointer");
    const int64_t rows = (int64_t)B * T;
    if (rows == 0) return VCLA_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VCLA_F32)
        embed_splice_kernel<float><<<(unsigned)rows, 256, 0, s>>>(ids, (const bf16_t*)table, (const float*)image_embeds,
                                                                  img_pos, (float*)out, T, Q, D, V);
    else
        embed_splice_kernel<bf16_t><<<(unsigned)rows, 256, 0, s>>>(ids, (const bf16_t*)table,
                                                                   (const bf16_t*)image_embeds, img_pos, (bf16_t*)out,
                                                                   T, Q, D, V);
    VCLA_CHECK_LAUNCH("embed_splice_kernel");
    return VCLA_OK;
}

// ------------------------------------------------------------------ RoPE (rotate-half form) + KV append
// qkv row r = b*T + t : [ q (H*d) | k (H*d) | v (H*d) ].  One workgroup per row; thread handles (h, i<d/2) pairs.
// q' = q*cos + rot(q)*sin with rot(x) = cat(-x[d/2:], x[:d/2]); cos/sin are rounded to the activation dtype
// first (HF casts the fp32 tables to x.dtype, hf:llama/modeling_llama.py:127).
// KV8 (bf16 activations): the cache holds OCP e4m3 bytes (unit scale); the rotated k is also written back into the qkv buffer so that
// the prefill's own attention reads exact bf16 rows (vcla_model_cfg.t_kv_fp8).
template <typename T, bool KV8 = false>
__global__ __launch_bounds__(256) void rope_kv_kernel(T* __restrict__ qkv, void* __restrict__ kc_, void* __restrict__ vc_,
                                                      const float* __restrict__ cos_tab,
                                                      const float* __restrict__ sin_tab, int T_, int H, int d,
                                                      int ctx_max, int pos0, const int32_t* __restrict__ pos_dev) {
    const int64_t row = blockIdx.x;
    const int b = (int)(row / T_), t = (int)(row % T_);
    const int pos = pos0 + (pos_dev ? *pos_dev : 0) + t;
    const int half = d / 2;
    const int HD = H * d;
    T* q = qkv + row * 3 * HD;
    T* k = q + HD;
    const T* v = k + HD;
    T* kc = (T*)kc_;
    T* vc = (T*)vc_;
    unsigned char* kc8 = (unsigned char*)kc_;
    unsigned char* vc8 = (unsigned char*)vc_;
    const float* cs = cos_tab + (int64_t)pos * half;
    const float* sn = sin_tab + (int64_t)pos * half;
    for (int idx = threadIdx.x; idx < H * half; idx += 256) {
        const int h = idx / half, i = idx % half;
        const float c = Act<T>::rnd(cs[i]), s = Act<T>::rnd(sn[i]);
        const int o = h * d + i;
        const float q0 = Act<T>::ld(q + o), q1 = Act<T>::ld(q + o + half);
        Act<T>::st(q + o, q0 * c - q1 * s);
        Act<T>::st(q + o + half, q1 * c + q0 * s);
        const float k0 = Act<T>::ld(k + o), k1 = Act<T>::ld(k + o + half);
        const int64_t ko = (((int64_t)b * H + h) * ctx_max + pos) * d + i;
        if constexpr (KV8) {
            const float r0 = Act<T>::rnd(k0 * c - k1 * s), r1 = Act<T>::rnd(k1 * c + k0 * s);
            Act<T>::st(k + o, r0);
            Act<T>::st(k + o + half, r1);
            const unsigned pk = (unsigned)VCLA_CVT_PK_FP8_SAT(r0, r1, 0, false);
            kc8[ko] = (unsigned char)(pk & 0xff);
            kc8[ko + half] = (unsigned char)((pk >> 8) & 0xff);
        } else {
            Act<T>::st(kc + ko, k0 * c - k1 * s);
            Act<T>::st(kc + ko + half, k1 * c + k0 * s);
        }
    }
    if constexpr (KV8) {
        for (int idx = threadIdx.x * 2; idx < HD; idx += 512) {     // two adjacent values per thread: one 2-byte store (d is even)
            const int h = idx / d, i = idx % d;
            const unsigned pk = (unsigned)VCLA_CVT_PK_FP8_SAT(Act<T>::ld(v + idx), Act<T>::ld(v + idx + 1), 0, false);
            *reinterpret_cast<unsigned short*>(vc8 + (((int64_t)b * H + h) * ctx_max + pos) * d + i) = (unsigned short)(pk & 0xffff);
        }
    } else {
        for (int idx = threadIdx.x; idx < HD; idx += 256) {
            const int h = idx / d, i = idx % d;
            vc[(((int64_t)b * H + h) * ctx_max + pos) * d + i] = v[idx];
        }
    }
}

// bf16, d/2 a multiple of 8: the same arithmetic on 16-byte chunks -- a thread owns 8 consecutive pair indices of one head (two 16-byte loads
// of q, two of k, 8 cos / sin values), rotates, stores q in place and k into the cache with 16-byte stores (8-byte stores of e4m3 bytes with
// KV8, which also writes k back in place); the value row moves as 16-byte chunks.  (The scalar kernel above moved 2 bytes per lane and
// instruction: 92.9 us per launch at B = 64, T = 128 = 3.6 TB/s for 335 MB.)
template <bool KV8>
__global__ __launch_bounds__(256) void rope_kv_vec_kernel(bf16_t* __restrict__ qkv, void* __restrict__ kc_, void* __restrict__ vc_,
                                                          const float* __restrict__ cos_tab, const float* __restrict__ sin_tab, int T_, int H, int d,
                                                          int ctx_max, int pos0, const int32_t* __restrict__ pos_dev) {
    const int64_t row = blockIdx.x;
    const int b = (int)(row / T_), t = (int)(row % T_);
    const int pos = pos0 + (pos_dev ? *pos_dev : 0) + t;
    const int half = d / 2, cph = half / 8;            // chunks of 8 pair indices per head
    const int HD = H * d;
    bf16_t* q = qkv + row * 3 * HD;
    bf16_t* k = q + HD;
    const bf16_t* v = k + HD;
    const float* cs = cos_tab + (int64_t)pos * half;
    const float* sn = sin_tab + (int64_t)pos * half;
    auto rot = [](const uint4& lo, const uint4& hi, const float* c, const float* s_, uint4& olo, uint4& ohi) {
        float a[8], b_[8];
        bf8_to_f32(lo, a); bf8_to_f32(hi, b_);
        float r0[8], r1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { r0[e] = a[e] * c[e] - b_[e] * s_[e]; r1[e] = b_[e] * c[e] + a[e] * s_[e]; }
        olo = make_uint4(pack_bf2(r0[0], r0[1]), pack_bf2(r0[2], r0[3]), pack_bf2(r0[4], r0[5]), pack_bf2(r0[6], r0[7]));
        ohi = make_uint4(pack_bf2(r1[0], r1[1]), pack_bf2(r1[2], r1[3]), pack_bf2(r1[4], r1[5]), pack_bf2(r1[6], r1[7]));
    };
    for (int idx = threadIdx.x; idx < H * cph; idx += 256) {
        const int h = idx / cph, i = (idx % cph) * 8;
        float c[8], s_[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { c[e] = Act<bf16_t>::rnd(cs[i + e]); s_[e] = Act<bf16_t>::rnd(sn[i + e]); }
        const int o = h * d + i;
        uint4 lo, hi;
        rot(*reinterpret_cast<const uint4*>(q + o), *reinterpret_cast<const uint4*>(q + o + half), c, s_, lo, hi);
        *reinterpret_cast<uint4*>(q + o) = lo;
        *reinterpret_cast<uint4*>(q + o + half) = hi;
        rot(*reinterpret_cast<const uint4*>(k + o), *reinterpret_cast<const uint4*>(k + o + half), c, s_, lo, hi);
        const int64_t ko = (((int64_t)b * H + h) * ctx_max + pos) * d + i;
        if constexpr (KV8) {
            *reinterpret_cast<uint4*>(k + o) = lo;
            *reinterpret_cast<uint4*>(k + o + half) = hi;
            unsigned char* kc8 = (unsigned char*)kc_;
            float f[8];
            bf8_to_f32(lo, f);
            unsigned w0 = VCLA_CVT_PK_FP8_SAT(f[0], f[1], 0, false); w0 = VCLA_CVT_PK_FP8_SAT(f[2], f[3], w0, true);
            unsigned w1 = VCLA_CVT_PK_FP8_SAT(f[4], f[5], 0, false); w1 = VCLA_CVT_PK_FP8_SAT(f[6], f[7], w1, true);
            *reinterpret_cast<uint2*>(kc8 + ko) = make_uint2(w0, w1);
            bf8_to_f32(hi, f);
            w0 = VCLA_CVT_PK_FP8_SAT(f[0], f[1], 0, false); w0 = VCLA_CVT_PK_FP8_SAT(f[2], f[3], w0, true);
            w1 = VCLA_CVT_PK_FP8_SAT(f[4], f[5], 0, false); w1 = VCLA_CVT_PK_FP8_SAT(f[6], f[7], w1, true);
            *reinterpret_cast<uint2*>(kc8 + ko + half) = make_uint2(w0, w1);
        } else {
            bf16_t* kc = (bf16_t*)kc_;
            *reinterpret_cast<uint4*>(kc + ko) = lo;
            *reinterpret_cast<uint4*>(kc + ko + half) = hi;
        }
    }
    for (int idx = threadIdx.x * 8; idx < HD; idx += 256 * 8) {
        const int h = idx / d, i = idx % d;
        const uint4 vv = *reinterpret_cast<const uint4*>(v + idx);
        const int64_t vo = (((int64_t)b * H + h) * ctx_max + pos) * d + i;
        if constexpr (KV8) {
            float f[8];
            bf8_to_f32(vv, f);
            unsigned w0 = VCLA_CVT_PK_FP8_SAT(f[0], f[1], 0, false); w0 = VCLA_CVT_PK_FP8_SAT(f[2], f[3], w0, true);
            unsigned w1 = VCLA_CVT_PK_FP8_SAT(f[4], f[5], 0, false); w1 = VCLA_CVT_PK_FP8_SAT(f[6], f[7], w1, true);
            *reinterpret_cast<uint2*>((unsigned char*)vc_ + vo) = make_uint2(w0, w1);
        } else {
            *reinterpret_cast<uint4*>((bf16_t*)vc_ + vo) = vv;
        }
    }
}

extern "C" int vcla_rope_kv_append(void* qkv, void* k_cache, void* v_cache, const float* cos_tab,
                                   const float* sin_tab, int B, int T, int H, int d, int ctx_max, int pos0,
                                   const int32_t* pos_dev, int dtype, void* stream) {
    const bool kv8 = (dtype & VCLA_KV_FP8) != 0;
    dtype &= ~VCLA_KV_FP8;
    VCLA_REQUIRE(dtype == VCLA_F32 || dtype == VCLA_BF16, VCLA_ERR_BAD_DTYPE, "rope_kv_append: bad dtype %d", dtype);
    VCLA_REQUIRE(!kv8 || dtype == VCLA_BF16, VCLA_ERR_BAD_DTYPE, "rope_kv_append: VCLA_KV_FP8 goes with VCLA_BF16 activations only");
    VCLA_REQUIRE(B >= 0 && T >= 0 && H > 0 && d > 0 && d % 2 == 0 && ctx_max > 0 && pos0 >= 0, VCLA_ERR_BAD_SHAPE,
                 "rope_kv_append: B=%d T=%d H=%d d=%d ctx_max=%d pos0=%d", B, T, H, d, ctx_max, pos0);
    VCLA_REQUIRE(pos_dev || pos0 + T <= ctx_max, VCLA_ERR_BAD_SHAPE, "rope_kv_append: pos0 %d + T %d > ctx_max %d",
                 pos0, T, ctx_max);
    VCLA_REQUIRE(qkv && k_cache && v_cache && cos_tab && sin_tab, VCLA_ERR_BAD_ARG, "rope_kv_append: null pointer");
    const int64_t rows = (int64_t)B * T;
    if (rows == 0) return VCLA_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VCLA_F32)
        rope_kv_kernel<float><<<(unsigned)rows, 256, 0, s>>>((float*)qkv, k_cache, v_cache, cos_tab,
                                                             sin_tab, T, H, d, ctx_max, pos0, pos_dev);
    else if (d % 16 == 0 && vcla_aligned(qkv, 16) && vcla_aligned(k_cache, 16) && vcla_aligned(v_cache, 16)) {      // bf16, 16-byte chunks
        if (kv8) rope_kv_vec_kernel<true><<<(unsigned)rows, 256, 0, s>>>((bf16_t*)qkv, k_cache, v_cache, cos_tab, sin_tab, T, H, d, ctx_max, pos0, pos_dev);
        else rope_kv_vec_kernel<false><<<(unsigned)rows, 256, 0, s>>>((bf16_t*)qkv, k_cache, v_cache, cos_tab, sin_tab, T, H, d, ctx_max, pos0, pos_dev);
    } else if (kv8)
        rope_kv_kernel<bf16_t, true><<<(unsigned)rows, 256, 0, s>>>((bf16_t*)qkv, k_cache, v_cache, cos_tab, sin_tab, T, H, d, ctx_max, pos0, pos_dev);
    else
        rope_kv_kernel<bf16_t><<<(unsigned)rows, 256, 0, s>>>((bf16_t*)qkv, k_cache, v_cache,
                                                              cos_tab, sin_tab, T, H, d, ctx_max, pos0, pos_dev);
    VCLA_CHECK_LAUNCH("rope_kv_kernel");
    return VCLA_OK;
}

// ------------------------------------------------------------------ argmax (first maximum, like torch.argmax)
#define ARGMAX_THREADS 1024
__global__ __launch_bounds__(ARGMAX_THREADS) void argmax_kernel(const float* __restrict__ logits, int64_t ld,
                                                                int64_t* __restrict__ out, int V) {
    __shared__ float sv[ARGMAX_THREADS / 64];
    __shared__ int si[ARGMAX_THREADS / 64];
    const float* x = logits + (int64_t)blockIdx.x * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    // 8 independent loads in flight per thread: the row is read once, latency is paid ~V / (8 * 1024) times
    for (int j0 = threadIdx.x; j0 < V; j0 += ARGMAX_THREADS * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * ARGMAX_THREADS;
            v[u] = j < V ? x[j] : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * ARGMAX_THREADS;
            if (v[u] > best) { best = v[u]; bi = j; }   // strictly greater: j increases, so ties keep the lowest index
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < ARGMAX_THREADS / 64; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        out[blockIdx.x] = (bi == 0x7fffffff) ? 0 : bi;
    }
}

extern "C" int vcla_argmax(const float* logits, int64_t ld, int64_t* ids_out, int B, int V, void* stream) {
    VCLA_REQUIRE(B >= 0 && V > 0 && ld >= V, VCLA_ERR_BAD_SHAPE, "argmax: B=%d V=%d ld=%lld", B, V, (long long)ld);
    VCLA_REQUIRE(logits && ids_out, VCLA_ERR_BAD_ARG, "argmax: null pointer");
    if (B == 0) return VCLA_OK;
    argmax_kernel<<<B, ARGMAX_THREADS, 0, (hipStream_t)stream>>>(logits, ld, ids_out, V);
    VCLA_CHECK_LAUNCH("argmax_kernel");
    return VCLA_OK;
}

// ------------------------------------------------------------------ causal-LM loss (forward(labels=...) -> .loss)
// hf:loss/loss_utils.py ForCausalLMLoss as reached from models/visualcla/modeling_visualcla.py:321-328: position t of every sequence is scored
// against labels[t + 1] (the last position has no target), rows whose target is ignore_index (-100) are skipped, mean over the rest.
// One workgroup per (sequence, position): log-sum-exp of the fp32 logits row (max pass + sum pass, the row stays in L2 between them),
// row_loss = lse - logit[target]; a second single-workgroup launch sums the rows in index order (deterministic) and divides by the count.
#define CE_THREADS 256
__global__ __launch_bounds__(CE_THREADS) void ce_rows_kernel(const float* __restrict__ logits, int64_t ld_row, const int64_t* __restrict__ labels, int T, int V,
                                                             int64_t ignore_index, float* __restrict__ row_loss) {
    __shared__ float red[8];
    const int row = blockIdx.x, b = row / T, t = row % T;
    const int64_t tgt = t + 1 < T ? labels[(int64_t)b * T + t + 1] : ignore_index;
    if (tgt == ignore_index || tgt < 0 || tgt >= V) {          // (out-of-range targets are refused by the host wrapper; skipped here to stay in bounds)
        if (threadIdx.x == 0) row_loss[row] = -1.0f;            // marker: not counted (a real row loss is >= 0, +inf or NaN)
        return;
    }
    const float* lr = logits + (int64_t)row * ld_row;
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < V; j += CE_THREADS) mx = fmaxf(mx, lr[j]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = threadIdx.x; j < V; j += CE_THREADS) sum += expf(lr[j] - mx);
    sum = block_sum_256(sum, red + 4);
    // a NaN / +inf logit anywhere in the row makes `sum` NaN (expf(NaN - mx), expf(inf - inf)) and the row loss NaN, as torch's log_softmax does:
    // the clamp below must not swallow it (fmaxf(NaN, 0) = 0 would count the row as a perfect prediction)
    const float l = mx + logf(sum) - lr[tgt];
    if (threadIdx.x == 0) row_loss[row] = l < 0.f ? 0.f : l;
}
__global__ __launch_bounds__(CE_THREADS) void ce_mean_kernel(const float* __restrict__ row_loss, int rows, float* __restrict__ out) {
    __shared__ float s_sum[CE_THREADS];
    __shared__ int s_cnt[CE_THREADS];
    // fixed assignment (thread i owns rows i, i + 256, ...) and a fixed-order tree: the same bits on every run
    float acc = 0.f;
    int cnt = 0;
    for (int r = threadIdx.x; r < rows; r += CE_THREADS) {
        const float v = row_loss[r];
        if (v != -1.0f) { acc += v; ++cnt; }          // -1.0f = "no target" marker (exactly; real row losses are >= 0, +inf or NaN -- all of which count)
    }
    s_sum[threadIdx.x] = acc;
    s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = CE_THREADS / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) { s_sum[threadIdx.x] += s_sum[threadIdx.x + o]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s_cnt[0] > 0 ? s_sum[0] / (float)s_cnt[0] : __builtin_nanf("");     // torch: mean over zero targets is nan
}

extern "C" int vcla_causal_lm_loss(const float* logits, int64_t ld_row, const int64_t* labels, int B, int T, int V, int64_t ignore_index,
                                   float* row_loss_ws, float* loss_out, void* stream) {
    VCLA_REQUIRE(logits && labels && row_loss_ws && loss_out, VCLA_ERR_BAD_ARG, "causal_lm_loss: null pointer");
    VCLA_REQUIRE(B > 0 && T > 0 && V > 0 && ld_row >= V, VCLA_ERR_BAD_SHAPE, "causal_lm_loss: B=%d T=%d V=%d ld=%lld", B, T, V, (long long)ld_row);
    hipStream_t s = (hipStream_t)stream;
    ce_rows_kernel<<<B * T, CE_THREADS, 0, s>>>(logits, ld_row, labels, T, V, ignore_index, row_loss_ws);
    VCLA_CHECK_LAUNCH("ce_rows_kernel");
    ce_mean_kernel<<<1, CE_THREADS, 0, s>>>(row_loss_ws, B * T, loss_out);
    VCLA_CHECK_LAUNCH("ce_mean_kernel");
    return VCLA_OK;
}
