// attention.hip -- softmax(scale * Q K^T + mask) V.
//
// Kernel 1 (generic, fp32 math, any strides): one wave per query row; lane-per-key dot products for the
// scores, scores kept in LDS, lane-per-dimension accumulation for P V.  Used by the fp32 parity mode, by
// decode (Tq = 1) and as the checker for the MFMA flash kernel.
// Kernel 2 (MFMA flash attention, bf16): see attention_mfma.hip.
#include "vcla_common.h"

#define ATT_ROWS_PER_BLOCK 16

int vcla_attention_mfma(const vcla_attn_args* a, void* stream);  // attention_mfma.hip
int vcla_attention_vit(const vcla_attn_args* a, void* stream);   // attention_mfma.hip: whole-sequence ViT self-attention (force_kernel 3)
bool vcla_attention_mfma_supported(const vcla_attn_args* a);

// dot of q (fp32 in LDS) with one key row in global memory
template <typename T, int D> struct KeyDot;
template <int D> struct KeyDot<float, D> {
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
template <int D> struct KeyDot<bf16_t, D> {
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

template <typename T> struct Pair;
template <> struct Pair<float> {
    __device__ static __forceinline__ void ld(const float* p, float& a, float& b) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        a = t.x; b = t.y;
    }
    __device__ static __forceinline__ void st(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
};
template <> struct Pair<bf16_t> {
    __device__ static __forceinline__ void ld(const bf16_t* p, float& a, float& b) {
        const uint32_t t = *reinterpret_cast<const uint32_t*>(p);
        a = __uint_as_float(t << 16); b = __uint_as_float(t & 0xffff0000u);
    }
    __device__ static __forceinline__ void st(bf16_t* p, float a, float b) { *reinterpret_cast<uint32_t*>(p) = pack_bf2(a, b); }
};

template <typename T, int D>
__global__ __launch_bounds__(256) void attn_generic_kernel(vcla_attn_args a, int tk_cap) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* qs = smem + wave * D;                      // [4][D]
    float* sc = smem + 4 * D + (int64_t)wave * tk_cap;  // [4][tk_cap]
    const int b = blockIdx.z, h = blockIdx.y;
    const int Tk = a.tk_dev ? (*a.tk_dev + a.tk_dev_add) : a.Tk;
    const T* qb = (const T*)a.q + b * a.q_bs + h * a.q_hs;
    const T* kb = (const T*)a.k + b * a.k_bs + h * a.k_hs;
    const T* vb = (const T*)a.v + b * a.v_bs + h * a.v_hs;
    T* ob = (T*)a.o + b * a.o_bs + h * a.o_hs;
    const int32_t* km = a.key_mask ? a.key_mask + b * a.key_mask_ld : nullptr;
    constexpr int HALF = D / 2;         // lanes per key group in the PV phase
    constexpr int GROUPS = 64 / HALF;   // keys processed concurrently in the PV phase
    const int grp = lane / HALF, dl = lane % HALF;

    for (int s = 0; s < ATT_ROWS_PER_BLOCK / 4; ++s) {
        const int i = blockIdx.x * ATT_ROWS_PER_BLOCK + s * 4 + wave;
        if (i >= a.Tq) break;  // wave-uniform
        int kv_len = Tk;
        if (a.causal) {
            const int lim = i + (Tk - a.Tq) + 1;
            kv_len = lim < Tk ? lim : Tk;
        }
        // stage q
        for (int c = lane; c < D; c += 64) qs[c] = Act<T>::ld(qb + i * a.q_rs + c);
        __builtin_amdgcn_wave_barrier();
        // scores
        float mx = -INFINITY;
        for (int j0 = 0; j0 < kv_len; j0 += 64) {
            const int j = j0 + lane;
            float sv = -INFINITY;
            if (j < kv_len && (!km || km[j] != 0)) sv = a.scale * KeyDot<T, D>::dot(qs, kb + j * a.k_rs);
            if (j < kv_len) sc[j] = sv;
            mx = fmaxf(mx, sv);
        }
        mx = wave_max(mx);
        float l = 0.f;
        if (mx > -INFINITY) {
            for (int j = lane; j < kv_len; j += 64) {
                const float e = __expf(sc[j] - mx);
                sc[j] = e;
                l += e;
            }
        }
        l = wave_sum(l);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        // probabilities are rounded to the activation dtype before P V (HF: softmax(...).to(query.dtype))
        for (int j = lane; j < kv_len; j += 64) sc[j] = (mx > -INFINITY) ? Act<T>::rnd(sc[j] * inv) : 0.f;
        __builtin_amdgcn_wave_barrier();
        // P V : lane -> (key group grp, dims 2*dl, 2*dl+1)
        float o0 = 0.f, o1 = 0.f;
        int j = grp;
        for (; j + 3 * GROUPS < kv_len; j += 4 * GROUPS) {
            float x0, y0, x1, y1, x2, y2, x3, y3;
            Pair<T>::ld(vb + (j) * a.v_rs + 2 * dl, x0, y0);
            Pair<T>::ld(vb + (j + GROUPS) * a.v_rs + 2 * dl, x1, y1);
            Pair<T>::ld(vb + (j + 2 * GROUPS) * a.v_rs + 2 * dl, x2, y2);
            Pair<T>::ld(vb + (j + 3 * GROUPS) * a.v_rs + 2 * dl, x3, y3);
            const float p0 = sc[j], p1 = sc[j + GROUPS], p2 = sc[j + 2 * GROUPS], p3 = sc[j + 3 * GROUPS];
            o0 += p0 * x0 + p1 * x1 + p2 * x2 + p3 * x3;
            o1 += p0 * y0 + p1 * y1 + p2 * y2 + p3 * y3;
        }
        for (; j < kv_len; j += GROUPS) {
            float x0, y0;
            Pair<T>::ld(vb + j * a.v_rs + 2 * dl, x0, y0);
            const float p0 = sc[j];
            o0 += p0 * x0;
            o1 += p0 * y0;
        }
#pragma unroll
        for (int off = HALF; off < 64; off <<= 1) {
            o0 += __shfl_xor(o0, off, 64);
            o1 += __shfl_xor(o1, off, 64);
        }
        if (grp == 0) Pair<T>::st(ob + i * a.o_rs + 2 * dl, o0, o1);
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename T, int D>
static int launch_generic(const vcla_attn_args* a, hipStream_t s) {
    const int tk_cap = (a->Tk + 63) & ~63;
    const size_t lds = (size_t)(4 * D + 4 * (size_t)tk_cap) * sizeof(float);
    VCLA_REQUIRE(lds <= 160 * 1024, VCLA_ERR_BAD_SHAPE, "attention: Tk=%d needs %zu B of LDS (max 160 KiB)", a->Tk, lds);
    auto kern = attn_generic_kernel<T, D>;
    if (lds > 64 * 1024)
        VCLA_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((a->Tq + ATT_ROWS_PER_BLOCK - 1) / ATT_ROWS_PER_BLOCK, a->H, a->B);
    kern<<<grid, 256, lds, s>>>(*a, tk_cap);
    VCLA_CHECK_LAUNCH("attn_generic_kernel");
    return VCLA_OK;
}

extern "C" int vcla_attention(const vcla_attn_args* a, int dtype, void* stream) {
    VCLA_REQUIRE(a, VCLA_ERR_BAD_ARG, "attention: null args");
    VCLA_REQUIRE(dtype == VCLA_F32 || dtype == VCLA_BF16, VCLA_ERR_BAD_DTYPE, "attention: bad dtype %d", dtype);
    VCLA_REQUIRE(a->D == 32 || a->D == 64 || a->D == 128, VCLA_ERR_BAD_SHAPE, "attention: head dim %d not in {32,64,128}", a->D);
    VCLA_REQUIRE(a->B >= 0 && a->H > 0 && a->Tq >= 0 && a->Tk > 0, VCLA_ERR_BAD_SHAPE, "attention: B=%d H=%d Tq=%d Tk=%d",
                 a->B, a->H, a->Tq, a->Tk);
    VCLA_REQUIRE(a->q && a->k && a->v && a->o, VCLA_ERR_BAD_ARG, "attention: null pointer");
    VCLA_REQUIRE(!a->tk_dev || a->Tq == 1, VCLA_ERR_BAD_ARG, "attention: tk_dev requires Tq == 1");
    // vector-load alignment: rows must start on 16-byte boundaries
    const int64_t va = dtype == VCLA_F32 ? 4 : 8;
    VCLA_REQUIRE(a->k_rs % va == 0 && a->k_hs % va == 0 && a->k_bs % va == 0 && vcla_aligned(a->k, 16), VCLA_ERR_BAD_SHAPE,
                 "attention: K rows must be 16-byte aligned");
    VCLA_REQUIRE(a->v_rs % 2 == 0 && a->v_hs % 2 == 0 && a->v_bs % 2 == 0 && a->o_rs % 2 == 0 && a->o_hs % 2 == 0 &&
                     a->o_bs % 2 == 0, VCLA_ERR_BAD_SHAPE, "attention: V/O strides must be even");
    if (a->B == 0 || a->Tq == 0) return VCLA_OK;
    hipStream_t s = (hipStream_t)stream;
    if (a->force_kernel == 3) {
        VCLA_REQUIRE(dtype == VCLA_BF16, VCLA_ERR_BAD_ARG, "attention: the whole-sequence ViT kernel needs bf16 activations");
        return vcla_attention_vit(a, stream);
    }
    if (a->force_kernel == 2 || (a->force_kernel == 0 && dtype == VCLA_BF16 && vcla_attention_mfma_supported(a))) {
        VCLA_REQUIRE(dtype == VCLA_BF16 && vcla_attention_mfma_supported(a), VCLA_ERR_BAD_ARG,
                     "attention: MFMA kernel does not support this problem");
        return vcla_attention_mfma(a, stream);
    }
#define ATT_CASE(TT, DD) return launch_generic<TT, DD>(a, s)
    if (dtype == VCLA_F32) {
        if (a->D == 32) ATT_CASE(float, 32);
        if (a->D == 64) ATT_CASE(float, 64);
        ATT_CASE(float, 128);
    } else {
        if (a->D == 32) ATT_CASE(bf16_t, 32);
        if (a->D == 64) ATT_CASE(bf16_t, 64);
        ATT_CASE(bf16_t, 128);
    }
#undef ATT_CASE
}
