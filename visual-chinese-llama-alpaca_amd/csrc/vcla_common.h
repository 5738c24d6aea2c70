// vcla_common.h -- shared device/host helpers for libvisualcla_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/visualcla_hip.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
// native 16-byte vector: use this (not HIP's uint4 struct) for register staging -- struct copies become memcpy in IR and
// memcpy-only private arrays are not promoted to registers (they end up in scratch)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// ------------------------------------------------------------------ error handling (host)
void vcla_set_error(const char* fmt, ...);
int vcla_fail(int code, const char* fmt, ...);

#define VCLA_CHECK_HIP(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return vcla_fail(VCLA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                             __FILE__, __LINE__);                                             \
    } while (0)

#define VCLA_CHECK_LAUNCH(name)                                                               \
    do {                                                                                      \
        hipError_t _e = hipGetLastError();                                                    \
        if (_e != hipSuccess)                                                                 \
            return vcla_fail(VCLA_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(_e)); \
    } while (0)

#define VCLA_REQUIRE(cond, code, ...)                                                         \
    do {                                                                                      \
        if (!(cond)) return vcla_fail(code, __VA_ARGS__);                                     \
    } while (0)

// sample.hip: the launch behind vcla_sample, shared with the decode loop
int vcla_sample_launch(float* logits, int64_t ld, int B, int V, int n_hist, const int32_t* n_hist_dev, const vcla_sample_args* a,
                       int64_t* out, hipStream_t s);

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): raise it once per device a kernel instantiation
// is launched on (`done` = a static flag array of that instantiation), not once per process
#define VCLA_MAX_DEVICES 32
static inline int vcla_raise_dyn_lds(const void* kern, size_t bytes, bool (&done)[VCLA_MAX_DEVICES]) {
    int dev = 0;
    VCLA_CHECK_HIP(hipGetDevice(&dev));
    const bool tracked = dev >= 0 && dev < VCLA_MAX_DEVICES;
    if (!tracked || !done[dev]) {
        VCLA_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        if (tracked) done[dev] = true;
    }
    return VCLA_OK;
}

static inline size_t vcla_dtype_size(int dtype) { return dtype == VCLA_F32 ? 4 : 2; }
static inline bool vcla_aligned(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }

// ------------------------------------------------------------------ device helpers
#if defined(__HIPCC__)

__device__ __forceinline__ bool vcla_aligned_dev(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// fp32 -> bf16, round-to-nearest-even (same rounding as torch.bfloat16 conversion).  The native __bf16 cast lowers to
// gfx950's v_cvt_pk_bf16_f32: branch-free, one instruction per pair.
__device__ __forceinline__ bf16_t f2bf(float f) {
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// activation-dtype I/O: T = float or bf16_t
template <typename T> struct Act;
template <> struct Act<float> {
    static constexpr int kDtype = VCLA_F32;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    __device__ static __forceinline__ float rnd(float v) { return v; }
    // 4 contiguous elements (p must be 16-byte aligned)
    __device__ static __forceinline__ void ld4(const float* p, float* v) {
        float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ static __forceinline__ void st4(float* p, const float* v) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct Act<bf16_t> {
    static constexpr int kDtype = VCLA_BF16;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
    __device__ static __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }
    // 4 contiguous elements (p must be 8-byte aligned)
    __device__ static __forceinline__ void ld4(const bf16_t* p, float* v) {
        uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    }
    __device__ static __forceinline__ void st4(bf16_t* p, const float* v) {
        *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
    }
};

// 8 bf16 (one uint4) -> 8 floats
__device__ __forceinline__ void bf8_to_f32(const uint4& t, float* v) {
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
    v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
}

// fp32 pair -> two OCP e4m3fn bytes, SATURATING: v_cvt_pk_fp8_f32 does not clamp (|x| > 448 would encode as NaN, and one NaN byte in
// the K/V cache poisons every later step of that (sequence, head)), so the operands go through v_med3_f32 first -- the same clamp
// quant_fp8_rows applies to the activation rows
#define VCLA_CVT_PK_FP8_SAT(a, b, old, hi)                                                                    \
    __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f((a), 448.0f, -448.0f), __builtin_amdgcn_fmed3f((b), 448.0f, -448.0f), (old), (hi))

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide reductions for 256-thread blocks (4 waves); `red` is a __shared__ float[8]
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();  // protect `red` against a previous use
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// x * sigmoid(a x) with the hardware reciprocal (1 ulp) instead of an IEEE division (~10 instructions): the activation
// epilogue of a 256x256 tile is 128 values per lane
__device__ __forceinline__ float act_quick_gelu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float act_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float act_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

#endif  // __HIPCC__
