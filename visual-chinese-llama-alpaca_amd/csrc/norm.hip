// norm.hip -- LayerNorm / RMSNorm / ViT embedding assembly.  HBM-bound row kernels:
// one 256-thread workgroup per row, the row is staged once through LDS as fp32 (one HBM read,
// one HBM write per element), statistics in fp32 with a two-pass variance.
#include "vcla_common.h"

#define NORM_MAX_COLS 8192

// load row -> LDS (fp32) with 4-wide vector loads when aligned
template <typename T>
__device__ __forceinline__ void stage_row(const T* __restrict__ x, float* row, int cols, bool vec_ok) {
    if (vec_ok) {
        for (int c = threadIdx.x * 4; c < cols; c += 256 * 4) {
            float v[4];
            Act<T>::ld4(x + c, v);
            row[c] = v[0]; row[c + 1] = v[1]; row[c + 2] = v[2]; row[c + 3] = v[3];
        }
    } else {
        for (int c = threadIdx.x; c < cols; c += 256) row[c] = Act<T>::ld(x + c);
    }
}

template <typename T>
__device__ __forceinline__ void layernorm_finish(float* row, float* red, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, T* __restrict__ y, int cols,
                                                 float eps, bool vec_ok) {
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) s += row[c];
    const float mean = block_sum_256(s, red) / (float)cols;
    float q = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float d = row[c] - mean;
        q += d * d;
    }
    const float var = block_sum_256(q, red) / (float)cols;
    const float rstd = rsqrtf(var + eps);
    if (vec_ok) {
        for (int c = threadIdx.x * 4; c < cols; c += 256 * 4) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (row[c + e] - mean) * rstd * gamma[c + e] + beta[c + e];
            Act<T>::st4(y + c, v);
        }
    } else {
        for (int c = threadIdx.x; c < cols; c += 256)
            Act<T>::st(y + c, (row[c] - mean) * rstd * gamma[c] + beta[c]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ y,
                                                        int64_t ldy, int cols, float eps, int vec_ok) {
    __shared__ float row[NORM_MAX_COLS];
    __shared__ float red[8];
    const int64_t r = blockIdx.x;
    stage_row<T>(x + r * ldx, row, cols, vec_ok);
    __syncthreads();
    layernorm_finish<T>(row, red, gamma, beta, y + r * ldy, cols, eps, vec_ok);
}

// bf16 rows of 512 / 1024 / 1536 / 2048 columns (the ViT and resampler widths): one WAVE per row, 4 rows per workgroup, the
// row lives in registers (8 elements per lane per 512 columns, 16-byte loads / stores), statistics by wave shuffles -- no LDS
// staging and no block barriers.  31 -> ~15 us for the 16448 x 1024 ViT activations (the generic kernel spends its time in
// three barriers per row).
template <int CH>   // 512-column chunks per row
__global__ __launch_bounds__(256) void layernorm_wave_kernel(const bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, bf16_t* __restrict__ y, int64_t ldy,
                                                             int rows, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    constexpr int cols = CH * 512;
    float v[CH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        bf8_to_f32(*reinterpret_cast<const uint4*>(x + r * ldx + c * 512 + lane * 8), v[c]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[c][e];
    }
    const float mean = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int k = c * 512 + lane * 8;
        const float4 g0 = *reinterpret_cast<const float4*>(gamma + k), g1 = *reinterpret_cast<const float4*>(gamma + k + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(beta + k), b1 = *reinterpret_cast<const float4*>(beta + k + 4);
        const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[c][e] - mean) * rstd * gm[e] + bt[e];
        const uint4 pk = make_uint4(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7]));
        *reinterpret_cast<uint4*>(y + r * ldy + k) = pk;
    }
}

// LlamaRMSNorm: fp32 statistics; normalised value rounded to the activation dtype BEFORE the gain multiply
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const T* __restrict__ x, int64_t ldx,
                                                      const float* __restrict__ gamma, T* __restrict__ y,
                                                      int64_t ldy, int cols, float eps, int vec_ok) {
    __shared__ float row[NORM_MAX_COLS];
    __shared__ float red[8];
    const int64_t r = blockIdx.x;
    stage_row<T>(x + r * ldx, row, cols, vec_ok);
    __syncthreads();
    float q = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) q += row[c] * row[c];
    const float rstd = rsqrtf(block_sum_256(q, red) / (float)cols + eps);
    T* yr = y + r * ldy;
    if (vec_ok) {
        for (int c = threadIdx.x * 4; c < cols; c += 256 * 4) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gamma[c + e] * Act<T>::rnd(row[c + e] * rstd);
            Act<T>::st4(yr + c, v);
        }
    } else {
        for (int c = threadIdx.x; c < cols; c += 256) Act<T>::st(yr + c, gamma[c] * Act<T>::rnd(row[c] * rstd));
    }
}


// LlamaRMSNorm of <= 64 bf16 rows, output in the MFMA-fragment-major layout the streaming decode GEMM reads
// (vcla_gemm_args.A_frag: [cols/32][MT][64 lanes][8], element (m, k) in fragment (k/32, m/16), lane ((k%32)/8)*16 + m%16,
// slot k%8).  Staging, statistics and rounding follow rmsnorm_kernel instruction for instruction (bit-identical values); only
// the store address differs: a thread's 8 consecutive columns are one 16-byte fragment slot.  gamma == NULL: plain re-layout.
__global__ __launch_bounds__(256) void rmsnorm_pack_kernel(const bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                           bf16_t* __restrict__ y, int cols, int mt, float eps, int vec_ok) {
    __shared__ float row[NORM_MAX_COLS];
    __shared__ float red[8];
    const int r = blockIdx.x;
    stage_row<bf16_t>(x + (int64_t)r * ldx, row, cols, vec_ok);
    __syncthreads();
    float rstd = 1.f;
    if (gamma) {
        float q = 0.f;
        for (int c = threadIdx.x; c < cols; c += 256) q += row[c] * row[c];
        rstd = rsqrtf(block_sum_256(q, red) / (float)cols + eps);
    }
    for (int c = threadIdx.x * 8; c < cols; c += 256 * 8) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gamma ? gamma[c + e] * Act<bf16_t>::rnd(row[c + e] * rstd) : row[c + e];
        bf16_t* dst = y + ((((int64_t)(c >> 5) * mt + (r >> 4)) * 64 + ((c & 31) >> 3) * 16 + (r & 15)) << 3);
        *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
    }
}


// Per-row dynamic fp8 (OCP e4m3fn) quantisation of bf16 activations for the fp8 MFMA GEMM: one WAVE per row for rows of up to
// 16384 columns held in registers would cost 256 VGPRs; instead one 256-thread workgroup per row, the row staged once in LDS
// as fp32 (one HBM read), absmax by shuffles, then 16 values -> one 16-byte store per thread.
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const bf16_t* __restrict__ x, int64_t ldx, unsigned char* __restrict__ q,
                                                             float* __restrict__ scale, int cols) {
    extern __shared__ __attribute__((aligned(16))) float qrow[];   // [cols]
    __shared__ float red[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    const bf16_t* xr = x + (int64_t)r * ldx;
    float mx = 0.f;
    for (int c = tid * 8; c < cols; c += 256 * 8) {
        float v[8];
        bf8_to_f32(*reinterpret_cast<const uint4*>(xr + c), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { qrow[c + e] = v[e]; mx = fmaxf(mx, fabsf(v[e])); }
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = fmaxf(mx / 448.0f, 1e-20f);
    const float inv = 1.0f / sc;
    if (tid == 0) scale[r] = sc;
    unsigned char* qr = q + (int64_t)r * cols;
    for (int c = tid * 16; c < cols; c += 256 * 16) {
        uint32_t w[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float f[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) f[e] = fminf(fmaxf(qrow[c + g * 4 + e] * inv, -448.0f), 448.0f);
            int pk = 0;
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], pk, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], pk, true);
            w[g] = (uint32_t)pk;
        }
        *reinterpret_cast<uint4*>(qr + c) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ViT input assembly: row (b, n): n == 0 -> class embedding, else patch embed (b, n-1); + position emb; pre-LN
template <typename T>
__global__ __launch_bounds__(256) void vit_assemble_kernel(const T* __restrict__ patch, const float* __restrict__ cls,
                                                           const float* __restrict__ pos,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, T* __restrict__ y, int np,
                                                           int D, float eps, int vec_ok) {
    __shared__ float row[NORM_MAX_COLS];
    __shared__ float red[8];
    const int N = np + 1;
    const int b = blockIdx.x / N, n = blockIdx.x % N;
    const float* pr = pos + (int64_t)n * D;
    if (n == 0) {
        for (int c = threadIdx.x; c < D; c += 256) row[c] = Act<T>::rnd(cls[c] + pr[c]);
    } else {
        const T* src = patch + ((int64_t)b * np + (n - 1)) * D;
        for (int c = threadIdx.x; c < D; c += 256) row[c] = Act<T>::rnd(Act<T>::ld(src + c) + pr[c]);
    }
    __syncthreads();
    layernorm_finish<T>(row, red, gamma, beta, y + (int64_t)blockIdx.x * D, D, eps, vec_ok);
}

// ------------------------------------------------------------------ host entry points
static inline bool vec4_ok(const void* p, int64_t ld, int cols, int dtype) {
    const size_t a = dtype == VCLA_F32 ? 16 : 8;
    return cols % 4 == 0 && ld % 4 == 0 && vcla_aligned(p, a);
}

extern "C" int vcla_layernorm(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y,
                              int64_t ldy, int rows, int cols, float eps, int dtype, void* stream) {
    VCLA_REQUIRE(dtype == VCLA_F32 || dtype == VCLA_BF16, VCLA_ERR_BAD_DTYPE, "layernorm: bad dtype %d", dtype);
    VCLA_REQUIRE(rows >= 0 && cols > 0 && cols <= NORM_MAX_COLS, VCLA_ERR_BAD_SHAPE,
                 "layernorm: rows=%d cols=%d (max cols %d)", rows, cols, NORM_MAX_COLS);
    VCLA_REQUIRE(x && y && gamma && beta, VCLA_ERR_BAD_ARG, "layernorm: null pointer");
    if (rows == 0) return VCLA_OK;
    hipStream_t s = (hipStream_t)stream;
    const int v = vec4_ok(x, ldx, cols, dtype) && vec4_ok(y, ldy, cols, dtype);
    if (dtype == VCLA_BF16 && cols % 512 == 0 && cols <= 2048 && ldx % 8 == 0 && ldy % 8 == 0 && vcla_aligned(x, 16) && vcla_aligned(y, 16) &&
        vcla_aligned(gamma, 16) && vcla_aligned(beta, 16)) {
        const unsigned blocks = (unsigned)((rows + 3) / 4);
        const bf16_t* xb = (const bf16_t*)x;
        bf16_t* yb = (bf16_t*)y;
        switch (cols / 512) {
            case 1: layernorm_wave_kernel<1><<<blocks, 256, 0, s>>>(xb, ldx, gamma, beta, yb, ldy, rows, eps); break;
            case 2: layernorm_wave_kernel<2><<<blocks, 256, 0, s>>>(xb, ldx, gamma, beta, yb, ldy, rows, eps); break;
            case 3: layernorm_wave_kernel<3><<<blocks, 256, 0, s>>>(xb, ldx, gamma, beta, yb, ldy, rows, eps); break;
            default: layernorm_wave_kernel<4><<<blocks, 256, 0, s>>>(xb, ldx, gamma, beta, yb, ldy, rows, eps); break;
        }
        VCLA_CHECK_LAUNCH("layernorm_wave_kernel");
        return VCLA_OK;
    }
    if (dtype == VCLA_F32)
        layernorm_kernel<float><<<rows, 256, 0, s>>>((const float*)x, ldx, gamma, beta, (float*)y, ldy, cols, eps, v);
    else
        layernorm_kernel<bf16_t><<<rows, 256, 0, s>>>((const bf16_t*)x, ldx, gamma, beta, (bf16_t*)y, ldy, cols, eps, v);
    VCLA_CHECK_LAUNCH("layernorm_kernel");
    return VCLA_OK;
}

extern "C" int vcla_rmsnorm(const void* x, int64_t ldx, const float* gamma, void* y, int64_t ldy, int rows,
                            int cols, float eps, int dtype, void* stream) {
    VCLA_REQUIRE(dtype == VCLA_F32 || dtype == VCLA_BF16, VCLA_ERR_BAD_DTYPE, "rmsnorm: bad dtype %d", dtype);
    VCLA_REQUIRE(rows >= 0 && cols > 0 && cols <= NORM_MAX_COLS, VCLA_ERR_BAD_SHAPE,
                 "rmsnorm: rows=%d cols=%d (max cols %d)", rows, cols, NORM_MAX_COLS);
    VCLA_REQUIRE(x && y && gamma, VCLA_ERR_BAD_ARG, "rmsnorm: null pointer");
    if (rows == 0) return VCLA_OK;
    hipStream_t s = (hipStream_t)stream;
    const int v = vec4_ok(x, ldx, cols, dtype) && vec4_ok(y, ldy, cols, dtype);
    if (dtype == VCLA_F32)
        rmsnorm_kernel<float><<<rows, 256, 0, s>>>((const float*)x, ldx, gamma, (float*)y, ldy, cols, eps, v);
    else
        rmsnorm_kernel<bf16_t><<<rows, 256, 0, s>>>((const bf16_t*)x, ldx, gamma, (bf16_t*)y, ldy, cols, eps, v);
    VCLA_CHECK_LAUNCH("rmsnorm_kernel");
    return VCLA_OK;
}


extern "C" int vcla_rmsnorm_pack(const void* x, int64_t ldx, const float* gamma, void* y_frag, int rows, int cols, float eps,
                                 void* stream) {
    VCLA_REQUIRE(rows >= 0 && rows <= 64 && cols > 0 && cols <= NORM_MAX_COLS && cols % 32 == 0, VCLA_ERR_BAD_SHAPE,
                 "rmsnorm_pack: rows=%d (max 64) cols=%d (multiple of 32, max %d)", rows, cols, NORM_MAX_COLS);
    VCLA_REQUIRE(x && y_frag && vcla_aligned(y_frag, 16), VCLA_ERR_BAD_ARG, "rmsnorm_pack: null / misaligned pointer");
    if (rows == 0) return VCLA_OK;
    const int v = vec4_ok(x, ldx, cols, VCLA_BF16);
    rmsnorm_pack_kernel<<<rows, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, gamma, (bf16_t*)y_frag, cols, (rows + 15) / 16, eps, v);
    VCLA_CHECK_LAUNCH("rmsnorm_pack_kernel");
    return VCLA_OK;
}


extern "C" int vcla_quant_fp8_rows(const void* x, int64_t ldx, void* q, float* scale, int rows, int cols, void* stream) {
    VCLA_REQUIRE(rows >= 0 && cols > 0 && cols % 16 == 0 && cols <= 16128 && ldx % 8 == 0, VCLA_ERR_BAD_SHAPE,   // cols * 4 B of dynamic LDS + 16 B static <= 64 KiB
                 "quant_fp8_rows: rows=%d cols=%d (multiple of 16, max 16128), ldx=%lld", rows, cols, (long long)ldx);
    VCLA_REQUIRE(x && q && scale && vcla_aligned(x, 16) && vcla_aligned(q, 16), VCLA_ERR_BAD_ARG, "quant_fp8_rows: null / misaligned pointer");
    if (rows == 0) return VCLA_OK;
    quant_fp8_rows_kernel<<<rows, 256, (size_t)cols * 4, (hipStream_t)stream>>>((const bf16_t*)x, ldx, (unsigned char*)q, scale, cols);
    VCLA_CHECK_LAUNCH("quant_fp8_rows_kernel");
    return VCLA_OK;
}

extern "C" int vcla_vit_assemble(const void* patch_embeds, const float* cls, const float* pos, const float* gamma,
                                 const float* beta, void* y, int B, int np, int D, float eps, int dtype,
                                 void* stream) {
    VCLA_REQUIRE(dtype == VCLA_F32 || dtype == VCLA_BF16, VCLA_ERR_BAD_DTYPE, "vit_assemble: bad dtype %d", dtype);
    VCLA_REQUIRE(B >= 0 && np > 0 && D > 0 && D <= NORM_MAX_COLS, VCLA_ERR_BAD_SHAPE,
                 "vit_assemble: B=%d np=%d D=%d", B, np, D);
    VCLA_REQUIRE(patch_embeds && cls && pos && gamma && beta && y, VCLA_ERR_BAD_ARG, "vit_assemble: null pointer");
    if (B == 0) return VCLA_OK;
    hipStream_t s = (hipStream_t)stream;
    const int rows = B * (np + 1);
    const int v = vec4_ok(y, D, D, dtype);
    if (dtype == VCLA_F32)
        vit_assemble_kernel<float><<<rows, 256, 0, s>>>((const float*)patch_embeds, cls, pos, gamma, beta, (float*)y,
                                                        np, D, eps, v);
    else
        vit_assemble_kernel<bf16_t><<<rows, 256, 0, s>>>((const bf16_t*)patch_embeds, cls, pos, gamma, beta,
                                                         (bf16_t*)y, np, D, eps, v);
    VCLA_CHECK_LAUNCH("vit_assemble_kernel");
    return VCLA_OK;
}
