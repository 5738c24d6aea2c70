// preprocess.hip -- next-row N1: CLIP image preprocessing on the GPU (what the reference does on the host with
// PIL + numpy per image: models/visualcla/modeling_utils.py:130,150-152 -> transformers CLIPImageProcessor).
//
// uint8 HWC RGB image -> bicubic resize of the shortest edge (Pillow's fixed-point separable resampler, bit-exact: 22-bit
// coefficients precomputed on the host, accumulate from 1<<21, shift, clip to uint8 after each pass) -> centre crop ->
// * 1/255 -> (x - mean) / std -> [3, S, S] in the activation dtype, written straight into the batch tensor the patch
// embedding reads.  Only the cropped S x S window is ever computed.  Integer / byte work, HBM-bound, two small kernels.
#include "vcla_common.h"

// horizontal pass: tmp[y][x][c] for every source row y and the S cropped output columns
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ img, int W, uint8_t* __restrict__ tmp, int S,
                                                       const int32_t* __restrict__ lo, const int32_t* __restrict__ cnt,
                                                       const int32_t* __restrict__ k, int kmax) {
    const int y = blockIdx.x;
    img += (int64_t)blockIdx.y * gridDim.x * W * 3;          // image blockIdx.y of the batch (gridDim.x = H)
    tmp += (int64_t)blockIdx.y * gridDim.x * S * 3;
    const uint8_t* row = img + (int64_t)y * W * 3;
    for (int idx = threadIdx.x; idx < S * 3; idx += 256) {
        const int x = idx / 3, c = idx % 3;
        const int x0 = lo[x], n = cnt[x];
        const int32_t* kk = k + (int64_t)x * kmax;
        int acc = 1 << 21;
        for (int t = 0; t < n; ++t) acc += (int)row[(x0 + t) * 3 + c] * kk[t];
        acc >>= 22;
        tmp[((int64_t)y * S + x) * 3 + c] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
    }
}

// vertical pass + rescale + normalise + HWC -> CHW
template <typename T>
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const uint8_t* __restrict__ tmp, int S, const int32_t* __restrict__ lo,
                                                            const int32_t* __restrict__ cnt, const int32_t* __restrict__ k, int kmax,
                                                            double rescale, float m0, float m1, float m2, float s0, float s1, float s2,
                                                            T* __restrict__ out, int H) {
    const int y = blockIdx.x;
    tmp += (int64_t)blockIdx.y * H * S * 3;                  // image blockIdx.y of the batch
    out += (int64_t)blockIdx.y * 3 * S * S;
    const int y0 = lo[y], n = cnt[y];
    const int32_t* kk = k + (int64_t)y * kmax;
    for (int idx = threadIdx.x; idx < S * 3; idx += 256) {
        const int x = idx / 3, c = idx % 3;
        int acc = 1 << 21;
        for (int t = 0; t < n; ++t) acc += (int)tmp[((int64_t)(y0 + t) * S + x) * 3 + c] * kk[t];
        acc >>= 22;
        acc = acc < 0 ? 0 : (acc > 255 ? 255 : acc);
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        float v;
        {
#pragma clang fp contract(off)   // numpy rounds after the multiply and after the subtract: an FMA here is 1 ulp off
            const float scaled = (float)((double)acc * rescale);   // HF rescales in float64, then rounds to float32
            const float centred = scaled - mean;
            v = centred / sd;   // hipcc's default fp32 divide is correctly rounded
        }
        Act<T>::st(out + ((int64_t)c * S + y) * S + x, v);
    }
}

extern "C" int vcla_image_preprocess_batch(const uint8_t* imgs, int N, int H, int W, uint8_t* tmp, int S, const int32_t* h_lo,
                                           const int32_t* h_cnt, const int32_t* h_k, int h_kmax, const int32_t* v_lo,
                                           const int32_t* v_cnt, const int32_t* v_k, int v_kmax, double rescale, const float* mean3,
                                           const float* std3, void* out, int dtype, void* stream) {
    VCLA_REQUIRE(dtype == VCLA_F32 || dtype == VCLA_BF16, VCLA_ERR_BAD_DTYPE, "image_preprocess: bad dtype %d", dtype);
    VCLA_REQUIRE(N >= 0 && N <= 65535 && H > 0 && W > 0 && S > 0 && h_kmax > 0 && v_kmax > 0, VCLA_ERR_BAD_SHAPE,
                 "image_preprocess: N=%d H=%d W=%d S=%d", N, H, W, S);
    VCLA_REQUIRE(imgs && tmp && h_lo && h_cnt && h_k && v_lo && v_cnt && v_k && mean3 && std3 && out, VCLA_ERR_BAD_ARG,
                 "image_preprocess: null pointer");
    if (N == 0) return VCLA_OK;
    hipStream_t s = (hipStream_t)stream;
    resize_h_kernel<<<dim3(H, N), 256, 0, s>>>(imgs, W, tmp, S, h_lo, h_cnt, h_k, h_kmax);
    VCLA_CHECK_LAUNCH("resize_h_kernel");
    if (dtype == VCLA_F32)
        resize_v_norm_kernel<float><<<dim3(S, N), 256, 0, s>>>(tmp, S, v_lo, v_cnt, v_k, v_kmax, rescale, mean3[0], mean3[1], mean3[2],
                                                               std3[0], std3[1], std3[2], (float*)out, H);
    else
        resize_v_norm_kernel<bf16_t><<<dim3(S, N), 256, 0, s>>>(tmp, S, v_lo, v_cnt, v_k, v_kmax, rescale, mean3[0], mean3[1], mean3[2],
                                                                std3[0], std3[1], std3[2], (bf16_t*)out, H);
    VCLA_CHECK_LAUNCH("resize_v_norm_kernel");
    return VCLA_OK;
}

extern "C" int vcla_image_preprocess(const uint8_t* img, int H, int W, uint8_t* tmp, int S, const int32_t* h_lo,
                                     const int32_t* h_cnt, const int32_t* h_k, int h_kmax, const int32_t* v_lo,
                                     const int32_t* v_cnt, const int32_t* v_k, int v_kmax, double rescale, const float* mean3,
                                     const float* std3, void* out, int dtype, void* stream) {
    return vcla_image_preprocess_batch(img, 1, H, W, tmp, S, h_lo, h_cnt, h_k, h_kmax, v_lo, v_cnt, v_k, v_kmax, rescale, mean3, std3, out,
                                       dtype, stream);
}
