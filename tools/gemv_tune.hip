// gemv_tune.hip -- tuning harness for the decode weight-streaming GEMV (M = 1, bf16): the same arithmetic as
// gemv_kernel in gemm.hip with the streaming knobs exposed (rows per wave, k-loop unroll, x staged in LDS vs re-read
// from L2, non-temporal weight loads, waves per workgroup), so one gpurun call can price every variant on the real
// LLaMA-7B shapes.  Exported as vcla_gemv_tune(); the winning configuration is what gemm.hip hard-codes.
#include "vcla_common.h"   // built with -I visual-chinese-llama-alpaca_amd/csrc into tools/libvcla_tune.so (make -C .../csrc tune)

// the harness is a stand-alone library: it carries its own copies of the two error helpers the product library defines in engine.hip
#include <stdarg.h>
static thread_local char t_err[512] = "";
void vcla_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(t_err, sizeof(t_err), fmt, ap); va_end(ap); }
int vcla_fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(t_err, sizeof(t_err), fmt, ap); va_end(ap); return code; }
extern "C" const char* vcla_tune_last_error(void) { return t_err; }

__device__ __forceinline__ uint4 tune_ld(const bf16_t* p, bool nt) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    if (nt) {
        const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        return make_uint4(t.x, t.y, t.z, t.w);
    }
    return *reinterpret_cast<const uint4*>(p);
}

// R rows per wave, U = k-steps (of 512 elements) issued together, XLDS = x (pre-multiplied by gamma) staged in LDS,
// WPB = waves per workgroup.  SWIGLU pairs rows (gate j, up j) like gemv_kernel.
template <int R, int U, bool XLDS, bool NT, int WPB, bool SWIGLU>
__global__ __launch_bounds__(WPB * 64) void gemv_tune_kernel(vcla_gemm_args a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [K] when XLDS
    __shared__ float red[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * WPB + wave;
    const bf16_t* X = (const bf16_t*)a.A;
    const bool fused_norm = a.norm_gamma != nullptr;
    float rstd = 1.f;
    if (XLDS) {
        float ss = 0.f;
        for (int k = threadIdx.x * 8; k < a.K; k += WPB * 64 * 8) {
            float xv[8];
            const uint4 t = *reinterpret_cast<const uint4*>(X + k);
            bf8_to_f32(t, xv);
            if (fused_norm) {
                const float4 g0 = *reinterpret_cast<const float4*>(a.norm_gamma + k);
                const float4 g1 = *reinterpret_cast<const float4*>(a.norm_gamma + k + 4);
                const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) { ss += xv[e] * xv[e]; xv[e] *= gm[e]; }
            }
            *reinterpret_cast<float4*>(xs + k) = make_float4(xv[0], xv[1], xv[2], xv[3]);
            *reinterpret_cast<float4*>(xs + k + 4) = make_float4(xv[4], xv[5], xv[6], xv[7]);
        }
        if (fused_norm) {
            ss = wave_sum(ss);
            if (lane == 0) red[wave] = ss;
        }
        __syncthreads();
        if (fused_norm) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < WPB; ++w) tot += red[w];
            rstd = rsqrtf(tot / (float)a.K + a.norm_eps);
        }
    }
    int rows[R];
    constexpr int P = SWIGLU ? R / 2 : R;
    const int n_out = SWIGLU ? a.N / 2 : a.N;
    if (gw * P >= n_out) return;
    if (SWIGLU) {
#pragma unroll
        for (int r = 0; r < P; ++r) {
            const int j = gw * P + r;
            rows[r] = (j >> 4) * 32 + (j & 15);
            rows[r + P] = rows[r] + 16;
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) rows[r] = gw * R + r;
    }
    const bf16_t* Wg = (const bf16_t*)a.W;
    const bf16_t* wp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wp[r] = Wg + (int64_t)rows[r] * a.K;
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    float ssq = 0.f;

    for (int k0 = lane * 8; k0 < a.K; k0 += 512 * U) {
        uint4 w[U][R];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * 512;
            if (k < a.K) {
#pragma unroll
                for (int r = 0; r < R; ++r) w[u][r] = tune_ld(wp[r] + k, NT);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * 512;
            if (k < a.K) {
                float xv[8];
                if (XLDS) {
                    const float4 x0 = *reinterpret_cast<const float4*>(xs + k), x1 = *reinterpret_cast<const float4*>(xs + k + 4);
                    xv[0] = x0.x; xv[1] = x0.y; xv[2] = x0.z; xv[3] = x0.w; xv[4] = x1.x; xv[5] = x1.y; xv[6] = x1.z; xv[7] = x1.w;
                } else {
                    const uint4 t = *reinterpret_cast<const uint4*>(X + k);
                    bf8_to_f32(t, xv);
                    if (fused_norm) {
                        const float4 g0 = *reinterpret_cast<const float4*>(a.norm_gamma + k);
                        const float4 g1 = *reinterpret_cast<const float4*>(a.norm_gamma + k + 4);
                        const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) { ssq += xv[e] * xv[e]; xv[e] *= gm[e]; }
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float wf[8];
                    bf8_to_f32(w[u][r], wf);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[r] += wf[e] * xv[e];
                }
            }
        }
    }
    if (!XLDS && fused_norm) rstd = rsqrtf(wave_sum(ssq) / (float)a.K + a.norm_eps);
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]) * rstd;
    bf16_t* Cg = (bf16_t*)a.C;
    if (SWIGLU) {
#pragma unroll
        for (int r = 0; r < P; ++r)
            if (lane == r) {
                float v = act_silu(acc[r]) * acc[r + P];
                const int n = gw * P + r;
                if (a.residual) v += bf2f(((const bf16_t*)a.residual)[n]);
                Cg[n] = f2bf(v);
            }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (lane == r && rows[r] < a.N) {
                float v = acc[r];
                if (a.bias) v += a.bias[rows[r]];
                if (a.residual) v += bf2f(((const bf16_t*)a.residual)[rows[r]]);
                Cg[rows[r]] = f2bf(v);
            }
    }
}

template <int R, int U, bool XLDS, bool NT, int WPB>
static int tune_launch(const vcla_gemm_args* a, hipStream_t s) {
    const bool sw = a->epilogue == VCLA_EPI_SWIGLU;
    const int per_wave = sw ? R / 2 : R;
    const int n_out = sw ? a->N / 2 : a->N;
    if (per_wave == 0) return vcla_fail(VCLA_ERR_BAD_ARG, "gemv_tune: SWIGLU needs R >= 2");
    const int waves = (n_out + per_wave - 1) / per_wave;
    const int blocks = (waves + WPB - 1) / WPB;
    const size_t lds = XLDS ? (size_t)a->K * 4 : 0;
    if (sw) {
        if constexpr (R >= 2) gemv_tune_kernel<R, U, XLDS, NT, WPB, true><<<blocks, WPB * 64, lds, s>>>(*a);
    } else {
        gemv_tune_kernel<R, U, XLDS, NT, WPB, false><<<blocks, WPB * 64, lds, s>>>(*a);
    }
    VCLA_CHECK_LAUNCH("gemv_tune_kernel");
    return VCLA_OK;
}

// variant = R (1,2,4,8) | U<<8 (1,2,4) | xlds<<16 | nt<<17 | (waves per block: 4 or 8)<<20
extern "C" int vcla_gemv_tune(const vcla_gemm_args* a, int variant, void* stream) {
    VCLA_REQUIRE(a && a->M == 1 && a->K % 64 == 0 && a->K <= 15360, VCLA_ERR_BAD_SHAPE, "gemv_tune: M must be 1, K %% 64 == 0, K <= 15360");
    const int R = variant & 0xff, U = (variant >> 8) & 0xff, xl = (variant >> 16) & 1, nt = (variant >> 17) & 1, wpb = (variant >> 20) & 0xf;
    hipStream_t s = (hipStream_t)stream;
#define TV(RR, UU, XX, NN, WW) \
    if (R == RR && U == UU && xl == XX && nt == NN && wpb == WW) return tune_launch<RR, UU, XX != 0, NN != 0, WW>(a, s);
#define TV_U(RR, XX, NN, WW) TV(RR, 1, XX, NN, WW) TV(RR, 2, XX, NN, WW) TV(RR, 4, XX, NN, WW)
#define TV_R(XX, NN, WW) TV_U(1, XX, NN, WW) TV_U(2, XX, NN, WW) TV_U(4, XX, NN, WW) TV_U(8, XX, NN, WW)
    TV_R(0, 1, 4) TV_R(1, 1, 4) TV_R(1, 0, 4) TV_R(1, 1, 8) TV_R(0, 1, 8)
#undef TV_R
#undef TV_U
#undef TV
    return vcla_fail(VCLA_ERR_BAD_ARG, "gemv_tune: variant %#x not built", variant);
}
