// gemv_decode.hip -- the M = 1 bf16 decode GEMV with a compile-time K (the LLaMA-7B / 13B widths): the B = 1 hot kernel.
//
// A decode layer at B = 1 is five launches of 33 - 180 MB of weights each; measured on MI355X (profiles/r01_bench_b1_kernel_stats.csv)
// every launch of the runtime-K kernel (gemv1_kernel, gemm.hip) costs  bytes / 6.9 TB/s + ~5.7 us: the fixed part (launch gap, x
// staging, the first HBM round trip, workgroup-count quantisation, the last round trip, the reduction) is a third of the
// layer.  This kernel attacks the fixed part, not the stream:
//   * persistent: as many workgroups as fit per CU at once (<= 3), each owns an equal contiguous share of the output rows and stages x * gamma in LDS
//     ONCE; a wave walks its rows with the loads of the NEXT row already in flight while it reduces the current one (a ring of
//     3 stages of 4 x 1 KiB per wave, never drained between rows).  No workgroup-count quantisation: the SwiGLU GEMV of 7B was
//     1376 workgroups = 5.375 per CU (a 6-vs-5 tail, 3 us of 31);
//   * x and gamma are requested first and the first weight stages right behind them, BEFORE x is staged: the LDS staging
//     (an L2 round trip + a barrier) overlaps the first HBM round trip instead of preceding it (vmcnt retires in order, so
//     the wait for x leaves the weight loads in flight);
//   * K is a template parameter: a row is straight-line code, every wait is a counted vmcnt, loads are buffer loads with
//     scalar row offsets (no 64-bit address VALU), the ragged last k-step (K = 11008 = 21.5 x 512) and the "no next row" case
//     are masked through the buffer bounds check (an out-of-range offset returns 0 and touches no memory) -- no branch
//     around any load, which would make the compiler drain the queue.
// Values: the summation structure of gemv1_kernel (fp32 x * gamma in LDS, lane-strided partial sums in k-step order, wave
// reduce, rstd applied to the reduced sum).
// Measured and dropped (round 2, profiles/r02_gemv_decode_ab.txt): warming L2 for the next launch -- the last workgroups
// of a GEMV touching the first 4 KiB per row of the next matrix (B = 1 decode 354 -> 338 tok/s: the producer pays more than the
// consumer saves), and 224 extra workgroups in the attention launch pulling o_proj through L2 (o_proj 7.5 -> 5.5 us, but the
// attention launch 7.2 -> 10.6 us: its dependent load chain runs at loaded instead of idle latency).
#include "gemm_epilogue.h"

typedef __attribute__((ext_vector_type(4))) unsigned int gv_u32x4;
#define GV_WPB 8
#define GV_OOB 0x80000000u

// FP8: W_q8 (OCP e4m3fn, one fp32 scale per row) instead of bf16 W -- a 16-byte load is 16 weights, a k-step 1024 elements
template <int R, int K, bool SWIGLU, bool FP8, typename OutT>
__global__ __launch_bounds__(GV_WPB * 64) void gemv1p_kernel(vcla_gemm_args a, int n_pad, int units) {
    constexpr int U = 4 / R;                 // a stage = 4 loads of 1 KiB per wave (U k-steps of R rows)
    constexpr int EB = FP8 ? 1 : 2, LE = 16 / EB, SE = 64 * LE;   // bytes per weight, weights per lane-load, elements per k-step
    constexpr int NSTEP = (K + SE - 1) / SE, NSTG = (NSTEP + U - 1) / U, DEPTH = NSTG < 3 ? NSTG : 3;
    constexpr int KP = NSTEP * SE;           // K padded to whole k-steps
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [KP] x * gamma (zero past K), then [GV_WPB] partial sums of squares
    float* red = xs + KP;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // unit = the R rows one wave reduces together (SwiGLU: the gate / up pair of one output); this workgroup owns units [lo, hi)
    const int lo = (int)((int64_t)blockIdx.x * units / gridDim.x), hi = (int)((int64_t)(blockIdx.x + 1) * units / gridDim.x);
    const int first = lo + wave;
    const int nu = first < hi ? (hi - first + GV_WPB - 1) / GV_WPB : 0;      // units of this wave: first, first + 8, ...
    auto unit_row = [&](int u, int r) -> int {                               // row r of unit u in the packed weight matrix
        if (SWIGLU) return (u >> 4) * 32 + (u & 15) + r * 16;                // 16-row gate block, 16-row up block, ...
        return u * R + r;                                                    // rows past N stay inside the 128-row padding of W
    };
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(FP8 ? a.W_q8 : a.W), 0, (int)((int64_t)n_pad * K * EB), 0x00020000);
    const unsigned voff = lane * 16;
    // lanes past K in the ragged last step: an offset beyond the buffer
    const unsigned voff_last = ((NSTEP - 1) * SE + lane * LE < K) ? voff : GV_OOB;

    // one stage of unit-row offsets `ro` (GV_OOB in ro[0] = no such unit: every load of the stage is masked)
    // (the mask is laundered through an empty asm: a select the compiler can see through becomes a BRANCH around the loads, and
    // a branch around a load makes every later wait a full drain)
    auto issue = [&](gv_u32x4 (&dst)[U][R], const unsigned (&ro)[R], int sg) {
        unsigned mask = ro[0] == GV_OOB ? GV_OOB : 0u;
        asm volatile("" : "+v"(mask));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int st = sg * U + u;
            if (st < NSTEP) {
                const unsigned vo = (st == NSTEP - 1 ? voff_last : voff) | mask;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    dst[u][r] = __builtin_bit_cast(gv_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, vo, (ro[r] & 0x7fffffffu) + st * 1024, 2 /* nt */));
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the ring: without it the scheduler hoists every load of the row to the top
    };
    auto row_offsets = [&](int i, unsigned (&ro)[R]) {                       // unit i of this wave (GV_OOB when past the end)
#pragma unroll
        for (int r = 0; r < R; ++r) ro[r] = i < nu ? (unsigned)unit_row(first + i * GV_WPB, r) * (unsigned)(K * EB) : GV_OOB;
    };

    // ---- x (and gamma) are requested first, the first DEPTH weight stages right behind them
    const bf16_t* X = (const bf16_t*)a.A;
    const bool fused_norm = a.norm_gamma != nullptr;
    constexpr int NXI = (KP + 4095) / 4096;     // staging passes of the 512 threads (8 elements each)
    gv_u32x4 xraw[NXI];
    float4 g0[NXI], g1[NXI];
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
        const int k = i * 4096 + (int)threadIdx.x * 8;
        const int kc = k < K ? k : 0;                    // clamped: the value is replaced by zeros below
        xraw[i] = *reinterpret_cast<const gv_u32x4*>(X + kc);
        if (fused_norm) {
            g0[i] = *reinterpret_cast<const float4*>(a.norm_gamma + kc);
            g1[i] = *reinterpret_cast<const float4*>(a.norm_gamma + kc + 4);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    gv_u32x4 w[NSTG][U][R];                              // stage registers of the CURRENT row (at most DEPTH stages live at a time)
    unsigned ro_cur[R];
    row_offsets(0, ro_cur);
#pragma unroll
    for (int sg = 0; sg < DEPTH; ++sg) issue(w[sg], ro_cur, sg);

    // ---- stage x * gamma as fp32 in LDS (+ sum of squares for the fused RMSNorm)
    float rstd = 1.f;
    {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int k = i * 4096 + (int)threadIdx.x * 8;
            if (k < KP) {
                float xv[8];
                bf8_to_f32(__builtin_bit_cast(uint4, xraw[i]), xv);
                if (fused_norm && k < K) {      // (the clamped loads past K must not reach the sum of squares)
                    const float gm[8] = {g0[i].x, g0[i].y, g0[i].z, g0[i].w, g1[i].x, g1[i].y, g1[i].z, g1[i].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ss += xv[e] * xv[e]; xv[e] *= gm[e]; }
                }
                if (k >= K) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) xv[e] = 0.f;
                }
                *reinterpret_cast<float4*>(xs + k) = make_float4(xv[0], xv[1], xv[2], xv[3]);
                *reinterpret_cast<float4*>(xs + k + 4) = make_float4(xv[4], xv[5], xv[6], xv[7]);
            }
        }
        if (fused_norm) {
            ss = wave_sum(ss);
            if (lane == 0) red[wave] = ss;
        }
        __syncthreads();
        if (fused_norm) {
            float tot = 0.f;
#pragma unroll
            for (int wv = 0; wv < GV_WPB; ++wv) tot += red[wv];
            rstd = rsqrtf(tot / (float)K + a.norm_eps);
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    OutT* Cg = (OutT*)a.C + remap_row(a, 0) * a.ldc;
    // residual / bias of a row are buffer loads too, requested at the START of the row (a plain load at its end would wait
    // vmcnt(0) = drain the ring every row); a NULL operand is a zero-length buffer -> the loads return 0 and are never branched on
    const int n_out = SWIGLU ? a.N / 2 : a.N;
    const __amdgpu_buffer_rsrc_t rRes = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.residual), 0, a.residual ? n_out * 2 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rBias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, a.bias ? n_pad * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rScale = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w_scale), 0, FP8 ? n_pad * 4 : 0, 0x00020000);
    for (int i = 0; i < nu; ++i) {
        const int unit = first + i * GV_WPB;
        // lane r < R finishes output r of the unit (SwiGLU: lane 0 the output, lanes 0 / 1 fetch the gate / up bias)
        unsigned evo = SWIGLU ? (lane == 0 ? (unsigned)unit * 2u : GV_OOB) : (lane < R ? (unsigned)(unit * R + lane) * 2u : GV_OOB);
        unsigned bvo = lane < R ? (unsigned)(SWIGLU ? unit_row(unit, lane) : unit * R + lane) * 4u : GV_OOB;
        asm volatile("" : "+v"(evo), "+v"(bvo));
        const float res_v = bf2f((bf16_t)__builtin_amdgcn_raw_buffer_load_b16(rRes, evo, 0, 0));
        const float bias_v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rBias, bvo, 0, 0));
        float wsc_v = 1.f;                               // fp8: dequantisation scale of this lane's row
        if constexpr (FP8) wsc_v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rScale, bvo, 0, 0));
        __builtin_amdgcn_sched_barrier(0);
        unsigned ro_next[R];
        row_offsets(i + 1, ro_next);
        gv_u32x4 wn[DEPTH][U][R];                        // the first stages of the next row, requested while this one is reduced
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
#pragma unroll
        for (int sg = 0; sg < NSTG; ++sg) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int st = sg * U + u;
                if (st < NSTEP) {
                    const int k = st * SE + lane * LE;
                    float xv[LE];
#pragma unroll
                    for (int q = 0; q < LE / 4; ++q) {
                        const float4 t = *reinterpret_cast<const float4*>(xs + k + q * 4);
                        xv[q * 4] = t.x; xv[q * 4 + 1] = t.y; xv[q * 4 + 2] = t.z; xv[q * 4 + 3] = t.w;
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        if constexpr (FP8) {
                            typedef __attribute__((ext_vector_type(2))) float f32x2_t;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const unsigned d = w[sg][u][r][q];
                                const f32x2_t lo2 = __builtin_amdgcn_cvt_pk_f32_fp8(d, false), hi2 = __builtin_amdgcn_cvt_pk_f32_fp8(d, true);
                                acc[r] += lo2.x * xv[q * 4] + lo2.y * xv[q * 4 + 1] + hi2.x * xv[q * 4 + 2] + hi2.y * xv[q * 4 + 3];
                            }
                        } else {
                            float wf[8];
                            bf8_to_f32(__builtin_bit_cast(uint4, w[sg][u][r]), wf);
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[r] += wf[e] * xv[e];
                        }
                    }
                }
            }
            // pin the stage: the FMA chain has no memory dependence, and the instruction selector otherwise sinks ALL of it below
            // the last load issue (every x value and weight register of the row live at once -> spills); the empty asm ties the
            // accumulators to program order
#pragma unroll
            for (int r = 0; r < R; ++r) asm volatile("" : "+v"(acc[r]));
            __builtin_amdgcn_sched_barrier(0);
            if (sg + DEPTH < NSTG) issue(w[sg + DEPTH < NSTG ? sg + DEPTH : 0], ro_cur, sg + DEPTH);
            else issue(wn[sg + DEPTH >= NSTG ? sg + DEPTH - NSTG : 0], ro_next, sg + DEPTH - NSTG);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]) * rstd;
        if (SWIGLU) {
            const float gt = acc[0] * wsc_v + bias_v, up = acc[1] * __shfl(wsc_v, 1, 64) + __shfl(bias_v, 1, 64);
            const float v = act_silu(gt) * up + res_v;
            if (lane == 0) Act<OutT>::st(Cg + unit, v);
        } else {
            float v = acc[0];
#pragma unroll
            for (int r = 1; r < R; ++r) v = lane == r ? acc[r] : v;
            v = v * wsc_v + bias_v;
            v += res_v;
            const int n = unit * R + lane;
            if (lane < R && n < a.N) Act<OutT>::st(Cg + n, v);
        }
#pragma unroll
        for (int sg = 0; sg < DEPTH; ++sg)
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < R; ++r) w[sg][u][r] = wn[sg][u][r];
#pragma unroll
        for (int r = 0; r < R; ++r) ro_cur[r] = ro_next[r];
    }
}

template <int R, int K, bool SWIGLU, bool FP8, typename OutT>
static int launch_gemv1p(const vcla_gemm_args* a, hipStream_t s) {
    constexpr int P = SWIGLU ? 1 : R, SE = FP8 ? 1024 : 512, KP = (K + SE - 1) / SE * SE;
    const int n_out = SWIGLU ? a->N / 2 : a->N;
    const int units = (n_out + P - 1) / P;
    const int n_pad = (a->N + 127) / 128 * 128;
    // persistent grid: every CU gets as many workgroups as fit at once (registers / LDS of THIS instance), at most 3
    // (measured on MI355X, tools/bench_kernels.py gemv1: 7B gate/up 30.7 / 30.3 / 29.3 us at 1 / 2 / 3 per CU)
    static const int occ_env = getenv("VCLA_GEMV_OCC") ? atoi(getenv("VCLA_GEMV_OCC")) : 3;
    static const int per_cu = [] {
        int occ = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 512;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gemv1p_kernel<R, K, SWIGLU, FP8, OutT>, GV_WPB * 64, (size_t)(KP + GV_WPB) * 4) != hipSuccess || occ < 1)
            occ = 1;
        return prop.multiProcessorCount * (occ < occ_env ? occ : occ_env);
    }();
    int grid = (units + GV_WPB - 1) / GV_WPB;
    if (grid > per_cu) grid = per_cu;
    gemv1p_kernel<R, K, SWIGLU, FP8, OutT><<<grid, GV_WPB * 64, (size_t)(KP + GV_WPB) * 4, s>>>(*a, n_pad, units);
    VCLA_CHECK_LAUNCH("gemv1p_kernel");
    return VCLA_OK;
}

template <int K>
static int launch_gemv1p_k(const vcla_gemm_args* a, hipStream_t s) {
    if (a->W_q8 && a->w_scale) {   // fp8 rows are half as long: always 2 rows per wave, so that a stage is still 4 KiB
        if (a->epilogue == VCLA_EPI_SWIGLU)
            return a->out_f32 ? launch_gemv1p<2, K, true, true, float>(a, s) : launch_gemv1p<2, K, true, true, bf16_t>(a, s);
        return a->out_f32 ? launch_gemv1p<2, K, false, true, float>(a, s) : launch_gemv1p<2, K, false, true, bf16_t>(a, s);
    }
    if (a->epilogue == VCLA_EPI_SWIGLU)
        return a->out_f32 ? launch_gemv1p<2, K, true, false, float>(a, s) : launch_gemv1p<2, K, true, false, bf16_t>(a, s);
    if (a->N >= 16384)  // very tall (lm_head): 2 rows per wave
        return a->out_f32 ? launch_gemv1p<2, K, false, false, float>(a, s) : launch_gemv1p<2, K, false, false, bf16_t>(a, s);
    return a->out_f32 ? launch_gemv1p<1, K, false, false, float>(a, s) : launch_gemv1p<1, K, false, false, bf16_t>(a, s);
}

// returns VCLA_OK (or a launch error) when it handled the call, -1 when this K has no compiled instance or the matrix is too
// large for one buffer descriptor (the caller falls back to gemv1_kernel)
int vcla_gemv1x_launch(const vcla_gemm_args* a, hipStream_t s) {
    static const int on = getenv("VCLA_GEMV1X") ? atoi(getenv("VCLA_GEMV1X")) : 1;
    if (!on || a->M != 1 || !(a->W || (a->W_q8 && a->w_scale))) return -1;
    if ((int64_t)((a->N + 127) / 128 * 128) * a->K * 2 >= (int64_t)1 << 31) return -1;
    switch (a->K) {
        case 4096: return launch_gemv1p_k<4096>(a, s);
        case 11008: return launch_gemv1p_k<11008>(a, s);
        case 5120: return launch_gemv1p_k<5120>(a, s);
        case 13824: return launch_gemv1p_k<13824>(a, s);
        default: return -1;
    }
}
