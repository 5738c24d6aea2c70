// gemm_epilogue.h -- epilogue shared by every MFMA GEMM kernel of libvisualcla_hip.so (gemm.hip, gemm_stream.hip).
#pragma once
#include "vcla_common.h"
#include <type_traits>

// ------------------------------------------------------------------ shared epilogue math
template <int EPI> __device__ __forceinline__ float epi_act(float x) {
    if (EPI == VCLA_EPI_QUICK_GELU) return act_quick_gelu(x);
    if (EPI == VCLA_EPI_GELU_ERF) return act_gelu_erf(x);
    return x;
}

__device__ __forceinline__ int64_t remap_row(const vcla_gemm_args& a, int m) {
    if (a.c_group_rows <= 0) return m;
    return (int64_t)(m / a.c_group_rows) * a.c_group_stride + (m % a.c_group_rows) + a.c_row_offset;
}

// ---- shared MFMA epilogue.  The wave owns MI x 4 accumulator tiles of 16x16 (operands swapped, see header):
// acc[i][j][r] = C[m][n] with m = mw + i*16 + (lane & 15), n = nw + j*16 + (lane >> 4)*4 + r  -> 4 consecutive
// columns per lane (8/16-byte stores); SWIGLU tiles (2j, 2j+1) = (gate, up) of output column nw/2 + j*16 + ...
// the lane's bias values (NJ tiles x 4 columns, 0 where there is none): what gemm_epilogue loads first.  A kernel whose K loop is long may
// fetch them BEFORE the loop and hand them over (`bia_pre`), so that the epilogue does not open with a dependent round trip to L2 / HBM.
template <int EPI, int NJ>
__device__ __forceinline__ void gemm_epilogue_bias(const vcla_gemm_args& a, int nw, int lane, float (&bia)[NJ][4]) {
    const int nq = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = nw + j * 16 + nq;            // packed column (SwiGLU: gate tiles 2jo, up tiles 2jo + 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) bia[j][r] = (a.bias && (EPI == VCLA_EPI_SWIGLU || n + r < a.N)) ? a.bias[n + r] : 0.f;
    }
}

// SCALES: the caller is the fp8 x fp8 kernel (w_scale / a_scale always present): its fast form is instantiated INSTEAD of the plain one, so
// that no kernel carries both (the persistent kernels have no registers to spare: with both, they spilled 20 values to scratch).
template <int EPI, typename OutT, int MI, int NJ = 4, bool SCALES = false>
__device__ __forceinline__ void gemm_epilogue(const vcla_gemm_args& a, f32x4_t (&acc)[MI][NJ], int mw, int nw, int lane, int m_end = 0x7fffffff,
                                              const float (*bia_pre)[4] = nullptr) {
    const int mrow = lane & 15, nq = (lane >> 4) * 4;
    OutT* Cg = (OutT*)a.C;
    constexpr bool kF32 = sizeof(OutT) == 4;
    const int n_out = (EPI == VCLA_EPI_SWIGLU) ? a.N / 2 : a.N;
    const bool vec_c = Cg && (a.ldc % 4 == 0) && vcla_aligned_dev(Cg, kF32 ? 16 : 8);
    const bool vec_r = a.residual && (a.ldr % 4 == 0) && vcla_aligned_dev(a.residual, 8);
    // values of output tile jo (0 .. NOUT-1) of row tile i for this lane: bias / fp8 scales / activation / residual applied;
    // returns the first of the lane's 4 output columns
    constexpr int NOUT = (EPI == VCLA_EPI_SWIGLU) ? NJ / 2 : NJ;
    // bias / fp8 weight scales of this lane's columns are the same for every row tile: fetched ONCE here.  (Inside the row loop
    // the compiler cannot hoist them past the C stores -- nothing tells it that C and bias do not alias -- and a 256 x 256 tile
    // paid 8 x 4 dependent loads per lane: the ViT GEMMs, which all carry a bias, ran 15-20 % slower in the model than the same
    // shapes without bias in the microbenchmark.)
    float bia[NJ][4], wsc[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = nw + j * 16 + nq;            // packed column (SwiGLU: gate tiles 2jo, up tiles 2jo + 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bia[j][r] = bia_pre ? bia_pre[j][r] : ((a.bias && (EPI == VCLA_EPI_SWIGLU || n + r < a.N)) ? a.bias[n + r] : 0.f);
            wsc[j][r] = a.w_scale ? a.w_scale[n + r] : 1.f;      // n + r < N_pad always
        }
    }
    auto tile_vals = [&](int i, int jo, int m, float ascale, float (&v)[4]) -> int {
        int n;
        if constexpr (EPI == VCLA_EPI_SWIGLU) {
            n = nw / 2 + jo * 16 + nq;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float gt = acc[i][2 * jo][r], up = acc[i][2 * jo + 1][r];
                if (a.w_scale) { gt *= wsc[2 * jo][r] * ascale; up *= wsc[2 * jo + 1][r] * ascale; }
                if (a.bias) { gt += bia[2 * jo][r]; up += bia[2 * jo + 1][r]; }
                v[r] = act_silu(gt) * up;
            }
        } else {
            n = nw + jo * 16 + nq;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[i][jo][r];
                if (a.w_scale) x *= wsc[jo][r] * ascale;
                if (a.bias && n + r < a.N) x += bia[jo][r];
                v[r] = epi_act<EPI>(x);
            }
        }
        if (a.residual && n < n_out) {
            const bf16_t* rp = (const bf16_t*)a.residual + (int64_t)m * a.ldr + n;
            if (vec_r && n + 3 < n_out) {
                float rv[4];
                Act<bf16_t>::ld4(rp, rv);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += rv[r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < n_out) v[r] += bf2f(rp[r]);
            }
        }
        return n;
    };
    // ---- wide stores (bf16 output, plain row-major C): the epilogue of a 256 x 256 tile is store-ISSUE bound (32 eight-byte
    // stores per lane).  Two adjacent output tiles are exchanged between the lane rows with v_permlane16_swap (rows 1 / 3 of the
    // first tile's registers <-> rows 0 / 2 of the second's): afterwards lane row r holds 8 CONSECUTIVE columns of tile
    // j + (r & 1), starting at column (r >> 1) * 8 -> one 16-byte store instead of two 8-byte ones.
    if constexpr (!kF32 && NOUT % 2 == 0) {
        const int n_first = (EPI == VCLA_EPI_SWIGLU) ? nw / 2 : nw;
        const bool wide = Cg && !a.C_frag && !a.c_row_ssq && (a.ldc % 8 == 0) && vcla_aligned_dev(Cg, 16) && (n_first % 8 == 0) &&
                          n_first + NOUT * 16 <= n_out;        // wave-uniform: the whole tile row is inside the matrix
        if (wide) {
            const int row_ = lane >> 4;
            // ---- fast form (residual rows 8-byte loadable): every wave-uniform question -- bias? residual? fp8 scales? -- is answered ONCE,
            // out here, and the element loops below are straight-line code.  The general form further down asks them per element (through
            // tile_vals); measured with per-workgroup stamps on the 256 x 256 kernel (profiles/r03_gemm256_ab.txt, run 33): of a tile's 6.4 us
            // epilogue (ViT qkv, bias only) 3.2 us were that arithmetic and only 1.7 us the stores.
#ifdef VCLA_EPI_NO_FAST      // A/B builds only (tools/debug): the general form for every tile
            if (false) {
#else
            if ((SCALES ? a.w_scale != nullptr : !a.w_scale && !a.a_scale) && (!a.residual || vec_r) && a.c_group_rows <= 0) {
#endif
                auto run = [&](auto hb_, auto hr_, auto hs_) {
                    constexpr bool kB = decltype(hb_)::value, kR = decltype(hr_)::value, kS = decltype(hs_)::value;   // bias, residual, fp8 scales
                    const bf16_t* Rg = (const bf16_t*)a.residual;
                    // the residual values of HB row tiles at a time, all requested before the first of their stores (the residual may alias C -- in
                    // place -- so a load issued after a store could not be hoisted above it): MI / HB round trips per tile instead of MI.  (All MI at
                    // once costs 64 registers: the persistent form of the 256 x 256 kernel then spilled 20 - 54 of them.)
                    constexpr int HB = MI >= 4 ? ((EPI == VCLA_EPI_GELU_ERF && SCALES) ? 1 : ((EPI == VCLA_EPI_GELU_ERF || SCALES) ? 2 : 4)) : MI;      // erf-GELU / the fp8 scales need registers: 2 row tiles per batch
#pragma unroll
                    for (int i0 = 0; i0 < MI; i0 += HB) {
                        uint2 rr[kR ? HB : 1][NOUT];
                        float asc[kS ? HB : 1];          // fp8 activations: the row's scale (1 when only the weights are fp8)
                        if constexpr (kR || kS) {
#pragma unroll
                            for (int ii = 0; ii < HB; ++ii) {
                                int m = mw + (i0 + ii) * 16 + mrow;
                                m = m < a.M ? m : a.M - 1;
                                if constexpr (kS) asc[ii] = a.a_scale ? a.a_scale[m] : 1.f;
                                if constexpr (kR) {
#pragma unroll
                                    for (int jo = 0; jo < NOUT; ++jo) rr[ii][jo] = *reinterpret_cast<const uint2*>(Rg + (int64_t)m * a.ldr + n_first + jo * 16 + nq);
                                }
                            }
                        }
#pragma unroll
                        for (int ii = 0; ii < HB; ++ii) {
                            const int i = i0 + ii;
                            const int m = mw + i * 16 + mrow;
                            if (m >= a.M || m >= m_end) continue;
#pragma unroll
                            for (int p = 0; p < NOUT / 2; ++p) {
                                float v[2][4];
#pragma unroll
                                for (int t = 0; t < 2; ++t) {
                                    const int jo = 2 * p + t;
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        if constexpr (EPI == VCLA_EPI_SWIGLU) {
                                            float gt = acc[i][2 * jo][r], up = acc[i][2 * jo + 1][r];
                                            if constexpr (kS) { gt *= wsc[2 * jo][r] * asc[ii]; up *= wsc[2 * jo + 1][r] * asc[ii]; }
                                            if constexpr (kB) { gt += bia[2 * jo][r]; up += bia[2 * jo + 1][r]; }
                                            v[t][r] = act_silu(gt) * up;
                                        } else {
                                            float x = acc[i][jo][r];
                                            if constexpr (kS) x *= wsc[jo][r] * asc[ii];
                                            if constexpr (kB) x += bia[jo][r];
                                            v[t][r] = epi_act<EPI>(x);
                                        }
                                    }
                                    if constexpr (kR) {
                                        v[t][0] += __uint_as_float(rr[ii][jo].x << 16); v[t][1] += __uint_as_float(rr[ii][jo].x & 0xffff0000u);
                                        v[t][2] += __uint_as_float(rr[ii][jo].y << 16); v[t][3] += __uint_as_float(rr[ii][jo].y & 0xffff0000u);
                                    }
                                }
                                const unsigned a0 = pack_bf2(v[0][0], v[0][1]), a1 = pack_bf2(v[0][2], v[0][3]);
                                const unsigned b0 = pack_bf2(v[1][0], v[1][1]), b1 = pack_bf2(v[1][2], v[1][3]);
                                const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                                const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                                const int n = n_first + (2 * p + (row_ & 1)) * 16 + (row_ >> 1) * 8;
                                *reinterpret_cast<uint4*>(Cg + (int64_t)m * a.ldc + n) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                            }
                        }
                    }
                };
                if constexpr (SCALES) {
                    if (a.bias) { if (a.residual) run(std::true_type{}, std::true_type{}, std::true_type{}); else run(std::true_type{}, std::false_type{}, std::true_type{}); }
                    else { if (a.residual) run(std::false_type{}, std::true_type{}, std::true_type{}); else run(std::false_type{}, std::false_type{}, std::true_type{}); }
                } else {
                    if (a.bias) { if (a.residual) run(std::true_type{}, std::true_type{}, std::false_type{}); else run(std::true_type{}, std::false_type{}, std::false_type{}); }
                    else { if (a.residual) run(std::false_type{}, std::true_type{}, std::false_type{}); else run(std::false_type{}, std::false_type{}, std::false_type{}); }
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = mw + i * 16 + mrow;
                if (m >= a.M || m >= m_end) continue;          // the 4 lanes that exchange data share m (m_end: rows of a tile strip that belong to the next tile)
                const int64_t crow = remap_row(a, m);
                const float ascale = a.a_scale ? a.a_scale[m] : 1.f;
#pragma unroll
                for (int p = 0; p < NOUT / 2; ++p) {
                    float v0[4], v1[4];
#if defined(VCLA_G2_EPI_ABLATE) && VCLA_G2_EPI_ABLATE == 2     // timing experiment: raw accumulators, no bias / activation / residual
                    for (int r = 0; r < 4; ++r) { v0[r] = acc[i][2 * p][r]; v1[r] = acc[i][2 * p + 1][r]; }
#else
                    tile_vals(i, 2 * p, m, ascale, v0);
                    tile_vals(i, 2 * p + 1, m, ascale, v1);
#endif
                    unsigned a0 = pack_bf2(v0[0], v0[1]), a1 = pack_bf2(v0[2], v0[3]);
                    unsigned b0 = pack_bf2(v1[0], v1[1]), b1 = pack_bf2(v1[2], v1[3]);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                    const int n = n_first + (2 * p + (row_ & 1)) * 16 + (row_ >> 1) * 8;
#if defined(VCLA_G2_EPI_ABLATE) && VCLA_G2_EPI_ABLATE == 1     // timing experiment: everything but the store itself
                    asm volatile("" :: "v"(s0[0]), "v"(s1[0]), "v"(s0[1]), "v"(s1[1]), "v"(n), "v"(crow));
#else
                    *reinterpret_cast<uint4*>(Cg + crow * a.ldc + n) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
#endif
                }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = mw + i * 16 + mrow;
        if (m >= a.M || m >= m_end) continue;
        const int64_t crow = remap_row(a, m);
        const float ascale = a.a_scale ? a.a_scale[m] : 1.f;   // fp8 activations: per-row scale (kernel 10)
#pragma unroll
        for (int j = 0; j < NOUT; ++j) {
            float v[4];
            const int n = tile_vals(i, j, m, ascale, v);   // first output column of this lane's 4
            if (n >= n_out) continue;
            if (a.c_row_ssq) {
                // deferred RMSNorm, producer side: sum of squares of the ROUNDED values of this row over this 16-column tile
                // (the 4 lanes l, l+16, l+32, l+48 hold the row's 16 columns); host guarantees N % 16 == 0 here
                float q = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float x = Act<OutT>::rnd(v[r]); q += x * x; }
                q += __shfl_xor(q, 16, 64);
                q += __shfl_xor(q, 32, 64);
                if (nq == 0) a.c_row_ssq[(int64_t)m * ((n_out + 15) >> 4) + (n >> 4)] = q;
            }
            if (a.C_frag) {
                // fragment-major copy for the next streaming GEMM (its K index = this output column n): the lane's 4
                // consecutive columns are half of one 8-element operand fragment -> one 8-byte store (n_out % 32 == 0)
                const int mt_c = (a.M + 15) >> 4;
                // 32-bit element offset (C_frag belongs to kernel 9: M <= 64 rows -> < 2^31 elements for any N): no 64-bit temporaries
                const unsigned fo = ((((unsigned)(n >> 5) * (unsigned)mt_c + (unsigned)(m >> 4)) * 64u + (unsigned)((n & 31) >> 3) * 16u + (unsigned)(m & 15)) << 3) + (unsigned)(n & 7);
                bf16_t* fp = (bf16_t*)a.C_frag + fo;
                float f[4] = {v[0], v[1], v[2], v[3]};
                if (a.c_frag_gamma) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) f[r] = a.c_frag_gamma[n + r] * Act<OutT>::rnd(v[r]);
                }
                *reinterpret_cast<uint2*>(fp) = make_uint2(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]));
            }
            if (!Cg) continue;
            OutT* cp = Cg + crow * a.ldc + n;
            if (vec_c && n + 3 < n_out) {
                Act<OutT>::st4(cp, v);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < n_out) Act<OutT>::st(cp + r, v[r]);
            }
        }
    }
}
