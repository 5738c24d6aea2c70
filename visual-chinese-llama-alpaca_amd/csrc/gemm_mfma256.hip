// gemm_mfma256.hip -- the 256 x 256 MFMA tile kernels of vcla_gemm (kernels 4 / 5: bf16, 64-deep K steps; kernel 10: fp8 x fp8, 128-deep),
// in their own translation unit: 48 + 8 instantiations, the longest compile of the library, now runs beside gemm.hip instead of inside it.
#include "vcla_common.h"
#include "gemm_epilogue.h"
#include "gemm_tiles.h"
#include <type_traits>
#include <stdlib.h>

// =================================================================== MFMA kernel, 256x256x64 tile, direct-to-LDS staging
// 8 waves (2 x 4), 128x64 outputs per wave (8 x 4 MFMA tiles, 128 fp32 accumulators per lane), one workgroup per CU
// (128 KiB of LDS: 2 buffers x (A 32 KiB + W 32 KiB)).  Tiles are staged with global_load_lds_dwordx4: each wave
// instruction moves 8 rows x 128 B straight into LDS (no VGPR round trip, no ds_write).  The LDS image of a wave
// instruction is lane-linear, so the bank-conflict swizzle is applied on the SOURCE side: lane (row, c') fetches global
// chunk c' ^ f(row), and the fragment reader applies the same XOR.  Tile k+1 streams in under the 64 MFMAs per wave of
// tile k; one barrier per K tile.
#define G2_BM 256
#define G2_BN 256
#define G2_TILE_BYTES (256 * GM_BK * 2)  // 32 KiB per operand tile

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// LDS-DMA from inline asm (16 B / 4 B per lane to LDS address `lds_dst` + lane * size): invisible to hipcc, which therefore neither
// counts it nor drains it at the next LDS access or barrier -- the PF form of the kernel below counts vmcnt by hand
__device__ __forceinline__ void g2_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void g2_dma4(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void g2_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#ifdef VCLA_G2_TIMELINE   // debug build only (make -C csrc timeline -> tools/libvcla_timeline.so): per-workgroup phase stamps, 100 MHz wall clock
__device__ unsigned long long* g2_timeline = nullptr;      // [workgroup][8]: entry, first slab landed, K loop done, epilogue issued, stores drained
extern "C" int vcla_debug_set_timeline(unsigned long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g2_timeline), &p, sizeof(p)); }
#define G2_STAMP(i_) do { if (g2_timeline && threadIdx.x == 0) g2_timeline[(size_t)blockIdx.x * 8 + (i_)] = wall_clock64(); } while (0)
// K-loop ablations are COMPILE-time (-DVCLA_G2_ABLATE=1/2/3; results are garbage): 1 = no fragment reads (MFMA + DMA only), 2 = no MFMAs
// (LDS reads + DMA only), 3 = no DMA after the first slab (MFMA + LDS reads only).  (A run-time switch wrecked the loop's code.)
#else
#define G2_STAMP(i_) do { } while (0)
#endif

// PF (VCLA_GEMM_PF=1, round 3): the L2-prefetch form.  The timeline (profiles/r03_gemm256_timeline.txt) shows a K step waiting
// ~0.6 us of its 1.8 for the next slab: all workgroups of an XCD walk K in step, so every slab is an L2 MISS for its first toucher
// (~1.9 us to MALL / HBM) and LDS has no room for a second slab in flight.  Here two waves per workgroup TOUCH the lines of the slab
// after next (one 4-byte LDS-DMA per line into a sink: 64 weight-row lines + 32 activation-row lines per workgroup -- the workgroups
// of an XCD that share a panel split its lines between them, blockIdx-derived, a pure speed assumption) two K steps before the slab's
// own DMA is issued, which then finds the lines in L2 or merges with the miss in flight.  All DMA is issued from inline asm and
// vmcnt is counted by hand (the newest instruction -- the touch -- may stay in flight across the barrier).  One tile per workgroup.
// XR = 1 (PF form only): tiles are 257 rows tall.  A ViT activation matrix has M = B * 257 rows (class token + 16 x 16 patches): with 256-row
// tiles every GEMM of the tower left a 64-row tail at B = 64 (a second, latency-bound launch: 10 us at K = 1024, 46 us at K = 4096 --
// 12 % of the vision stack); with 257-row tiles the launch is B x N/256 whole tiles and nothing else.  The 257th row rides along as a
// 17th 16-row MFMA strip of which only row 0 is real: its LDS piece (8 rows, 1 KiB) is one more DMA per K step for wave 0, its A
// fragment one more ds_read per wave and k-step, and its 16 output tiles are dealt two to each wave (+2 MFMAs on 32, the W fragments
// are already in registers).  Rows 1..15 of the strip are rows of the NEXT tile or stale LDS: computed, never stored (m_end).
template <int EPI, typename OutT, bool SGB, bool PF = false, int XR = 0>
__global__ __launch_bounds__(512) void gemm_mfma256_kernel(vcla_gemm_args a, int tiles_m, int tiles_n, int n_pad) {
    static_assert(XR == 0 || PF, "257-row tiles exist in the PF form only");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds2[];  // [buf][A|W][32 KiB] (+ 512 B sink, PF); XR: A = 34 KiB
    constexpr int G2_TM = G2_BM + XR;                                      // rows of the output tile
    constexpr int A_BYTES = XR ? 272 * 128 : G2_TILE_BYTES;                // A region of one stage (XR: 17 strips of 16 rows)
    constexpr int STAGE = A_BYTES + G2_TILE_BYTES;
    G2_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = PF ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int ntiles = tiles_m * tiles_n;

    // ---- staging: wave w owns pieces w*4 .. w*4+3 of each operand tile; piece = 8 rows x 128 B = one wave instruction
    const bf16_t* Ag = (const bf16_t*)a.A;
    const bf16_t* Wg = (const bf16_t*)a.W;
    const bf16_t* asrc[4];
    const bf16_t* wsrc[4];
    auto set_src = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave * 4 + i;
            const int row = piece * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);  // source-side swizzle (involution shared with lds_off)
            int am = m0 + row, wr = n0 + row;
            if (am >= a.M) am = a.M - 1;
            if (wr >= n_pad) wr = n_pad - 1;
            asrc[i] = Ag + (int64_t)am * a.lda + chunk * 8;
            wsrc[i] = Wg + (int64_t)wr * a.K + chunk * 8;
        }
    };
    auto issue = [&](int kt, int buf) {
        unsigned char* ab = lds2 + buf * 2 * G2_TILE_BYTES;
        unsigned char* wb = ab + G2_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave * 4 + i;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(asrc[i] + (int64_t)kt * GM_BK), (lds_ptr_t)(ab + piece * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wsrc[i] + (int64_t)kt * GM_BK), (lds_ptr_t)(wb + piece * 1024), 16, 0, 0);
        }
    };
    const int frow = lane & 15, fch = lane >> 4;
    const int nk = a.K / GM_BK;

    // Optionally persistent over output tiles (VCLA_GEMM_PERSIST=1: grid = 256, tile id = blockIdx.x, + gridDim.x, ...; the
    // XCD-aware order of tile_assign is kept because the grid is a multiple of 8): the FIRST K slab of the next tile is requested
    // during the last K step of the current one, so its latency and the drain of the epilogue's stores overlap instead of
    // opening every tile with a cold fetch.  Measured on MI355X: no difference (the per-round overhead is not the cold fetch),
    // so the default launch stays one workgroup per tile.
    int tile = blockIdx.x;
    int tm, tn;
    tile_assign(tile, tiles_m, tiles_n, 4, tm, tn);
    int m0 = tm * G2_TM, n0 = tn * G2_BN;
    set_src(m0, n0);
    if constexpr (PF) {
        const unsigned lds_u = (unsigned)(uintptr_t)(lds_ptr_t)lds2;
        // Operand addressing in BYTES: row-major (rows lda / K elements apart, K slabs 128 B apart) or SLAB-major (vcla_gemm_args.A_slab / W_slab,
        // [K/64][rows][64]: rows 128 B apart, slabs rows * 128 B apart).  The LDS image is the same either way; with slab-major operands the 8 rows
        // of a piece are ONE contiguous KiB, which the DMA path moves ~2.5x faster than the 8-row gather (profiles/r05_l2_intake.txt).
        const char* Ab = (const char*)(a.A_slab ? a.A_slab : a.A);
        const char* Wb = (const char*)(a.W_slab ? a.W_slab : a.W);
        const int64_t a_rs = a.A_slab ? 128 : a.lda * 2, a_ss = a.A_slab ? a.a_slab_rows * 128 : 128;
        const int64_t w_rs = a.W_slab ? 128 : (int64_t)a.K * 2, w_ss = a.W_slab ? (int64_t)n_pad * 128 : 128;
        const char* asb[4];
        const char* wsb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave * 4 + i;
            const int row = piece * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);  // source-side swizzle (involution shared with lds_off)
            int am = m0 + row, wr = n0 + row;
            if (am >= a.M) am = a.M - 1;
            if (wr >= n_pad) wr = n_pad - 1;
            asb[i] = Ab + (int64_t)am * a_rs + chunk * 16;
            wsb[i] = Wb + (int64_t)wr * w_rs + chunk * 16;
        }
        // XR: the piece that carries row 256 (rows 256 .. 263 of the tile; 257.. are the next tile's first rows, clamped into the matrix)
        const char* xsrc = nullptr;
        if constexpr (XR) {
            const int row = 256 + (lane >> 3);
            int am = m0 + row;
            if (am >= a.M) am = a.M - 1;
            xsrc = Ab + (int64_t)am * a_rs + ((lane & 7) ^ ((row >> 1) & 7)) * 16;
        }
        auto issue_pf = [&](int kt, int buf) {
            const unsigned ab = lds_u + buf * STAGE, wb = ab + A_BYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned piece = __builtin_amdgcn_readfirstlane((wave * 4 + i) * 1024);
                g2_dma16(asb[i] + (int64_t)kt * a_ss, ab + piece);
                g2_dma16(wsb[i] + (int64_t)kt * w_ss, wb + piece);
            }
            if constexpr (XR) {
                if (wave == 0) g2_dma16(xsrc + (int64_t)kt * a_ss, ab + 32 * 1024);
            }
        };
        // touch lines: wave 0 = 64 weight rows (quarter tm & 3 of the 256), wave 1 = 32 activation rows (eighth tn & 7); one line per
        // row and K step
        const bool pf_wave = wave < 2;
        int prow = wave == 0 ? n0 + 64 * (tm & 3) + lane : m0 + 32 * (tn & 7) + (lane & 31);
        if (wave == 0) prow = prow < n_pad ? prow : n_pad - 1; else prow = prow < a.M ? prow : a.M - 1;
        const char* pfsrc = wave == 0 ? Wb + (int64_t)prow * w_rs : Ab + (int64_t)prow * a_rs;
        const int64_t pf_ss = wave == 0 ? w_ss : a_ss;
        // (every workgroup touching ALL 512 lines of its own slab -- one touch per wave and K step -- is slower: B = 64 prefill 98.1 vs 92.5 ms,
        // profiles/r03_gemm256_ab.txt run 25: the touches are not free, sharing them across the workgroups of a panel is what pays)
        const unsigned sink = __builtin_amdgcn_readfirstlane(lds_u + 2 * STAGE + (wave & 1) * 256);   // (sink bytes are never read: waves may share them)
        auto touch = [&](int kt) {
            if (pf_wave) g2_dma4(pfsrc + (int64_t)(kt < nk ? kt : nk - 1) * pf_ss, sink);
        };
        f32x4_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        f32x4_t accx[1][2] = {{f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}}};   // XR: tiles (2 wm, 2 wm + 1) of the wave's columns, strip 16
        // the lane's 16 bias values are requested HERE, before the K loop (16 registers carried through it): fetched after the loop they cost the
        // epilogue a dependent round trip before its first store (~2 us of a 5 us epilogue)
        float bia_pre[4][4];
        gemm_epilogue_bias<EPI, 4>(a, n0 + wn * 64, lane, bia_pre);
        touch(1);
        issue_pf(0, 0);
        touch(2);
        // the K loop, with the wave's row half as a LITERAL in the 257-row form (the extra strip's W fragments must be literal register
        // indices, and a wave-uniform branch around its two MFMAs would split the block the sched_group_barrier pattern orders)
        auto k_loop = [&](auto wmc) {
            constexpr int WMC = decltype(wmc)::value;
            const int wmr = WMC < 0 ? wm : WMC;
            for (int kt = 0; kt < nk; ++kt) {
                const int cur = kt & 1;
                if (pf_wave) g2_vmcnt<1>(); else g2_vmcnt<0>();      // this wave's pieces of slab kt have landed (the newest touch may not have)
                __builtin_amdgcn_s_barrier();                        // ... everyone's; and buffer cur^1 is no longer read
                asm volatile("" ::: "memory");
                if (kt == 0) G2_STAMP(1);
                if (kt + 1 < nk) issue_pf(kt + 1, cur ^ 1);
                touch(kt + 3);
                const unsigned char* As = lds2 + cur * STAGE;
                const unsigned char* Ws = As + A_BYTES;
    #pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    bf16x8_t wf[4];
                    bf16x8_t af[8];
    #pragma unroll
                    for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(Ws + lds_off(wn * 64 + j * 16 + frow, kk * 4 + fch));
    #pragma unroll
                    for (int i = 0; i < 8; ++i)
                        af[i] = *reinterpret_cast<const bf16x8_t*>(As + lds_off(wmr * 128 + i * 16 + frow, kk * 4 + fch));
    #pragma unroll
                    for (int i = 0; i < 8; ++i)
    #pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
                    if constexpr (XR) {   // strip 16 (row 256 of the tile): tiles 2 WMC, 2 WMC + 1 of this wave's columns -- literal indices, one basic block
                        const bf16x8_t afx = *reinterpret_cast<const bf16x8_t*>(As + lds_off(256 + frow, kk * 4 + fch));
                        accx[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * (WMC < 0 ? 0 : WMC)], afx, accx[0][0], 0, 0, 0);
                        accx[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * (WMC < 0 ? 0 : WMC) + 1], afx, accx[0][1], 0, 0, 0);
                    }
                }
                if (SGB) {
                    // issue order, ONE pipeline over both 32-deep halves of the K step (round 4; was one pattern per half): 6 fragment reads (4 W + 2 A),
                    // then 4 MFMAs per further A read so that every ds_read runs two fragments ahead of the MFMAs that consume it (XR: a 13th read, 34
                    // MFMAs), and the first 6 reads of the SECOND half ride under the last MFMAs of the first instead of opening a second read ramp:
                    // +1 - 3 % on the LLaMA prefill shapes, nothing on the 257-row ViT tiles; 8 reads of lead are no better (profiles/r04_gemm256_sgb_ab.txt)
                    constexpr int LEAD = 6;
                    constexpr int NR = 12 + XR, NM = 32 + 2 * XR;          // reads / MFMAs per half
                    __builtin_amdgcn_sched_group_barrier(0x100, LEAD, 0);
    #pragma unroll
                    for (int i = 0; i < NR - LEAD; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    // first half: 4 (NR - LEAD) MFMAs issued; the second half's first LEAD reads ride under the rest, spread evenly
                    constexpr int REST = NM - 4 * (NR - LEAD), PER = REST / LEAD;
    #pragma unroll
                    for (int i = 0; i < LEAD; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    if constexpr (REST - PER * LEAD > 0) __builtin_amdgcn_sched_group_barrier(0x008, REST - PER * LEAD, 0);
    #pragma unroll
                    for (int i = 0; i < NR - LEAD; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, NM - 4 * (NR - LEAD), 0);
                }
                asm volatile("" ::: "memory");                       // the fragment reads stay on this side of the next barrier
            }
        };
        if constexpr (XR) {
            if (wm == 0) k_loop(std::integral_constant<int, 0>{}); else k_loop(std::integral_constant<int, 1>{});
        } else {
            k_loop(std::integral_constant<int, -1>{});
        }
        g2_vmcnt<0>();                                           // no DMA may land in LDS after the workgroup has given it up
        G2_STAMP(2);
        gemm_epilogue<EPI, OutT, 8>(a, acc, m0 + wm * 128, n0 + wn * 64, lane, 0x7fffffff, bia_pre);
        if constexpr (XR) {
            float bx[2][4];          // the strip's two tiles are tiles (2 wm, 2 wm + 1) of the wave: selects with literal indices, not a pointer into the registers
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) bx[t][r] = wm == 0 ? bia_pre[t][r] : bia_pre[2 + t][r];
            gemm_epilogue<EPI, OutT, 1, 2>(a, accx, m0 + 256, n0 + wn * 64 + wm * 32, lane, m0 + 257, bx);
        }
        G2_STAMP(3);
#ifdef VCLA_G2_TIMELINE
        __builtin_amdgcn_s_waitcnt(0);
        G2_STAMP(4);
#endif
        return;
    }
    int p0 = 0;                 // LDS buffer that holds K slab 0 of the current tile
    issue(0, p0);
    while (true) {
        f32x4_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        int nm0 = 0, nn0 = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = (kt + p0) & 1;
            __syncthreads();  // (compiler adds vmcnt(0)): slab kt has landed for every wave, and buffer cur^1 is no longer read
            if (kt == 0) G2_STAMP(1);
#if defined(VCLA_G2_ABLATE) && VCLA_G2_ABLATE == 3
            if (kt >= 1) { /* no more DMA */ } else
#endif
            if (kt + 1 < nk) {
                issue(kt + 1, cur ^ 1);
            } else if (has_next) {   // last K step: the staging pointers of this tile are dead -> re-aim them at the next tile
                tile_assign(next, tiles_m, tiles_n, 4, tm, tn);
                nm0 = tm * G2_BM; nn0 = tn * G2_BN;
                set_src(nm0, nn0);
                issue(0, cur ^ 1);
            }
            const unsigned char* As = lds2 + cur * 2 * G2_TILE_BYTES;
            const unsigned char* Ws = As + G2_TILE_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8_t wf[4];
                bf16x8_t af[8];
#if defined(VCLA_G2_ABLATE) && VCLA_G2_ABLATE == 1
#pragma unroll
                for (int j = 0; j < 4; ++j) { wf[j] = __builtin_bit_cast(bf16x8_t, acc[j][0]); asm volatile("" : "+v"(wf[j])); }
#pragma unroll
                for (int i = 0; i < 8; ++i) { af[i] = __builtin_bit_cast(bf16x8_t, acc[i][1]); asm volatile("" : "+v"(af[i])); }
#else
#pragma unroll
                for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(Ws + lds_off(wn * 64 + j * 16 + frow, kk * 4 + fch));
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    af[i] = *reinterpret_cast<const bf16x8_t*>(As + lds_off(wm * 128 + i * 16 + frow, kk * 4 + fch));
#endif
#if defined(VCLA_G2_ABLATE) && VCLA_G2_ABLATE == 2
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(wf[j]));      // keep the reads live (guide rule 17)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(af[i]));
#else
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
#endif
                if (SGB) {
                    // issue order: 6 fragment reads (4 W + 2 A), then 4 MFMAs per further A read, so every ds_read runs
                    // two fragments ahead of the MFMAs that consume it
                    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                }
            }
        }
        G2_STAMP(2);
        gemm_epilogue<EPI, OutT, 8>(a, acc, m0 + wm * 128, n0 + wn * 64, lane);
        G2_STAMP(3);
#ifdef VCLA_G2_TIMELINE
        __builtin_amdgcn_s_waitcnt(0);
        G2_STAMP(4);
#endif
        if (!has_next) break;
        p0 = (nk + p0) & 1;     // slab 0 of the next tile went into the buffer the last K step did not read
        tile = next; m0 = nm0; n0 = nn0;
    }
}

// =================================================================== fp8 x fp8 MFMA kernel, 256x256x128 tile (kernel 10)
// The same staging as gemm_mfma256_kernel -- a K tile of 128 fp8 values is 128 BYTES per row, exactly the row of a 64-wide bf16
// tile, so the direct-to-LDS pieces, the source-side swizzle and the 128 KiB double buffer are unchanged -- but each lane now
// carries 32 consecutive k (two 16-byte LDS reads) per operand tile and ONE v_mfma_scale_f32_16x16x128_f8f6f4 (unit block
// scales) replaces two bf16 MFMA k-steps at twice the K: half the MFMA issue slots and half the LDS / HBM bytes per flop.
// Per-row activation scales and per-row weight scales are applied to the fp32 accumulators in the epilogue.
typedef __attribute__((ext_vector_type(8))) int i32x8_t;

template <int EPI, typename OutT>
__global__ __launch_bounds__(512) void gemm_mfma256_fp8_kernel(vcla_gemm_args a, int tiles_m, int tiles_n, int n_pad) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds8[];  // [buf][A|W][32 KiB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int ntiles = tiles_m * tiles_n;
    const unsigned char* Ag = (const unsigned char*)a.A_q8;
    const unsigned char* Wg = (const unsigned char*)a.W_q8;
    const unsigned char* asrc[4];
    const unsigned char* wsrc[4];
    auto set_src = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave * 4 + i;
            const int row = piece * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);  // source-side swizzle (involution shared with lds_off)
            int am = m0 + row, wr = n0 + row;
            if (am >= a.M) am = a.M - 1;
            if (wr >= n_pad) wr = n_pad - 1;
            asrc[i] = Ag + (int64_t)am * a.K + chunk * 16;
            wsrc[i] = Wg + (int64_t)wr * a.K + chunk * 16;
        }
    };
    auto issue = [&](int kt, int buf) {
        unsigned char* ab = lds8 + buf * 2 * G2_TILE_BYTES;
        unsigned char* wb = ab + G2_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave * 4 + i;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(asrc[i] + (int64_t)kt * 128), (lds_ptr_t)(ab + piece * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wsrc[i] + (int64_t)kt * 128), (lds_ptr_t)(wb + piece * 1024), 16, 0, 0);
        }
    };
    const int frow = lane & 15, fch = (lane >> 4) * 2;   // this lane's 32 k = 16-byte chunks fch, fch + 1 of the 128-byte row
    const int nk = a.K / 128;
    auto frag = [&](const unsigned char* base, int row) {
        const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(base + lds_off(row, fch));
        const u32x4_t hi = *reinterpret_cast<const u32x4_t*>(base + lds_off(row, fch + 1));
        return i32x8_t{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    };
    // persistent over output tiles, next tile's first K slab requested during the last K step (see gemm_mfma256_kernel)
    int tile = blockIdx.x;
    int tm, tn;
    tile_assign(tile, tiles_m, tiles_n, 4, tm, tn);
    int m0 = tm * G2_BM, n0 = tn * G2_BN;
    set_src(m0, n0);
    int p0 = 0;
    issue(0, p0);
    while (true) {
        f32x4_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        int nm0 = 0, nn0 = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = (kt + p0) & 1;
            __syncthreads();  // (compiler adds vmcnt(0)): slab kt has landed for every wave, and buffer cur^1 is no longer read
            if (kt + 1 < nk) {
                issue(kt + 1, cur ^ 1);
            } else if (has_next) {
                tile_assign(next, tiles_m, tiles_n, 4, tm, tn);
                nm0 = tm * G2_BM; nn0 = tn * G2_BN;
                set_src(nm0, nn0);
                issue(0, cur ^ 1);
            }
            const unsigned char* As = lds8 + cur * 2 * G2_TILE_BYTES;
            const unsigned char* Ws = As + G2_TILE_BYTES;
            i32x8_t wf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = frag(Ws, wn * 64 + j * 16 + frow);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const i32x8_t af = frag(As, wm * 128 + i * 16 + frow);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], af, acc[i][j], 0 /* A = fp8 e4m3 */, 0 /* B = fp8 e4m3 */,
                                                                                0, 127 /* E8M0 1.0 */, 0, 127);
            }
        }
        gemm_epilogue<EPI, OutT, 8, 4, true>(a, acc, m0 + wm * 128, n0 + wn * 64, lane);
        if (!has_next) break;
        p0 = (nk + p0) & 1;
        tile = next; m0 = nm0; n0 = nn0;
    }
}

template <int EPI, typename OutT>
static int launch_mfma256_fp8(const vcla_gemm_args* a, hipStream_t s) {
    const int tiles_m = (a->M + G2_BM - 1) / G2_BM, tiles_n = (a->N + G2_BN - 1) / G2_BN;
    const int n_pad = (a->N + 127) / 128 * 128;
    const size_t lds = 4 * G2_TILE_BYTES;  // 128 KiB
    auto kern = gemm_mfma256_fp8_kernel<EPI, OutT>;
    static bool attr_set[VCLA_MAX_DEVICES] = {};   // per instantiation and device
    { const int rc_ = vcla_raise_dyn_lds((const void*)kern, lds, attr_set); if (rc_) return rc_; }
    static const int pg8 = getenv("VCLA_GEMM_PERSIST") ? atoi(getenv("VCLA_GEMM_PERSIST")) : 0;   // measured equal (see gemm_mfma256_kernel)
    const int nt8 = tiles_m * tiles_n;
    kern<<<(pg8 && nt8 > 256) ? 256 : nt8, 512, lds, s>>>(*a, tiles_m, tiles_n, n_pad);   // one workgroup per CU, persistent over tiles
    VCLA_CHECK_LAUNCH("gemm_mfma256_fp8_kernel");
    return VCLA_OK;
}

// 257-row tiles (gemm_mfma256_kernel<..., XR = 1>): M is a whole number of ViT sequences (class token + 16 x 16 patches) and the tile
// grid fills the chip about as well as the 256-row grid would
static bool vcla_gemm_tile257(const vcla_gemm_args* a) {
    return a->M >= 257 && a->M % 257 == 0 && a->c_group_rows <= 0;
}

template <int EPI, typename OutT, bool SGB>
static int launch_mfma256(const vcla_gemm_args* a, hipStream_t s) {
    const int tiles_m = (a->M + G2_BM - 1) / G2_BM, tiles_n = (a->N + G2_BN - 1) / G2_BN;
    const int n_pad = (a->N + 127) / 128 * 128;
    const size_t lds = 4 * G2_TILE_BYTES;  // 128 KiB
    auto kern = gemm_mfma256_kernel<EPI, OutT, SGB>;
    static bool attr_set[VCLA_MAX_DEVICES] = {};   // per instantiation and device
    { const int rc_ = vcla_raise_dyn_lds((const void*)kern, lds, attr_set); if (rc_) return rc_; }
    static const int pg = getenv("VCLA_GEMM_PERSIST") ? atoi(getenv("VCLA_GEMM_PERSIST")) : 0;   // measured equal: ViT fc1 180 vs 182 us, LLaMA gate/up 1288 vs 1265 us
    // PF (default since round 3: B = 64 prefill 98.4 -> 92.4 ms, vision stack 15.8 -> 15.3 ms in the model; VCLA_GEMM_PF=0 = the plain form)
    static const int pf = getenv("VCLA_GEMM_PF") ? atoi(getenv("VCLA_GEMM_PF")) : 1;
    static const int xr_env = getenv("VCLA_GEMM_XR") ? atoi(getenv("VCLA_GEMM_XR")) : 1;   // 0: 256-row tiles also when M % 257 == 0
    const int nt = tiles_m * tiles_n;
    if (!(pf && SGB) && (a->A_slab || a->W_slab)) return vcla_fail(VCLA_ERR_BAD_ARG, "gemm: slab-major operands need the direct-to-LDS form of the 256 x 256 kernel (force_kernel 4, VCLA_GEMM_PF=1)");
    if (pf && SGB && a->K >= 3 * GM_BK) {   // force_kernel 5 (SGB = false) stays the plain form: both forms remain under test
        if (xr_env && vcla_gemm_tile257(a)) {   // M = B * 257 (the ViT's token count): 257-row tiles, no ragged tail
            auto kx = gemm_mfma256_kernel<EPI, OutT, SGB, true, 1>;
            const size_t ldx = 2 * (272 * 128 + G2_TILE_BYTES) + 512;   // 132.5 KiB
            static bool attr_x[VCLA_MAX_DEVICES] = {};
            { const int rc_ = vcla_raise_dyn_lds((const void*)kx, ldx, attr_x); if (rc_) return rc_; }
            const int tmx = a->M / 257;
            kx<<<tmx * tiles_n, 512, ldx, s>>>(*a, tmx, tiles_n, n_pad);
            VCLA_CHECK_LAUNCH("gemm_mfma256_kernel<PF, 257>");
            return VCLA_OK;
        }
        auto kpf = gemm_mfma256_kernel<EPI, OutT, SGB, true>;
        static bool attr_pf[VCLA_MAX_DEVICES] = {};
        { const int rc_ = vcla_raise_dyn_lds((const void*)kpf, lds + 512, attr_pf); if (rc_) return rc_; }
        kpf<<<nt, 512, lds + 512, s>>>(*a, tiles_m, tiles_n, n_pad);
        VCLA_CHECK_LAUNCH("gemm_mfma256_kernel<PF>");
        return VCLA_OK;
    }
    kern<<<(pg && nt > 256) ? 256 : nt, 512, lds, s>>>(*a, tiles_m, tiles_n, n_pad);   // one workgroup per CU, persistent over tiles
    VCLA_CHECK_LAUNCH("gemm_mfma256_kernel");
    return VCLA_OK;
}

// ---- entry points for gemm.hip's dispatch (arguments validated there)
bool vcla_gemm_tile257_ok(const vcla_gemm_args* a) { return vcla_gemm_tile257(a); }

template <int EPI>
static int mfma256_epi(const vcla_gemm_args* a, bool sgb, hipStream_t s) {
    if (sgb) return a->out_f32 ? launch_mfma256<EPI, float, true>(a, s) : launch_mfma256<EPI, bf16_t, true>(a, s);
    return a->out_f32 ? launch_mfma256<EPI, float, false>(a, s) : launch_mfma256<EPI, bf16_t, false>(a, s);   // kernel 5: compiler-chosen interleave, plain form
}
int vcla_gemm_mfma256_launch(const vcla_gemm_args* a, bool sgb, hipStream_t s) {
    switch (a->epilogue) {
        case VCLA_EPI_NONE: return mfma256_epi<VCLA_EPI_NONE>(a, sgb, s);
        case VCLA_EPI_QUICK_GELU: return mfma256_epi<VCLA_EPI_QUICK_GELU>(a, sgb, s);
        case VCLA_EPI_GELU_ERF: return mfma256_epi<VCLA_EPI_GELU_ERF>(a, sgb, s);
        case VCLA_EPI_SWIGLU: return mfma256_epi<VCLA_EPI_SWIGLU>(a, sgb, s);
    }
    return vcla_fail(VCLA_ERR_BAD_ARG, "gemm: bad epilogue %d", a->epilogue);
}
template <int EPI>
static int mfma256_fp8_epi(const vcla_gemm_args* a, hipStream_t s) {
    return a->out_f32 ? launch_mfma256_fp8<EPI, float>(a, s) : launch_mfma256_fp8<EPI, bf16_t>(a, s);
}
int vcla_gemm_mfma256_fp8_launch(const vcla_gemm_args* a, hipStream_t s) {
    switch (a->epilogue) {
        case VCLA_EPI_NONE: return mfma256_fp8_epi<VCLA_EPI_NONE>(a, s);
        case VCLA_EPI_QUICK_GELU: return mfma256_fp8_epi<VCLA_EPI_QUICK_GELU>(a, s);
        case VCLA_EPI_GELU_ERF: return mfma256_fp8_epi<VCLA_EPI_GELU_ERF>(a, s);
        case VCLA_EPI_SWIGLU: return mfma256_fp8_epi<VCLA_EPI_SWIGLU>(a, s);
    }
    return vcla_fail(VCLA_ERR_BAD_ARG, "gemm: bad epilogue %d", a->epilogue);
}
