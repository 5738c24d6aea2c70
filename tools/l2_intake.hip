// l2_intake.hip -- how fast does ONE CU of an MI355X ingest an L2-resident panel, alone and beside an HBM weight stream?
//
// Background: the batch-decode GEMMs (M = 64 / 256 rows) are modelled as
//     t ~ W_share / r_hbm + A_panel / r_l2 + launch,      r_hbm ~ 11 B/clk/CU,  r_l2 ~ 24 B/clk/CU (a FIT, never measured),
// while MI355X_MICROARCH.md gives ~34.5 TB/s aggregate L2 = ~56 B/clk/CU.  This probe measures r_l2 directly:
//   * every workgroup (one per CU, 256 of them) reads the SAME panel (512 KB = the o_proj activation panel at M = 64, 1.4 MB =
//     down_proj's, 2 MB = M = 256) `passes` times, with 1 / 2 / 4 / 8 reader waves x 2 / 4 / 8 sixteen-byte loads in flight per lane,
//     either into registers (global_load_dwordx4) or straight into LDS (global_load_lds_dwordx4, hand-counted vmcnt);
//   * the same beside a weight stream: 4 more waves of the workgroup stream a private slice of a 1.5 GB buffer with
//     non-temporal loads (what gemm_dstream_kernel / gemv1p_kernel do), the readers run until the stream ends.
// Output: bytes / clk / CU at 2.4 GHz and GB/s per CU for each role.  Standalone: hipcc --offload-arch=gfx950 -O3 -o tools/l2_intake tools/l2_intake.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s (%s:%d)\n", #e, hipGetErrorString(_e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ u32x4_t ld_nt(const u32x4_t* p) { return __builtin_nontemporal_load(p); }

struct Args {
    const u32x4_t* panel;       // shared, L2-resident after the first pass
    long long panel_kb;         // KiB, multiple of NA * L
    int passes;                 // reader passes when there is no stream (else: until the stream ends)
    const u32x4_t* wbuf;        // weight stream, private slice per workgroup
    long long w_kb_per_wg;      // KiB per workgroup (0: no stream)
    unsigned long long* a_kb;   // [grid] KiB the readers of a workgroup fetched
    unsigned int* sink;
    long long ld;               // PAT = 1: the panel is a [rows][ld bytes] matrix read as 8-row x 128-byte pieces (the tile kernels' DMA shape)
};

// NA reader waves (register loads, L in flight per lane, two batches alternating so that L stay in flight while L are consumed),
// NW streamer waves (nt loads, 3 stages x 4 KiB per wave in flight: the gemv1p / dstream ring).
template <int NA, int L, int NW, bool DMA, int PAT = 0>
__global__ __launch_bounds__((NA + NW) * 64) void intake_kernel(Args a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    __shared__ int stream_left;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) stream_left = NW;
    __syncthreads();
    u32x4_t acc = {0u, 0u, 0u, 0u};
    if (wave < NA) {
        // chunk c of the panel = L KiB; wave w takes chunks w, w + NA, ...
        const long long chunks = a.panel_kb / L;
        const char* base = (const char*)a.panel + (PAT ? 0 : lane * 16);
        // PAT = 1: piece p = (row block p % rbl, K slab p / rbl): lane -> row (lane >> 3) of the block, 16-byte chunk (lane & 7) of the slab's 128 bytes
        // PAT = 2: contiguous 1 KiB, lanes XOR-permuted inside each 128-byte row (the source-side bank swizzle of a slab-major operand);
        // PAT = 3 / 4: 2 rows x 512 B / 4 rows x 256 B per instruction (K slabs of 256 / 128 of a row-major matrix)
        auto off = [&](long long p) -> long long {
            if (PAT == 0) return p * 1024;
            if (PAT == 2) return p * 1024 + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 4) & 7)) * 16) - lane * 0;
            const int rows_per = PAT == 1 ? 8 : (PAT == 3 ? 2 : 4), seg = 1024 / rows_per, lpr = 64 / rows_per;   // bytes and lanes per row segment
            const long long rbl = (a.panel_kb * 1024 / a.ld) / rows_per;
            const long long rb = p % rbl, slab = p / rbl;
            return (rb * rows_per + lane / lpr) * a.ld + slab * seg + (lane % lpr) * 16;
        };
        unsigned long long got = 0;
        long long c = wave;
        int pass = 0;
        auto next = [&]() { c += NA; if (c >= chunks) { c -= chunks; ++pass; } };
        // beside a stream: until the stream of this workgroup ends -- but never more than 4096 passes (a starved stream must not turn the probe into a hang:
        // the first run of this tool sat in the 8-wave x 8-load configuration until its time limit)
        auto more = [&]() { return NW > 0 ? (pass < 4096 && __hip_atomic_load(&stream_left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > 0) : (pass < a.passes); };
        if constexpr (!DMA) {
            u32x4_t b0[L], b1[L];
#pragma unroll
            for (int j = 0; j < L; ++j) b0[j] = *(const u32x4_t*)(base + off(c * L + j));
            next();
            while (more()) {
#pragma unroll
                for (int j = 0; j < L; ++j) b1[j] = *(const u32x4_t*)(base + off(c * L + j));
                next();
#pragma unroll
                for (int j = 0; j < L; ++j) acc ^= b0[j];
#pragma unroll
                for (int j = 0; j < L; ++j) b0[j] = *(const u32x4_t*)(base + off(c * L + j));
                next();
#pragma unroll
                for (int j = 0; j < L; ++j) acc ^= b1[j];
                got += 2 * L;
            }
#pragma unroll
            for (int j = 0; j < L; ++j) acc ^= b0[j];
        } else {
            // LDS-DMA: two slots of L KiB per wave; issue slot s^1, wait for slot s (vmcnt(L)), read it back with ds_read_b128
            const unsigned lds_u = (unsigned)(uintptr_t)(lds_ptr_t)lds + wave * (2 * L * 1024);
            const unsigned char* mine = lds + wave * (2 * L * 1024);
#pragma unroll
            for (int j = 0; j < L; ++j) dma16(base + off(c * L + j), lds_u + j * 1024);
            next();
            while (more()) {
#pragma unroll
                for (int j = 0; j < L; ++j) dma16(base + off(c * L + j), lds_u + (L + j) * 1024);
                next();
                vmcnt<L>();
#pragma unroll
                for (int j = 0; j < L; ++j) acc ^= *(const u32x4_t*)(mine + j * 1024 + lane * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < L; ++j) dma16(base + off(c * L + j), lds_u + j * 1024);
                next();
                vmcnt<L>();
#pragma unroll
                for (int j = 0; j < L; ++j) acc ^= *(const u32x4_t*)(mine + (L + j) * 1024 + lane * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                got += 2 * L;
            }
            vmcnt<0>();
        }
        if (lane == 0) atomicAdd(&a.a_kb[blockIdx.x], got);
    } else {
        // weight stream: wave v of NW takes 4 KiB pieces v, v + NW, ... of the workgroup's slice; 3 pieces in flight
        const int v = wave - NA;
        const char* base = (const char*)a.wbuf + (long long)blockIdx.x * a.w_kb_per_wg * 1024 + lane * 16;
        const long long pieces = a.w_kb_per_wg / 4;
        u32x4_t r0[4], r1[4], r2[4];
        long long p = v;
        auto ldp = [&](u32x4_t (&r)[4]) {
            const long long q = p < pieces ? p : pieces - 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = ld_nt((const u32x4_t*)(base + (q * 4 + j) * 1024));
            p += NW;
        };
        auto use = [&](u32x4_t (&r)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc ^= r[j];
        };
        ldp(r0); ldp(r1); ldp(r2);
        for (long long it = v; it < pieces; it += 3 * NW) {
            use(r0); ldp(r0);
            use(r1); ldp(r1);
            use(r2); ldp(r2);
        }
        use(r0); use(r1); use(r2);
        if (lane == 0) __hip_atomic_fetch_add(&stream_left, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) a.sink[0] = 1;   // never true on random data; keeps the loads alive
}

static double g_clk_ghz = 2.4;

template <int NA, int L, int NW, bool DMA, int PAT = 0>
static void run(const char* tag, Args a, int grid, FILE* out) {
    const size_t lds_bytes = DMA ? (size_t)NA * 2 * L * 1024 : 0;
    if (lds_bytes > 160 * 1024) return;
    auto kern = intake_kernel<NA, L, NW, DMA, PAT>;
    if (lds_bytes > 48 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    if constexpr (NA > 0) a.panel_kb = a.panel_kb / (NA * L) * (NA * L);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e30;
    unsigned long long a_kb_tot = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(a.a_kb, 0, (size_t)grid * 8));
        CK(hipEventRecord(e0));
        kern<<<grid, (NA + NW) * 64, lds_bytes>>>(a);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 0) continue;           // first launch: panel not yet in every XCD's L2, code object load
        if (ms < best) {
            best = ms;
            std::vector<unsigned long long> h(grid);
            CK(hipMemcpy(h.data(), a.a_kb, (size_t)grid * 8, hipMemcpyDeviceToHost));
            a_kb_tot = 0;
            for (auto x : h) a_kb_tot += x;
        }
    }
    const double sec = best * 1e-3;
    const double a_bytes_cu = (double)a_kb_tot * 1024 / grid, w_bytes_cu = (double)a.w_kb_per_wg * 1024;
    const double a_gbs = a_bytes_cu / sec / 1e9, w_gbs = w_bytes_cu / sec / 1e9;
    fprintf(out, "%-34s NA=%d L=%d NW=%d %s panel=%5lld KB  t=%8.1f us  A: %6.1f GB/s/CU = %5.1f B/clk/CU (chip %5.2f TB/s)", tag, NA, L, NW,
            DMA ? (PAT == 1 ? "lds-dma/8x128" : PAT == 2 ? "lds-dma/1K-swz" : PAT == 3 ? "lds-dma/2x512" : PAT == 4 ? "lds-dma/4x256" : "lds-dma") : (PAT ? "vgpr/rows" : "vgpr   "), a.panel_kb, sec * 1e6, a_gbs, a_gbs / g_clk_ghz, a_gbs * grid / 1e3);
    if (NW) fprintf(out, "   W: %6.1f GB/s/CU = %5.1f B/clk/CU (chip %5.2f TB/s)", w_gbs, w_gbs / g_clk_ghz, w_gbs * grid / 1e3);
    fprintf(out, "\n");
    fflush(out);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    FILE* out = stdout;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    fprintf(out, "# l2_intake: %s, %d CUs, clockRate %.0f MHz (B/clk figures use 2.4 GHz)\n", prop.gcnArchName, cus, prop.clockRate / 1e3);
    const size_t panel_max = 8u << 20, wbytes = (size_t)1536 << 20;
    u32x4_t *panel, *wbuf;
    unsigned long long* a_kb;
    unsigned int* sink;
    CK(hipMalloc(&panel, panel_max)); CK(hipMalloc(&wbuf, wbytes)); CK(hipMalloc(&a_kb, 4096 * 8)); CK(hipMalloc(&sink, 4));
    {   // random-ish fill (DVFS: zero-filled inputs clock higher)
        std::vector<unsigned> h(panel_max / 4);
        unsigned s = 12345u;
        for (auto& x : h) { s = s * 1664525u + 1013904223u; x = s; }
        CK(hipMemcpy(panel, h.data(), panel_max, hipMemcpyHostToDevice));
        for (size_t off = 0; off < wbytes; off += panel_max) CK(hipMemcpy((char*)wbuf + off, h.data(), panel_max, hipMemcpyHostToDevice));
    }
    const int grid = cus;
    Args a{};
    a.panel = panel; a.wbuf = wbuf; a.a_kb = a_kb; a.sink = sink;
    const bool full = argc > 1 && argv[1][0] == 'f';
    if (full) {
    const long long panels_kb[] = {512, 1408, 2048, 5632};
    fprintf(out, "\n## 1. panel alone (every CU reads the same L2-resident panel, %d workgroups)\n", grid);
    for (long long pk : panels_kb) {
        a.panel_kb = pk; a.w_kb_per_wg = 0;
        a.passes = (int)(16384 / pk) + 2;
#define ALONE(NA, L) run<NA, L, 0, false>("alone", a, grid, out); run<NA, L, 0, true>("alone", a, grid, out);
        ALONE(1, 4) ALONE(1, 8) ALONE(2, 4) ALONE(2, 8) ALONE(4, 2) ALONE(4, 4) ALONE(4, 8) ALONE(8, 2) ALONE(8, 4) ALONE(8, 8)
        run<16, 4, 0, false>("alone", a, grid, out);
    }
    fprintf(out, "\n## 2. weight stream alone (nt loads, 4 or 8 waves x 12 KiB in flight, private 1.4 - 5.6 MB slice per CU)\n");
    a.panel_kb = 512; a.passes = 0;
    a.w_kb_per_wg = 704;  run<0, 4, 4, false>("stream 704 KB/CU (gate/up M=1 share)", a, grid, out);
    a.w_kb_per_wg = 2816; run<0, 4, 4, false>("stream 2.8 MB/CU", a, grid, out);
    a.w_kb_per_wg = 5632; run<0, 4, 4, false>("stream 5.6 MB/CU", a, grid, out);
    a.w_kb_per_wg = 5632; run<0, 4, 8, false>("stream 5.6 MB/CU", a, grid, out);
    fprintf(out, "\n## 3. panel readers beside the weight stream (readers run until the stream of their workgroup ends)\n");
    for (long long pk : {512LL, 1408LL, 2048LL}) {
        a.panel_kb = pk; a.w_kb_per_wg = 5632;
#define BESIDE(NA, L, NW) run<NA, L, NW, false>("beside stream", a, grid, out); run<NA, L, NW, true>("beside stream", a, grid, out);
        BESIDE(2, 4, 4) BESIDE(4, 4, 4) BESIDE(4, 8, 4) BESIDE(8, 4, 4) BESIDE(4, 4, 8)
    }
    fprintf(out, "\n## 4. two workgroups per CU (grid = 2 x CUs), panel alone and beside the stream\n");
    a.panel_kb = 2048; a.w_kb_per_wg = 0; a.passes = 10;
    run<4, 4, 0, false>("alone, 2 WG/CU", a, 2 * grid, out);
    run<8, 4, 0, false>("alone, 2 WG/CU", a, 2 * grid, out);
    a.w_kb_per_wg = 2816;
    run<4, 4, 4, false>("beside stream, 2 WG/CU", a, 2 * grid, out);
    }
    // ---- 5. the tile kernels' DMA shape: 8 rows x 128 B per wave instruction out of a [256][ld] matrix (ld = 8 KiB: K = 4096; 22016 B: K = 11008),
    //         against 1 KiB contiguous per instruction (sections 1 - 4).  Same bytes, same L2 residency.
    fprintf(out, "\n## 5. 8-row x 128-byte pieces of a [256][ld] row-major matrix vs contiguous 1 KiB pieces (alone)\n");
    for (long long ld : {8192LL, 22016LL}) {
        a.ld = ld; a.panel_kb = 256 * ld / 1024; a.w_kb_per_wg = 0; a.passes = (int)(16384 / a.panel_kb) + 2;
        fprintf(out, "# ld = %lld bytes, panel %lld KB\n", ld, a.panel_kb);
        run<8, 4, 0, false, 0>("contiguous", a, grid, out); run<8, 4, 0, true, 0>("contiguous", a, grid, out);
        run<8, 4, 0, false, 1>("row pieces", a, grid, out); run<8, 4, 0, true, 1>("row pieces", a, grid, out);
        run<8, 8, 0, true, 1>("row pieces", a, grid, out);
        run<8, 8, 0, true, 0>("contiguous", a, grid, out);
        run<8, 4, 0, true, 2>("1 KiB, lanes swizzled in rows", a, grid, out);
        run<8, 4, 0, true, 3>("2 rows x 512 B", a, grid, out);
        run<8, 4, 0, true, 4>("4 rows x 256 B", a, grid, out);
        a.w_kb_per_wg = 2816;
        run<8, 4, 4, true, 0>("contiguous beside stream", a, grid, out);
        run<8, 4, 4, true, 1>("row pieces beside stream", a, grid, out);
        a.w_kb_per_wg = 0;
    }
    return 0;
}
