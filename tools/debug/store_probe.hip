// store_probe.hip -- how fast can ONE workgroup per CU (512 threads) write its 256 x 256 bf16 output tile (128 KiB), by store shape?
// The 256 x 256 GEMM epilogue takes 6 - 7.5 us per tile with 16-byte stores that cover 16 rows x 64 B per wave instruction, and staggering
// the workgroups over time does not shorten it (profiles/r03_gemm256_ab.txt, run 29): a per-CU limit, not HBM bandwidth.  Shapes:
//   0: 16 rows x 64 B per wave instruction (the epilogue's shape)        1: 8 rows x 128 B (full cache lines)
//   2: 2 rows x 512 B (a whole tile row per 32 lanes)                    3: shape 1 with the non-temporal policy      4: shape 0, 8-byte stores
// build: hipcc --offload-arch=gfx950 -O3 -o tools/debug/store_probe tools/debug/store_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

template <int SHAPE>
__global__ __launch_bounds__(512) void probe(unsigned short* __restrict__ out, int ldc, int tiles_n, int stagger_mask) {
    const int tile = blockIdx.x, tm = tile / tiles_n, tn = tile % tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned short* base = out + (size_t)tm * 256 * ldc + tn * 256;
    const u32x4 v = {(unsigned)tid, (unsigned)tile, 0x3f803f80u, 0x40004000u};
    if (SHAPE == 0 || SHAPE == 4) {
        // wave (wm, wn) owns 128 rows x 64 cols; per instruction: rows i*16 + (lane & 15), 64 B at column half p
        const int wm = wave >> 2, wn = wave & 3;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                unsigned short* d = base + (size_t)(wm * 128 + i * 16 + (lane & 15)) * ldc + wn * 64 + p * 32 + (lane >> 4) * 8;
                if (SHAPE == 0) *reinterpret_cast<u32x4*>(d) = v;
                else { *reinterpret_cast<u32x2*>(d) = u32x2{v.x, v.y}; *reinterpret_cast<u32x2*>(d + 4) = u32x2{v.z, v.w}; }
            }
    } else if (SHAPE == 1 || SHAPE == 3) {
        // the same sub-tile, per instruction: 8 rows x 128 B (lane -> row lane >> 3, 16-byte chunk lane & 7)
        const int wm = wave >> 2, wn = wave & 3;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            unsigned short* d = base + (size_t)(wm * 128 + i * 8 + (lane >> 3)) * ldc + wn * 64 + (lane & 7) * 8;
            if (SHAPE == 1) *reinterpret_cast<u32x4*>(d) = v;
            else __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(d));
        }
    } else {
        // whole tile rows: wave w owns rows w*32 .. w*32+31; per instruction 2 rows x 512 B
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            unsigned short* d = base + (size_t)(wave * 32 + i * 2 + (lane >> 5)) * ldc + (lane & 31) * 8;
            *reinterpret_cast<u32x4*>(d) = v;
        }
    }
}

template <int SHAPE> static float run(unsigned short* out, int M, int N, int reps) {
    const int tiles_n = N / 256, tiles = (M / 256) * tiles_n;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<SHAPE><<<tiles, 512>>>(out, N, tiles_n, 0);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) probe<SHAPE><<<tiles, 512>>>(out, N, tiles_n, 0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

int main() {
    for (int cfg = 0; cfg < 4; ++cfg) {
        // 256 tiles = one round; 1024 tiles = four rounds; 32 / 8 tiles: a few CUs storing alone (the per-CU limit)
        const int M = cfg == 2 ? 2048 : (cfg == 3 ? 512 : 16384), N = cfg == 1 ? 4096 : 1024;
        unsigned short* out; hipMalloc(&out, (size_t)M * N * 2 * 4);
        const double mb = (double)M * N * 2 / 1e6;
        const float t0 = run<0>(out, M, N, 50), t1 = run<1>(out, M, N, 50), t2 = run<2>(out, M, N, 50), t3 = run<3>(out, M, N, 50), t4 = run<4>(out, M, N, 50);
        printf("M=%d N=%d (%d tiles, %.0f MB): 16r x 64B %.1f us (%.2f TB/s) | 8r x 128B %.1f us (%.2f) | 2r x 512B %.1f us (%.2f) | 8r x 128B nt %.1f us (%.2f) | 16r x 64B as 8-byte stores %.1f us (%.2f)\n",
               M, N, (M / 256) * (N / 256), mb, t0, mb / t0, t1, mb / t1, t2, mb / t2, t3, mb / t3, t4, mb / t4);
        hipFree(out);
    }
    return 0;
}
