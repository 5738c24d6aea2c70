#!/usr/bin/env python
"""Floor of a dependent launch on this GPU: a chain of kernels that do (almost) nothing, at the grids the B = 1 decode GEMVs use.
The per-launch time of the chain is what five launches per layer cost before any byte is streamed."""
import os, sys, torch
from torch.utils.cpp_extension import load_inline
src = r'''
#include <hip/hip_runtime.h>
__global__ void k_empty(float* p) { if (p && threadIdx.x == 0 && blockIdx.x == 0 && p[0] == 123.f) p[1] = 1.f; }
__global__ void k_touch(const float* x, float* y) { // 8 KB read + 1 store per workgroup: the x staging of a GEMV without the weights
    __shared__ float s[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) s[i] = x[i];
    __syncthreads();
    if (threadIdx.x == 0) y[blockIdx.x] = s[blockIdx.x & 2047];
}
void run(int which, int grid, int block, int n, uint64_t x, uint64_t y, uint64_t stream) {
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < n; ++i) {
        if (which == 0) {
            k_empty<<<grid, block, 0, s>>>((float*)y);
        } else {
            k_touch<<<grid, block, 0, s>>>((const float*)x, (float*)y);
        }
    }
}
'''
os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
m = load_inline("launch_floor", cpp_sources="void run(int which, int grid, int block, int n, uint64_t x, uint64_t y, uint64_t stream);",
                cuda_sources=src, functions=["run"], verbose=False, build_directory=None)
x = torch.randn(4096, device="cuda"); y = torch.zeros(4096, device="cuda")
for which, nm in ((0, "empty"), (1, "x-staging only")):
    for grid in (32, 512, 768):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            m.run(which, grid, 512, 10, x.data_ptr(), y.data_ptr(), s.cuda_stream)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                m.run(which, grid, 512, 160, x.data_ptr(), y.data_ptr(), torch.cuda.current_stream().cuda_stream)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): g.replay()
            e1.record(); torch.cuda.synchronize()
            print(f"{nm:16s} grid {grid:4d} x 512 threads: {e0.elapsed_time(e1) / 20 / 160 * 1e3:.2f} us per dependent launch (hipGraph chain of 160)")
