import torch, os, subprocess, ctypes, tempfile
src = r'''
#include <hip/hip_runtime.h>
extern "C" __global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = threadIdx.x + 1000;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
}
extern "C" void run(unsigned* out) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out); hipDeviceSynchronize(); }
'''
d = tempfile.mkdtemp()
open(d + "/p.hip", "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", d + "/p.hip", "-o", d + "/p.so"])
lib = ctypes.CDLL(d + "/p.so")
out = torch.zeros(128, dtype=torch.int32, device="cuda")
lib.run(ctypes.c_void_p(out.data_ptr()))
o = out.cpu().tolist()
print("vdst':", o[:64])
print("src' :", o[64:])
