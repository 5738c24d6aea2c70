// probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read): which element does lane l, slot j receive, as a function of the
// per-lane addresses?  LDS holds lds[i] = i (16-bit); lane l supplies byte address addr[l]; out[l*4+j] = the value it received.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_v4_t;
__global__ void tr_probe_kernel(const unsigned* addr, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)((__attribute__((address_space(3))) char*)lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
extern "C" int tr_probe(const unsigned* addr, unsigned short* out) {
    tr_probe_kernel<<<1, 64>>>(addr, out);
    return (int)hipDeviceSynchronize();
}
