// tr_read_probe.hip -- which LDS element does each (lane, element) of ds_read_b64_tr_b16 return?
//   hipcc --offload-arch=gfx950 -O2 tr_read_probe.hip -o tr_read_probe && ./tr_read_probe
// LDS holds lds[i] = i (16-bit).  Lane l passes the byte address 8 * perm(l) (its own 4 contiguous elements);
// the output lists, per lane, the 4 returned ids, from which the (source lane, source element) of each is read off.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void probe(short *out, int mode) {
    __shared__ short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int slot = l;                       // mode 0: lane l -> elements 4l..4l+3
    if (mode == 1) slot = 63 - l;       // mode 1: reversed, to separate "lane" from "address"
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(lds + slot * 4));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short *d, h[256];
    hipMalloc(&d, sizeof(h));
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d: lane: returned element ids (id = 4*source_slot + source_elem)\n", mode);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    }
    return 0;
}
