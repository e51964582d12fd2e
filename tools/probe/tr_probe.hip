// Probe the lane/element mapping of ds_read_b64_tr_b16 on gfx950: LDS holds u16 value = its own element index.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out, int stride_elems) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int lane = threadIdx.x;
    // lane supplies an 8-byte aligned address: row = lane / 4 (stride_elems apart), 4-element group = lane % 4
    uint16_t* addr = lds + (lane / 4) * stride_elems + (lane % 4) * 4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)addr);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {16, 32}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
        uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride %d elems; lane supplies addr row=lane/4, col4=(lane%%4)*4\n", stride);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l * 4 + j] / stride, h[l * 4 + j] % stride);
            printf("\n");
        }
    }
    return 0;
}
