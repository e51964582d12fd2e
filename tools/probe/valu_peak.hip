// Sustained f32 FMA issue rate of gfx950 by instruction form and occupancy: long kernels, wall-clock (HIP events).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/valu_peak.hip -o tools/probe/valu_peak && tools/probe/valu_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s0, float s1, float s2, float s3) {
    float a[24], x[8];
    for (int i = 0; i < 24; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.5f + i;
    f2 p[12], q[4];
    for (int i = 0; i < 12; ++i) p[i] = (f2){a[2 * i], a[2 * i + 1]};
    for (int i = 0; i < 4; ++i) q[i] = (f2){x[2 * i], x[2 * i + 1]};
    f2 sp = (f2){s2, s3};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 24; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(i & 1 ? s0 : s1), "v"(x[i & 7]));
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 24; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x[(i + 3) & 7]), "v"(x[i & 7]));
            } else if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 12; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(q[i & 3]), "v"(q[(i + 1) & 3]));
            } else if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 12; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "s"(sp), "v"(q[i & 3]));
            } else {        // VOP3 v_fma_f32 with the SGPR as src1
#pragma unroll
                for (int i = 0; i < 24; ++i) asm volatile("v_fma_f32 %0, %2, %1, %0" : "+v"(a[i]) : "s"(i & 1 ? s0 : s1), "v"(x[i & 7]));
            }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 24; ++i) r += a[i];
    for (int i = 0; i < 12; ++i) r += p[i][0] + p[i][1];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xfffff] = r;
}

template <int MODE>
void run(const char* name, int fma_per_instr, int instr_per_rep) {
    float* out;
    (void)hipMalloc(&out, 1 << 22);
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps, iters = 4096;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 64, 1.0001f, 0.9999f, 1.0002f, 0.9998f);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.9999f, 1.0002f, 0.9998f);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)blocks * 4 * iters * 4 * instr_per_rep;          // wave instructions on the chip
        const double per_simd_ns = ms * 1e6 / (instr / 1024.0);
        printf("%-28s %d waves/SIMD: %7.3f ms  %.3f ns per instr per SIMD (%.2f cycles @2.4 GHz)  %.1f TFLOP/s\n", name, wps, ms,
               per_simd_ns, per_simd_ns * 2.4, instr * 64 * fma_per_instr * 2 / (ms * 1e-3) / 1e12);
    }
    (void)hipFree(out);
}

int main() {
    run<1>("v_fmac_f32 v, v, v", 1, 24);
    run<0>("v_fmac_f32 v, s, v", 1, 24);
    run<4>("v_fma_f32 v, v, s, v (VOP3)", 1, 24);
    run<2>("v_pk_fma_f32 v2, v2, v2", 2, 12);
    run<3>("v_pk_fma_f32 v2, s2, v2", 2, 12);
    return 0;
}
