// Micro-benchmark: issue rate of f32 FMA forms on gfx950 (cycles per wave64 instruction, one and four waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/fma_rate.hip -o /tmp/fma_rate && /tmp/fma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, long long* cyc, float s0, float s1, float s2, float s3) {
    float a[24];
    for (int i = 0; i < 24; ++i) a[i] = threadIdx.x * 0.001f + i;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.5f + i;
    f2 p[12], q[4];
    for (int i = 0; i < 12; ++i) p[i] = (f2){a[2 * i], a[2 * i + 1]};
    for (int i = 0; i < 4; ++i) q[i] = (f2){x[2 * i], x[2 * i + 1]};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 256; ++it) {
        if (MODE == 0) {            // v_fmac v, s, v  (24 independent accumulators)
#pragma unroll
            for (int i = 0; i < 24; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(i & 1 ? s0 : s1), "v"(x[i & 7]));
        } else if (MODE == 1) {     // v_fmac v, v, v
#pragma unroll
            for (int i = 0; i < 24; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x[(i + 3) & 7]), "v"(x[i & 7]));
        } else if (MODE == 2) {     // v_pk_fma_f32 v2, v2, v2
#pragma unroll
            for (int i = 0; i < 12; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(q[i & 3]), "v"(q[(i + 1) & 3]));
        } else {                    // v_pk_fma_f32 with an SGPR pair
            f2 sp = (f2){s2, s3};
#pragma unroll
            for (int i = 0; i < 12; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "s"(sp), "v"(q[i & 3]));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
    for (int i = 0; i < 24; ++i) r += a[i];
    for (int i = 0; i < 12; ++i) r += p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, int ninstr) {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 4096 * 8);
    const int blocks = 256;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.9999f, 1.0002f, 0.9998f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.9999f, 1.0002f, 0.9998f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
    const double per = avg / (256.0 * ninstr);
    printf("%-34s %4d thr/WG (%d waves/SIMD): %.2f counter ticks per instr per wave, kernel %.3f ms\n", name, threads, threads / 256,
           per, ms);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int thr : {256, 1024}) {
        run<0>("v_fmac_f32 v, s, v", thr, 24);
        run<1>("v_fmac_f32 v, v, v", thr, 24);
        run<2>("v_pk_fma_f32 v2, v2, v2", thr, 12);
        run<3>("v_pk_fma_f32 v2, s2, v2", thr, 12);
    }
    return 0;
}
