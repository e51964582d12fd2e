// Probe of the 2:4 structured-sparsity matrix instructions of gfx950 (v_smfmac_f32_16x16x64_bf16 / _32x32x32_bf16): operand layout,
// meaning of the index register, whether the two kept elements of a group must be ordered, and the issue rate against the dense
// v_mfma_f32_16x16x32_bf16 / _32x32x16_bf16.      hipcc --offload-arch=gfx950 -O2 smfmac_probe.hip -o smfmac_probe && ./smfmac_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// one wave: D (16 x 16) = A_sparse (16 x 64, 32 kept per row) x B (64 x 16)
__global__ void one16(const __bf16* a /*[64 lanes][8]*/, const __bf16* b /*[64][16]*/, const int* idx, float* d /*[64][4]*/, int abid) {
    const int l = threadIdx.x;
    bf16x8 av; bf16x16 bv;
    for (int j = 0; j < 8; ++j) av[j] = a[l * 8 + j];
    for (int j = 0; j < 16; ++j) bv[j] = b[l * 16 + j];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (abid == 0) acc = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(av, bv, acc, idx[l], 0, 0);
    else acc = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(av, bv, acc, idx[l], 0, 1);
    for (int j = 0; j < 4; ++j) d[l * 4 + j] = acc[j];
}
__global__ void one32(const __bf16* a, const __bf16* b, const int* idx, float* d /*[64][16]*/) {
    const int l = threadIdx.x;
    bf16x8 av; bf16x16 bv;
    for (int j = 0; j < 8; ++j) av[j] = a[l * 8 + j];
    for (int j = 0; j < 16; ++j) bv[j] = b[l * 16 + j];
    f32x16 acc = {};
    acc = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(av, bv, acc, idx[l], 0, 0);
    for (int j = 0; j < 16; ++j) d[l * 16 + j] = acc[j];
}

template <int KIND>
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
    bf16x8 a8; bf16x16 b16;
    for (int j = 0; j < 8; ++j) a8[j] = (__bf16)(0.001f * (threadIdx.x + j));
    for (int j = 0; j < 16; ++j) b16[j] = (__bf16)(0.002f * (threadIdx.x + j));
    f32x4 c4[4] = {};
    f32x16 c16[2] = {};
    const int idx = 0x44444444;        // pairs (0, 1)
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) { for (int u = 0; u < 4; ++u) c4[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, a8, c4[u], 0, 0, 0); }
        if constexpr (KIND == 1) { for (int u = 0; u < 4; ++u) c4[u] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a8, b16, c4[u], idx, 0, 0); }
        if constexpr (KIND == 2) { for (int u = 0; u < 2; ++u) c16[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, a8, c16[u], 0, 0, 0); }
        if constexpr (KIND == 3) { for (int u = 0; u < 2; ++u) c16[u] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a8, b16, c16[u], idx, 0, 0); }
    }
    float s = 0.f;
    for (int u = 0; u < 4; ++u) s += c4[u][0];
    for (int u = 0; u < 2; ++u) s += c16[u][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static __bf16 h2b(float f) { unsigned u; memcpy(&u, &f, 4); unsigned short s = (unsigned short)(u >> 16); __bf16 r; memcpy(&r, &s, 2); return r; }

int main() {
    __bf16 *a, *b; int* idx; float* d;
    hipMalloc(&a, 64 * 8 * 2); hipMalloc(&b, 64 * 16 * 2); hipMalloc(&idx, 64 * 4); hipMalloc(&d, 64 * 16 * 4);
    std::vector<__bf16> ha(64 * 8), hb(64 * 16); std::vector<int> hi(64); std::vector<float> hd(64 * 16);
    // ---- B layout: lane l, element j.  Put the value (1 + l * 16 + j) there, A = one kept element; D tells which B entry was hit.
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j) hb[l * 16 + j] = h2b((float)(1 + ((l * 16 + j) % 250)));   // bf16-exact up to 256
    hipMemcpy(b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    printf("== 16x16x64: A lane (row m = lane & 15?, k group = lane >> 4), slot s, index value v  ->  which B (lane, element) feeds D[m][n]\n");
    for (int g = 0; g < 4; ++g) for (int s = 0; s < 8; s += 1) for (int v = 0; v < 4; ++v) {
        if (g != 0 && g != 3 && !(s == 0 && v == 0)) continue;       // full sweep on two lane groups, one sample on the others
        std::fill(ha.begin(), ha.end(), h2b(0.f));
        const int la = 3 + 16 * g;                                   // row 3 of lane group g
        ha[la * 8 + s] = h2b(1.f);
        for (int l = 0; l < 64; ++l) {
            int w = 0;
            for (int t = 0; t < 8; ++t) w |= ((t & 1) ? 3 : 0) << (2 * t);        // default pairs (0, 3)
            if (l == la) {
                w &= ~(3 << (2 * s)); w |= v << (2 * s);
                const int p = s ^ 1;                                                // the partner slot of the pair: keep it different
                int pv = (v == 3) ? 0 : 3; w &= ~(3 << (2 * p)); w |= pv << (2 * p);
            }
            hi[l] = w;
        }
        hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(idx, hi.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(one16, dim3(1), dim3(64), 0, 0, a, b, idx, d, 0);
        hipMemcpy(hd.data(), d, 64 * 4 * 4, hipMemcpyDeviceToHost);
        // find nonzero outputs
        printf("g %d slot %d idx %d:", g, s, v);
        int shown = 0;
        for (int l = 0; l < 64 && shown < 3; ++l) for (int j = 0; j < 4; ++j) if (hd[l * 4 + j] != 0.f && shown < 3) { printf("  D[lane %d][reg %d] = %.0f", l, j, hd[l * 4 + j]); ++shown; }
        printf("\n");
    }
    printf("== ordering: both slots of pair 0 kept (values 1 and 2), index (v0, v1): D = 1 * B[k(v0)] + 2 * B[k(v1)] for lane group 0 row 3, column n = 0\n");
    for (int v0 = 0; v0 < 4; ++v0) for (int v1 = 0; v1 < 4; ++v1) {
        std::fill(ha.begin(), ha.end(), h2b(0.f));
        ha[3 * 8 + 0] = h2b(1.f); ha[3 * 8 + 1] = h2b(2.f);
        for (int l = 0; l < 64; ++l) { int w = 0; for (int t = 0; t < 8; ++t) w |= ((t & 1) ? 3 : 0) << (2 * t); if (l == 3) { w &= ~15; w |= v0 | (v1 << 2); } hi[l] = w; }
        hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(idx, hi.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(one16, dim3(1), dim3(64), 0, 0, a, b, idx, d, 0);
        hipMemcpy(hd.data(), d, 64 * 4 * 4, hipMemcpyDeviceToHost);
        printf("(%d,%d):", v0, v1);
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (hd[l * 4 + j] != 0.f && (l & 15) == 0) printf(" D[lane %d][reg %d] = %.0f", l, j, hd[l * 4 + j]);
        printf("\n");
    }
    printf("== abid = 1 (second index set?) with slot 0 idx 2 in bits 0..1 and idx 1 in bits 16..17\n");
    {
        std::fill(ha.begin(), ha.end(), h2b(0.f)); ha[3 * 8 + 0] = h2b(1.f);
        for (int l = 0; l < 64; ++l) { int w = 0; for (int t = 0; t < 8; ++t) w |= ((t & 1) ? 3 : 0) << (2 * t); w |= w << 16; if (l == 3) { w &= ~3; w |= 2; w &= ~(3 << 16); w |= 1 << 16; } hi[l] = w; }
        hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(idx, hi.data(), 256, hipMemcpyHostToDevice);
        for (int ab = 0; ab < 2; ++ab) {
            hipLaunchKernelGGL(one16, dim3(1), dim3(64), 0, 0, a, b, idx, d, ab);
            hipMemcpy(hd.data(), d, 64 * 4 * 4, hipMemcpyDeviceToHost);
            printf("abid %d:", ab);
            for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (hd[l * 4 + j] != 0.f && (l & 15) == 0) printf(" D[lane %d][reg %d] = %.0f", l, j, hd[l * 4 + j]);
            printf("\n");
        }
    }
    printf("== 32x32x32: A lane 5 (+32 h), slot s, idx v\n");
    for (int h = 0; h < 2; ++h) for (int s = 0; s < 8; ++s) for (int v = 0; v < 4; v += (s < 2 ? 1 : 3)) {
        std::fill(ha.begin(), ha.end(), h2b(0.f));
        const int la = 5 + 32 * h;
        ha[la * 8 + s] = h2b(1.f);
        for (int l = 0; l < 64; ++l) {
            int w = 0;
            for (int t = 0; t < 8; ++t) w |= ((t & 1) ? 3 : 0) << (2 * t);
            if (l == la) { w &= ~(3 << (2 * s)); w |= v << (2 * s); const int p = s ^ 1; int pv = (v == 3) ? 0 : 3; w &= ~(3 << (2 * p)); w |= pv << (2 * p); }
            hi[l] = w;
        }
        hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(idx, hi.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(one32, dim3(1), dim3(64), 0, 0, a, b, idx, d);
        hipMemcpy(hd.data(), d, 64 * 16 * 4, hipMemcpyDeviceToHost);
        printf("h %d slot %d idx %d:", h, s, v);
        int shown = 0;
        for (int l = 0; l < 64 && shown < 3; ++l) for (int j = 0; j < 16; ++j) if (hd[l * 16 + j] != 0.f && shown < 3) { printf("  D[lane %d][reg %d] = %.0f", l, j, hd[l * 16 + j]); ++shown; }
        printf("\n");
    }
    // ---- rates
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    const char* names[4] = {"mfma 16x16x32 bf16 (dense)", "smfmac 16x16x64 bf16", "mfma 32x32x16 bf16 (dense)", "smfmac 32x32x32 bf16"};
    const double flops[4] = {2.0 * 16 * 16 * 32 * 4, 2.0 * 16 * 16 * 64 * 4, 2.0 * 32 * 32 * 16 * 2, 2.0 * 32 * 32 * 32 * 2};
    for (int kind = 0; kind < 4; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(1024), dim3(256), 0, 0, out, iters);
            if (kind == 1) hipLaunchKernelGGL(rate<1>, dim3(1024), dim3(256), 0, 0, out, iters);
            if (kind == 2) hipLaunchKernelGGL(rate<2>, dim3(1024), dim3(256), 0, 0, out, iters);
            if (kind == 3) hipLaunchKernelGGL(rate<3>, dim3(1024), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-30s %.3f ms  %.0f TFLOP/s (counting the dense-equivalent products)\n", names[kind], ms, flops[kind] * iters * 1024 * 4 / ms / 1e9);
        }
    }
    return 0;
}
