// ConstrainedConv2D kernel re-normalisation (models/layers.py:45-53), forward and backward.
//   nf = K * (1 - M);  df[o] = sum_{h,w,i} nf[h,w,i,o];  nf = strength * nf / df - strength * M
// with M the centre mask on the (i,i) diagonal (helpers/kernels.py:117-123).  Layout (kh,kw,Cin,Cout), 5x5x3x3 = 225
// elements; a single workgroup.  The convolution itself (SYMMETRIC pad 2 + VALID conv, layers.py:56-57) runs on the
// generic conv kernels with pad_mode = SYMMETRIC.
#include "common.h"

namespace {

__device__ __forceinline__ bool centre_mask(int e, int ks, int c) {
    // e indexes (kh,kw,ci,co) flattened
    const int co = e % c, ci = (e / c) % c, kw = (e / (c * c)) % ks, kh = e / (c * c * ks);
    return kh == ks / 2 && kw == ks / 2 && ci == co;
}

__global__ __launch_bounds__(256) void constrained_fwd_kernel(const float* __restrict__ k, float* __restrict__ nf,
                                                              int ks, int c, float strength) {
    __shared__ float df[16];
    const int total = ks * ks * c * c, tid = threadIdx.x;
    if (tid < c) {
        float s = 0.f;
        for (int e = tid; e < total; e += c)            // all elements with co == tid
            if (!centre_mask(e, ks, c)) s += k[e];
        df[tid] = s;
    }
    __syncthreads();
    for (int e = tid; e < total; e += blockDim.x) {
        const bool m = centre_mask(e, ks, c);
        nf[e] = m ? -strength : strength * k[e] / df[e % c];
    }
}

__global__ __launch_bounds__(256) void constrained_bwd_kernel(const float* __restrict__ k,
                                                              const float* __restrict__ dnf, float* __restrict__ dk,
                                                              int ks, int c, float strength) {
    __shared__ float df[16], dot[16];
    const int total = ks * ks * c * c, tid = threadIdx.x;
    if (tid < c) {
        float s = 0.f, d = 0.f;
        for (int e = tid; e < total; e += c)
            if (!centre_mask(e, ks, c)) { s += k[e]; d += dnf[e] * k[e]; }
        df[tid] = s;
        dot[tid] = d;
    }
    __syncthreads();
    for (int e = tid; e < total; e += blockDim.x) {
        const int co = e % c;
        dk[e] = centre_mask(e, ks, c) ? 0.f : strength * (dnf[e] / df[co] - dot[co] / (df[co] * df[co]));
    }
}

}  // namespace

extern "C" {

int nimg_constrained_kernel_fwd(const float* kernel, float* nf, int ks, int channels, float strength, void* stream) {
    if (!kernel || !nf || ks <= 0 || channels <= 0 || channels > 16) return NIMG_ERR_ARG;
    hipLaunchKernelGGL(constrained_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, kernel, nf, ks, channels,
                       strength);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_constrained_kernel_bwd(const float* kernel, const float* dnf, float* dkernel, int ks, int channels,
                                float strength, void* stream) {
    if (!kernel || !dnf || !dkernel || ks <= 0 || channels <= 0 || channels > 16) return NIMG_ERR_ARG;
    hipLaunchKernelGGL(constrained_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, kernel, dnf, dkernel, ks,
                       channels, strength);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

}  // extern "C"
