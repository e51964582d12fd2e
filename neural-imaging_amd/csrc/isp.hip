// Element-wise pieces of the classic camera ISP (reference models/pipelines.py:416-453 `_ClassicISP`,
// models/layers.py:206-258 `DemosaicingLayer`): the residual combination y = clip_ste(x - alpha f) with its scalar
// gradient, the sigmoid of the non-residual head, and the gamma stage pow(clip_ste(x, 1/255, 1), 1/2.2).
// All HBM-bound streams: float4 grid-stride loops; the alpha gradient is a fixed-order two-stage sum (deterministic).
#include "common.h"

namespace {
using namespace nimg;

constexpr int RED_BLOCKS = 1024;

inline int grid_for(long count) {
    long g = (count + 1023) / 1024;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

__device__ __forceinline__ float clip01f(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

// y = [clip01](x - alpha * f); the clip is straight-through, so no mask is kept
__global__ void residual_fwd_kernel(const float* __restrict__ x, const float* __restrict__ f,
                                    const float* __restrict__ alpha, float* __restrict__ y, long count, int clip) {
    const float a = f ? alpha[0] : 0.f;
    const long c4 = count >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < c4; i += (long)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        if (f) {
            const float4 r = reinterpret_cast<const float4*>(f)[i];
            v.x -= a * r.x; v.y -= a * r.y; v.z -= a * r.z; v.w -= a * r.w;
        }
        if (clip) { v.x = clip01f(v.x); v.y = clip01f(v.y); v.z = clip01f(v.z); v.w = clip01f(v.w); }
        reinterpret_cast<float4*>(y)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
        const long i = (c4 << 2) + threadIdx.x;
        float v = x[i] - (f ? a * f[i] : 0.f);
        y[i] = clip ? clip01f(v) : v;
    }
}

// df = -alpha dy;  partial[block] = sum over the block's elements of dy * f
__global__ __launch_bounds__(256) void residual_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ f,
                                                           const float* __restrict__ alpha, float* __restrict__ df,
                                                           float* __restrict__ partial, long count) {
    __shared__ float red[4];
    const float a = alpha[0];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float g = dy[i];
        acc += g * f[i];
        df[i] = -a * g;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void residual_final_kernel(const float* __restrict__ partial, int blocks,
                                                            float* __restrict__ dalpha, int accumulate) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < blocks; i += 64) acc += partial[i];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) dalpha[0] = (accumulate ? dalpha[0] : 0.f) - acc;
}

__global__ void sigmoid_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long count) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        y[i] = 1.0f / (1.0f + expf(-x[i]));
}
__global__ void sigmoid_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                   long count) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float s = y[i];
        dx[i] = dy[i] * s * (1.0f - s);
    }
}

// activation_mapping of helpers/tf_helpers.py:22-28 as one element-wise pair; float4 streams with a scalar tail
template <int KIND>
__device__ __forceinline__ float act_f(float x, float alpha) {
    if (KIND == 0) return x > 0.f ? x : alpha * x;
    if (KIND == 1) return fmaxf(x, 0.f);
    if (KIND == 2) return tanhf(x);
    if (KIND == 3) return 1.0f / (1.0f + expf(-x));
    return x / (1.0f + fabsf(x));
}
template <int KIND>
__device__ __forceinline__ float act_d(float y, float alpha) {          // derivative, from the OUTPUT
    if (KIND == 0) return y > 0.f ? 1.0f : alpha;
    if (KIND == 1) return y > 0.f ? 1.0f : 0.f;
    if (KIND == 2) return 1.0f - y * y;
    if (KIND == 3) return y * (1.0f - y);
    const float t = 1.0f - fabsf(y);                                    // y = x / (1 + |x|)  ->  1 / (1 + |x|) = 1 - |y|
    return t * t;
}
template <int KIND>
__global__ void activation_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long count, float alpha) {
    const long n4 = count >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        reinterpret_cast<float4*>(y)[i] = make_float4(act_f<KIND>(v.x, alpha), act_f<KIND>(v.y, alpha), act_f<KIND>(v.z, alpha),
                                                      act_f<KIND>(v.w, alpha));
    }
    for (long i = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        y[i] = act_f<KIND>(x[i], alpha);
}
template <int KIND>
__global__ void activation_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                      long count, float alpha) {
    const long n4 = count >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 g = reinterpret_cast<const float4*>(dy)[i], v = reinterpret_cast<const float4*>(y)[i];
        reinterpret_cast<float4*>(dx)[i] = make_float4(g.x * act_d<KIND>(v.x, alpha), g.y * act_d<KIND>(v.y, alpha),
                                                       g.z * act_d<KIND>(v.z, alpha), g.w * act_d<KIND>(v.w, alpha));
    }
    for (long i = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        dx[i] = dy[i] * act_d<KIND>(y[i], alpha);
}

// y = pow(clip(x, lo, hi), e); the clip is straight-through: dx = dy e pow(clip(x), e - 1) everywhere
__global__ void gamma_ste_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long count, float lo, float hi,
                                     float e) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        y[i] = powf(fminf(fmaxf(x[i], lo), hi), e);
}
__global__ void gamma_ste_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                     long count, float lo, float hi, float e) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        dx[i] = dy[i] * e * powf(fminf(fmaxf(x[i], lo), hi), e - 1.0f);
}

// Keras Dropout at training time, forward and backward alike: y = keep[i] ? x * scale : 0 (models/forensics.py:88)
__global__ void mask_scale_kernel(const float* __restrict__ x, const uint8_t* __restrict__ keep, float* __restrict__ y,
                                  long count, float scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        y[i] = keep[i] ? x[i] * scale : 0.f;
}

// Classifier decisions and their confusion matrix on the device (training/validation.py:163-202 does this on the host,
// one batch at a time): pred = first arg-max of the probability row (numpy.argmax), conf[label][pred] += 1.
__global__ void confusion_kernel(const float* __restrict__ probs, const int* __restrict__ labels, int* __restrict__ pred,
                                 unsigned long long* __restrict__ conf, int n, int k) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float* row = probs + (long)i * k;
        int best = 0;
        float bv = row[0];
        for (int c = 1; c < k; ++c) {
            const float v = row[c];
            if (v > bv) { bv = v; best = c; }
        }
        if (pred) pred[i] = best;
        if (conf && labels) {
            const int l = labels[i];
            if (l >= 0 && l < k) atomicAdd(conf + (long)l * k + best, 1ull);
        }
    }
}

}  // namespace

extern "C" {

int nimg_confusion_accumulate(const float* probs, const int* labels, int* pred, unsigned long long* conf, int n, int k,
                              void* stream) {
    if (n < 0 || k < 1) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    if (!probs || (!pred && !conf) || (conf && !labels)) return NIMG_ERR_ARG;
    hipLaunchKernelGGL(confusion_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, probs, labels, pred, conf, n, k);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_mask_scale(const float* x, const uint8_t* keep, float* y, long count, float scale, void* stream) {
    if (count < 0) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    if (!x || !keep || !y) return NIMG_ERR_ARG;
    hipLaunchKernelGGL(mask_scale_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, x, keep, y, count, scale);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_isp_residual_fwd(const float* x, const float* f, const float* alpha, float* y, long count, int clip,
                          void* stream) {
    if (count < 0) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    if (!x || !y || (f && !alpha)) return NIMG_ERR_ARG;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)f) & 15) return NIMG_ERR_ARG;
    hipLaunchKernelGGL(residual_fwd_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, x, f, alpha, y,
                       count, clip);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

long nimg_isp_residual_workspace_bytes(void) { return (long)RED_BLOCKS * sizeof(float); }

int nimg_isp_residual_bwd(const float* dy, const float* f, const float* alpha, float* df, float* dalpha,
                          float* workspace, long count, int accumulate, void* stream) {
    if (count < 0) return NIMG_ERR_ARG;
    if (!dalpha || !alpha) return NIMG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (count == 0) {
        if (!accumulate && hipMemsetAsync(dalpha, 0, sizeof(float), s) != hipSuccess) return NIMG_ERR_LAUNCH;
        return NIMG_OK;
    }
    if (!dy || !f || !df || !workspace) return NIMG_ERR_ARG;
    const long want = (count + 255) / 256;
    const int blocks = (int)(want > RED_BLOCKS ? RED_BLOCKS : want);
    hipLaunchKernelGGL(residual_bwd_kernel, dim3(blocks), dim3(256), 0, s, dy, f, alpha, df, workspace, count);
    hipLaunchKernelGGL(residual_final_kernel, dim3(1), dim3(64), 0, s, workspace, blocks, dalpha, accumulate);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_sigmoid_fwd(const float* x, float* y, long count, void* stream) {
    if (count < 0) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    if (!x || !y) return NIMG_ERR_ARG;
    hipLaunchKernelGGL(sigmoid_fwd_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, x, y, count);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_sigmoid_bwd(const float* dy, const float* y, float* dx, long count, void* stream) {
    if (count < 0) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    if (!dy || !y || !dx) return NIMG_ERR_ARG;
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, count);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_activation_fwd(const float* x, float* y, long count, int kind, float alpha, void* stream) {
    if (count < 0 || kind < 0 || kind > 4) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    if (!x || !y || (((uintptr_t)x | (uintptr_t)y) & 15)) return NIMG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int g = grid_for((count + 3) / 4);
    switch (kind) {
        case 0: hipLaunchKernelGGL(activation_fwd_kernel<0>, dim3(g), dim3(256), 0, s, x, y, count, alpha); break;
        case 1: hipLaunchKernelGGL(activation_fwd_kernel<1>, dim3(g), dim3(256), 0, s, x, y, count, alpha); break;
        case 2: hipLaunchKernelGGL(activation_fwd_kernel<2>, dim3(g), dim3(256), 0, s, x, y, count, alpha); break;
        case 3: hipLaunchKernelGGL(activation_fwd_kernel<3>, dim3(g), dim3(256), 0, s, x, y, count, alpha); break;
        default: hipLaunchKernelGGL(activation_fwd_kernel<4>, dim3(g), dim3(256), 0, s, x, y, count, alpha); break;
    }
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_activation_bwd(const float* dy, const float* y, float* dx, long count, int kind, float alpha, void* stream) {
    if (count < 0 || kind < 0 || kind > 4) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    if (!dy || !y || !dx || (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dx) & 15)) return NIMG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int g = grid_for((count + 3) / 4);
    switch (kind) {
        case 0: hipLaunchKernelGGL(activation_bwd_kernel<0>, dim3(g), dim3(256), 0, s, dy, y, dx, count, alpha); break;
        case 1: hipLaunchKernelGGL(activation_bwd_kernel<1>, dim3(g), dim3(256), 0, s, dy, y, dx, count, alpha); break;
        case 2: hipLaunchKernelGGL(activation_bwd_kernel<2>, dim3(g), dim3(256), 0, s, dy, y, dx, count, alpha); break;
        case 3: hipLaunchKernelGGL(activation_bwd_kernel<3>, dim3(g), dim3(256), 0, s, dy, y, dx, count, alpha); break;
        default: hipLaunchKernelGGL(activation_bwd_kernel<4>, dim3(g), dim3(256), 0, s, dy, y, dx, count, alpha); break;
    }
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_gamma_ste_fwd(const float* x, float* y, long count, float lo, float hi, float exponent, void* stream) {
    if (count < 0 || !(lo > 0.f) || !(hi >= lo)) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    if (!x || !y) return NIMG_ERR_ARG;
    hipLaunchKernelGGL(gamma_ste_fwd_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, x, y, count, lo,
                       hi, exponent);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_gamma_ste_bwd(const float* x, const float* dy, float* dx, long count, float lo, float hi, float exponent,
                       void* stream) {
    if (count < 0 || !(lo > 0.f) || !(hi >= lo)) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    if (!x || !dy || !dx) return NIMG_ERR_ARG;
    hipLaunchKernelGGL(gamma_ste_bwd_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, count,
                       lo, hi, exponent);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

}  // extern "C"
