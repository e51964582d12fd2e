// GPU-resident training-data feed (reference helpers/dataset.py:89-131 `Dataset.next_training_batch`,
// helpers/loading.py:132-211 `sample_patch`).  The full-resolution images stay in HBM as the reference keeps them in
// host memory - RAW as uint16 RGGB stacks (N, H/2, W/2, 4), RGB as uint8 (N, H, W, 3) - and a batch is cut on the device:
//   1. patch_stats:  variance / mean of every candidate patch (B images x A attempts), exact integer moments;
//   2. patch_select: the reference's discard policy ('flat', 'flat-aggressive', 'dark-n-textured') walked over each
//                    image's candidate list - the sequential part of sample_patch, one lane per image;
//   3. patch_gather: crop + normalise (uint16 / 65535, uint8 / 255, rounded like numpy's float64 division followed by
//                    the float32 store) into the NHWC float32 batch.
// No host round trip: the candidates come from a device RNG (or from the host, for parity tests).  HBM-bound byte work:
// 2-byte / 8-byte loads along the contiguous row, float4 stores.
#include "common.h"

namespace {
using namespace nimg;

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one workgroup per candidate: S = sum of the 3 p^2 byte values, SS = sum of their squares (exact);
// var = (n SS - S^2) / (n^2 255^2), mean = S / (255 n)   [np.var / np.mean of patch / 255, loading.py:166-168]
__global__ __launch_bounds__(256) void patch_stats_kernel(const uint8_t* __restrict__ rgb, int H, int W,
                                                          const int* __restrict__ image_idx,
                                                          const int* __restrict__ cand_xy, int A, int patch,
                                                          double* __restrict__ var_out, double* __restrict__ mean_out) {
    __shared__ unsigned long long red[8];
    const int cand = blockIdx.x, b = cand / A;
    const int xx = cand_xy[2 * cand], yy = cand_xy[2 * cand + 1];
    const uint8_t* img = rgb + (size_t)image_idx[b] * H * W * 3;
    const int row_pairs = (3 * patch) >> 1;                       // patch and xx are even: rows are 2-byte aligned
    const int total = patch * row_pairs;
    unsigned long long s = 0, ss = 0;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int r = i / row_pairs, c = i % row_pairs;
        const unsigned short v = *reinterpret_cast<const unsigned short*>(img + ((size_t)(yy + r) * W + xx) * 3 + 2 * c);
        const unsigned a = v & 255u, d = v >> 8;
        s += a + d;
        ss += a * a + d * d;
    }
    s = wave_sum_u64(s);
    ss = wave_sum_u64(ss);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; red[4 + (threadIdx.x >> 6)] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long S = red[0] + red[1] + red[2] + red[3], SS = red[4] + red[5] + red[6] + red[7];
        const unsigned long long n = (unsigned long long)patch * patch * 3;
        const unsigned long long num = n * SS - S * S;                 // >= 0 (Cauchy-Schwarz), exact
        var_out[cand] = (double)num / ((double)n * (double)n * 65025.0);
        mean_out[cand] = (double)S / ((double)n * 255.0);
    }
}

// the discard policy of sample_patch over attempt k = 0, 1, ... of one image; mode 0 none, 1 flat, 2 flat-aggressive,
// 3 dark-n-textured.  If the A candidates run out before the policy settles (only 'flat' can do that, by losing its coin
// flip more than A - max_attempts times) the last candidate is taken.
__global__ void patch_select_kernel(const int* __restrict__ cand_xy, const float* __restrict__ uniforms,
                                    const double* __restrict__ var, const double* __restrict__ mean, int B, int A,
                                    int max_attempts, int mode, int* __restrict__ chosen_xy,
                                    int* __restrict__ attempts_used) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int panic = max_attempts, pick = A - 1, used = A;
    int best = 0;
    double best_var = 0.0, best_mean = 0.0;
    for (int k = 0; k < A; ++k) {
        const int c = b * A + k;
        const double v = mode ? var[c] : 0.0, m = mode ? mean[c] : 0.0;       // mode 0 passes no statistics
        bool found = true;
        int at = k;
        if (mode == 1) {
            if (v < 0.005) { panic -= 1; found = !(panic > 0); }
            else if (v < 0.01) found = uniforms[c] > 0.5f;
        } else if (mode == 2) {
            if (v < 0.02) {
                if (panic == max_attempts || v > best_var) { best = k; best_var = v; }
                panic -= 1;
                found = !(panic > 0);
                if (found) at = best;
            }
        } else if (mode == 3) {
            if (!(0.0 < v && v < 0.005 && 0.35 < m && m < 0.99)) {
                if (panic == max_attempts || (v < 2.0 * best_var && m > 1.1 * best_mean)) { best = k; best_mean = m; best_var = v; }
                panic -= 1;
                found = !(panic > 0);
                if (found) at = best;
            }
        }
        if (found) { pick = at; used = k + 1; break; }
    }
    chosen_xy[2 * b] = cand_xy[2 * (b * A + pick)];
    chosen_xy[2 * b + 1] = cand_xy[2 * (b * A + pick) + 1];
    if (attempts_used) attempts_used[b] = used;
}

// x[b] = raw[image][yy/2 : yy/2 + p/2, xx/2 : xx/2 + p/2] / 65535;  one lane per RAW pixel (4 planes = 8 bytes)
__global__ void gather_raw_kernel(const uint16_t* __restrict__ raw, int H2, int W2, const int* __restrict__ image_idx,
                                  const int* __restrict__ xy, int B, int ps, float* __restrict__ out) {
    const long total = (long)B * ps * ps;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ps), r = (int)((i / ps) % ps), b = (int)(i / ((long)ps * ps));
        const int rx = xy[2 * b] >> 1, ry = xy[2 * b + 1] >> 1;
        const uint2 v = *reinterpret_cast<const uint2*>(raw + (((size_t)image_idx[b] * H2 + ry + r) * W2 + rx + c) * 4);
        float4 o;
        o.x = (float)((double)(v.x & 0xFFFFu) / 65535.0);
        o.y = (float)((double)(v.x >> 16) / 65535.0);
        o.z = (float)((double)(v.y & 0xFFFFu) / 65535.0);
        o.w = (float)((double)(v.y >> 16) / 65535.0);
        reinterpret_cast<float4*>(out)[i] = o;
    }
}

// y[b] = rgb[image][yy : yy + p, xx : xx + p] / 255;  one lane per 2 bytes of a row (rows are 2-byte aligned)
__global__ void gather_rgb_kernel(const uint8_t* __restrict__ rgb, int H, int W, const int* __restrict__ image_idx,
                                  const int* __restrict__ xy, int B, int p, float* __restrict__ out) {
    const int row_pairs = (3 * p) >> 1;
    const long total = (long)B * p * row_pairs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % row_pairs), r = (int)((i / row_pairs) % p), b = (int)(i / ((long)row_pairs * p));
        const int xx = xy[2 * b], yy = xy[2 * b + 1];
        const unsigned short v = *reinterpret_cast<const unsigned short*>(
            rgb + (((size_t)image_idx[b] * H + yy + r) * W + xx) * 3 + 2 * c);
        float2 o;
        o.x = (float)((double)(v & 255u) / 255.0);
        o.y = (float)((double)(v >> 8) / 255.0);
        reinterpret_cast<float2*>(out)[i] = o;
    }
}

inline int grid_for(long count) {
    long g = (count + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" {

int nimg_patch_stats(const uint8_t* rgb, int n_images, int h, int w, const int* image_idx, const int* cand_xy, int b,
                     int attempts, int patch, double* var_out, double* mean_out, void* stream) {
    if (b < 0 || attempts <= 0) return NIMG_ERR_ARG;
    if (b == 0) return NIMG_OK;
    if (!rgb || !image_idx || !cand_xy || !var_out || !mean_out || n_images <= 0) return NIMG_ERR_ARG;
    if (patch <= 0 || (patch & 1) || patch > h || patch > w || patch > 1024) return NIMG_ERR_ARG;   /* n SS - S^2 stays below 2^64 */
    hipLaunchKernelGGL(patch_stats_kernel, dim3(b * attempts), dim3(256), 0, (hipStream_t)stream, rgb, h, w, image_idx,
                       cand_xy, attempts, patch, var_out, mean_out);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_patch_select(const int* cand_xy, const float* uniforms, const double* var, const double* mean, int b,
                      int attempts, int max_attempts, int mode, int* chosen_xy, int* attempts_used, void* stream) {
    if (b < 0 || attempts <= 0 || max_attempts <= 0 || mode < 0 || mode > 3) return NIMG_ERR_ARG;
    if (b == 0) return NIMG_OK;
    if (!cand_xy || !chosen_xy) return NIMG_ERR_ARG;
    if (mode != 0 && (!var || !mean)) return NIMG_ERR_ARG;
    if (mode == 1 && !uniforms) return NIMG_ERR_ARG;
    hipLaunchKernelGGL(patch_select_kernel, dim3((b + 63) / 64), dim3(64), 0, (hipStream_t)stream, cand_xy, uniforms, var,
                       mean, b, attempts, max_attempts, mode, chosen_xy, attempts_used);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_patch_gather(const uint16_t* raw, const uint8_t* rgb, int n_images, int h, int w, const int* image_idx,
                      const int* xy, int b, int patch, float* x_out, float* y_out, void* stream) {
    if (b < 0) return NIMG_ERR_ARG;
    if (b == 0) return NIMG_OK;
    if (!image_idx || !xy || n_images <= 0 || patch <= 0 || (patch & 1) || (h & 1) || (w & 1) || patch > h || patch > w)
        return NIMG_ERR_ARG;
    if ((x_out && !raw) || (y_out && !rgb) || (!x_out && !y_out)) return NIMG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (x_out) {
        const int ps = patch / 2;
        hipLaunchKernelGGL(gather_raw_kernel, dim3(grid_for((long)b * ps * ps)), dim3(256), 0, s, raw, h / 2, w / 2,
                           image_idx, xy, b, ps, x_out);
        NIMG_CHECK_LAUNCH();
    }
    if (y_out) {
        hipLaunchKernelGGL(gather_rgb_kernel, dim3(grid_for((long)b * patch * ((3 * patch) >> 1))), dim3(256), 0, s, rgb,
                           h, w, image_idx, xy, b, patch, y_out);
        NIMG_CHECK_LAUNCH();
    }
    return NIMG_OK;
}

}  // extern "C"
