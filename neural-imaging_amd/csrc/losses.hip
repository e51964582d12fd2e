// The L1 and SSIM training losses of the imaging pipelines (reference helpers/tf_helpers.py:35-40, selected by
// NIPModel.construct_loss, models/pipelines.py:53-63), each with the gradient w.r.t. the developed image.
//   mae255:    loss = mean |255 y - 255 t|;                        dy (+)= gscale * 255 sign(y - t) / count
//   ssim_loss: loss = mean_n 255 (1 - tf.image.ssim(y, t, max_val)_n)   (11x11 Gaussian window, sigma 1.5, VALID)
// SSIM backward: with the window moments m = (E y, E t, E yy, E tt, E yt) at window position q,
//   S = (2 Ey Et + c1)(2 (Eyt - Ey Et) + c2) / ((Ey^2 + Et^2 + c1)(Eyy - Ey^2 + Ett - Et^2 + c2)),
// and every moment is a Gaussian-weighted sum of the pixels, so
//   d loss / d y(p) = sum_q g(p - q) [ dS/dEy(q) + 2 y(p) dS/dEyy(q) + t(p) dS/dEyt(q) ] * (-255 / items).
// Pass 1 writes the three derivative maps (already scaled) and the per-workgroup sums of S (double, fixed order);
// pass 2 gathers them through the transposed window.  HBM/L2-bound gathers; not on the default (L2) training path.
#include "common.h"

namespace {
using namespace nimg;

constexpr int WIN = 11;
constexpr int BPI = 64;          // workgroups per image in pass 1

inline int grid_for(long count) {
    long g = (count + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

__global__ __launch_bounds__(256) void mae255_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     float* grad_a, double* __restrict__ partial, long count,
                                                     float gscale, int accumulate) {
    __shared__ double red[4];
    double s = 0.0;
    const float gk = gscale * 255.0f / (float)count;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float d = 255.0f * a[i] - 255.0f * b[i];
        s += (double)fabsf(d);
        const float g = d > 0.f ? gk : (d < 0.f ? -gk : 0.f);               // tf.abs gradient: sign(d), sign(0) = 0
        if (grad_a) grad_a[i] = accumulate ? grad_a[i] + g : g;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void mean_final_kernel(const double* __restrict__ partial, int nblocks, double inv_count, double scale,
                                  double offset, float* loss) {
    double s = 0.0;
    for (int k = threadIdx.x; k < nblocks; k += 64) s += partial[k];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) loss[0] = (float)(offset + scale * s * inv_count);
}

__global__ __launch_bounds__(256) void ssim_loss_stats_kernel(const float* __restrict__ y, const float* __restrict__ t,
                                                              double* __restrict__ partial, float* __restrict__ maps,
                                                              long plane, int h, int w, int c, float max_val,
                                                              const float* __restrict__ gk, double coef) {
    __shared__ double red[4];
    __shared__ float gs[WIN * WIN];
    if (threadIdx.x < WIN * WIN) gs[threadIdx.x] = gk[threadIdx.x];
    __syncthreads();
    const int n = blockIdx.x / BPI, blk = blockIdx.x % BPI;
    const int ho = h - WIN + 1, wo = w - WIN + 1;
    const long items = (long)ho * wo * c;
    const double c1 = (0.01 * max_val) * (0.01 * max_val), c2 = (0.03 * max_val) * (0.03 * max_val);
    double sum = 0.0;
    for (long i = (long)blk * 256 + threadIdx.x; i < items; i += (long)BPI * 256) {
        const int ch = (int)(i % c), x0 = (int)((i / c) % wo), y0 = (int)(i / ((long)c * wo));
        double ey = 0, et = 0, eyy = 0, ett = 0, eyt = 0;
        for (int dy = 0; dy < WIN; ++dy) {
            const long row = (((long)n * h + y0 + dy) * w + x0) * c + ch;
#pragma unroll
            for (int dx = 0; dx < WIN; ++dx) {
                const double wt = (double)gs[dy * WIN + dx];
                const double vy = y[row + (long)dx * c], vt = t[row + (long)dx * c];
                ey += wt * vy; et += wt * vt; eyy += wt * vy * vy; ett += wt * vt * vt; eyt += wt * vy * vt;
            }
        }
        const double a1 = 2 * ey * et + c1, a2 = 2 * (eyt - ey * et) + c2;
        const double b1 = ey * ey + et * et + c1, b2 = (eyy - ey * ey) + (ett - et * et) + c2;
        const double inv = 1.0 / (b1 * b2), s = a1 * a2 * inv;
        sum += s;
        if (maps) {
            const long o = (long)n * items + i;
            maps[o] = (float)(coef * (2 * et * (a2 - a1) * inv - s * 2 * ey * (b2 - b1) * inv));   // dS / dEy
            maps[plane + o] = (float)(coef * (-s / b2));                                            // dS / dEyy
            maps[2 * plane + o] = (float)(coef * (2 * a1 * inv));                                   // dS / dEyt
        }
    }
    sum = wave_sum_d(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void ssim_loss_grad_kernel(const float* __restrict__ y, const float* __restrict__ t,
                                                             const float* __restrict__ maps, long plane,
                                                             float* grad, int n, int h, int w, int c,
                                                             const float* __restrict__ gk, float gscale,
                                                             int accumulate, const float* __restrict__ coef) {
    __shared__ float gs[WIN * WIN];
    if (threadIdx.x < WIN * WIN) gs[threadIdx.x] = gk[threadIdx.x];
    __syncthreads();
    const int ho = h - WIN + 1, wo = w - WIN + 1;
    const long total = (long)n * h * w * c;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c), px = (int)((i / c) % w), py = (int)((i / ((long)c * w)) % h);
        const long im = i / ((long)c * w * h);
        float sm = 0.f, sxx = 0.f, sxy = 0.f;
        const int dy0 = max(0, py - ho + 1), dy1 = min(WIN - 1, py);
        const int dx0 = max(0, px - wo + 1), dx1 = min(WIN - 1, px);
        for (int dy = dy0; dy <= dy1; ++dy) {
            const long row = ((im * ho + (py - dy)) * wo) * c + ch;
            for (int dx = dx0; dx <= dx1; ++dx) {
                const float wt = gs[dy * WIN + dx];
                const long o = row + (long)(px - dx) * c;
                sm = fmaf(wt, maps[o], sm);
                sxx = fmaf(wt, maps[plane + o], sxx);
                sxy = fmaf(wt, maps[2 * plane + o], sxy);
            }
        }
        const float cf = coef ? coef[im * c + ch] : 1.0f;          // MS-SSIM: the maps are unscaled, the weight is per (n, c)
        const float g = gscale * cf * (sm + 2.0f * y[i] * sxx + t[i] * sxy);
        grad[i] = accumulate ? grad[i] + g : g;
    }
}

// MS-SSIM building block (tf.image.ssim_multiscale -> _ssim_per_channel): per (image, channel) sums of the SSIM map
// (luminance x contrast-structure) and of the contrast-structure map alone, plus - optionally - the UNSCALED derivative maps of
// one of them w.r.t. the window moments (which = 1: SSIM, 2: cs).  Workgroups (n, ch, blk): every block stays in one plane.
constexpr int BPP = 8;           // workgroups per (image, channel) plane
__global__ __launch_bounds__(256) void ssim_planes_kernel(const float* __restrict__ y, const float* __restrict__ t,
                                                          double* __restrict__ partial, float* __restrict__ maps,
                                                          long plane, int h, int w, int c, float max_val,
                                                          const float* __restrict__ gk, int which) {
    __shared__ double red[8];
    __shared__ float gs[WIN * WIN];
    if (threadIdx.x < WIN * WIN) gs[threadIdx.x] = gk[threadIdx.x];
    __syncthreads();
    const int blk = blockIdx.x % BPP, ch = (blockIdx.x / BPP) % c, n = blockIdx.x / (BPP * c);
    const int ho = h - WIN + 1, wo = w - WIN + 1;
    const int items = ho * wo;
    const double c1 = (0.01 * max_val) * (0.01 * max_val), c2 = (0.03 * max_val) * (0.03 * max_val);
    double s_ssim = 0.0, s_cs = 0.0;
    for (int i = blk * 256 + threadIdx.x; i < items; i += BPP * 256) {
        const int x0 = i % wo, y0 = i / wo;
        double ey = 0, et = 0, eyy = 0, ett = 0, eyt = 0;
        for (int dy = 0; dy < WIN; ++dy) {
            const long row = (((long)n * h + y0 + dy) * w + x0) * c + ch;
#pragma unroll
            for (int dx = 0; dx < WIN; ++dx) {
                const double wt = (double)gs[dy * WIN + dx];
                const double vy = y[row + (long)dx * c], vt = t[row + (long)dx * c];
                ey += wt * vy; et += wt * vt; eyy += wt * vy * vy; ett += wt * vt * vt; eyt += wt * vy * vt;
            }
        }
        const double a1 = 2 * ey * et + c1, a2 = 2 * (eyt - ey * et) + c2;
        const double b1 = ey * ey + et * et + c1, b2 = (eyy - ey * ey) + (ett - et * et) + c2;
        const double cs = a2 / b2, ss = a1 * cs / b1;
        s_ssim += ss;
        s_cs += cs;
        if (maps && which) {
            const long o = (((long)n * ho + y0) * wo + x0) * c + ch;
            if (which == 1) {
                const double inv = 1.0 / (b1 * b2);
                maps[o] = (float)(2 * et * (a2 - a1) * inv - ss * 2 * ey * (b2 - b1) * inv);
                maps[plane + o] = (float)(-ss / b2);
                maps[2 * plane + o] = (float)(2 * a1 * inv);
            } else {
                maps[o] = (float)((-2 * et * b2 + 2 * ey * a2) / (b2 * b2));
                maps[plane + o] = (float)(-a2 / (b2 * b2));
                maps[2 * plane + o] = (float)(2.0 / b2);
            }
        }
    }
    s_ssim = wave_sum_d(s_ssim);
    s_cs = wave_sum_d(s_cs);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s_ssim; red[4 + (threadIdx.x >> 6)] = s_cs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = red[0] + red[1] + red[2] + red[3];
        partial[2 * blockIdx.x + 1] = red[4] + red[5] + red[6] + red[7];
    }
}

__global__ void ssim_planes_final_kernel(const double* __restrict__ partial, int planes, double inv_items,
                                         float* __restrict__ mean_ssim, float* __restrict__ mean_cs) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= planes) return;
    double a = 0.0, b = 0.0;
    for (int k = 0; k < BPP; ++k) { a += partial[2 * (p * BPP + k)]; b += partial[2 * (p * BPP + k) + 1]; }
    if (mean_ssim) mean_ssim[p] = (float)(a * inv_items);
    if (mean_cs) mean_cs[p] = (float)(b * inv_items);
}

// ms[n][c] = prod_k relu(v[k][n][c]) ^ w[k];  loss = 255 (1 - mean ms);  coef[k][n][c] = d loss / d v[k][n][c] / items[k]
__global__ __launch_bounds__(64) void msssim_combine_kernel(const float* __restrict__ v, const float* __restrict__ items,
                                                            int scales, int planes, float* __restrict__ loss,
                                                            float* __restrict__ coef) {
    const float wts[5] = {0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f};
    double acc = 0.0;
    for (int p = threadIdx.x; p < planes; p += 64) {
        double ms = 1.0;
        for (int k = 0; k < scales; ++k) ms *= pow((double)fmaxf(v[k * planes + p], 0.f), (double)wts[k]);
        acc += ms;
        if (coef)
            for (int k = 0; k < scales; ++k) {
                const float vk = v[k * planes + p];
                coef[k * planes + p] = vk > 0.f ? (float)(-255.0 / planes * wts[k] * ms / vk / items[k]) : 0.f;
            }
    }
    acc = wave_sum_d(acc);
    if (threadIdx.x == 0) loss[0] = (float)(255.0 * (1.0 - acc / planes));
}

}  // namespace

extern "C" {

size_t nimg_mae255_workspace_bytes(void) { return 2048 * sizeof(double); }

int nimg_mae255(const float* a, const float* b, float* loss, float* grad_a, long count, float grad_scale,
                int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (!a || !b || !loss || count <= 0 || !workspace) return NIMG_ERR_ARG;
    if (workspace_bytes < nimg_mae255_workspace_bytes()) return NIMG_ERR_WORKSPACE;
    const int grid = grid_for(count);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mae255_kernel, dim3(grid), dim3(256), 0, s, a, b, grad_a, (double*)workspace, count, grad_scale,
                       accumulate);
    NIMG_CHECK_LAUNCH();
    hipLaunchKernelGGL(mean_final_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, grid, 1.0 / (double)count,
                       1.0, 0.0, loss);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

size_t nimg_ssim_loss_workspace_bytes(int n, int h, int w, int c, int with_grad) {
    if (n <= 0 || h < WIN || w < WIN || c <= 0) return 0;
    const size_t partials = (size_t)n * BPI * sizeof(double);
    const size_t maps = with_grad ? (size_t)3 * n * (h - WIN + 1) * (w - WIN + 1) * c * sizeof(float) : 0;
    return partials + maps;
}

int nimg_ssim_loss(const float* y, const float* t, float* loss, float* grad_y, int n, int h, int w, int c,
                   float max_val, const float* gauss_win, float grad_scale, int accumulate, void* workspace,
                   size_t workspace_bytes, void* stream) {
    if (!y || !t || !loss || !gauss_win || !workspace || n <= 0 || c <= 0 || h < WIN || w < WIN || !(max_val > 0.f))
        return NIMG_ERR_ARG;
    if (workspace_bytes < nimg_ssim_loss_workspace_bytes(n, h, w, c, grad_y != nullptr)) return NIMG_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const long items = (long)(h - WIN + 1) * (w - WIN + 1) * c, plane = (long)n * items;
    double* partial = (double*)workspace;
    float* maps = grad_y ? (float*)(partial + (size_t)n * BPI) : nullptr;
    const double coef = -255.0 / ((double)n * (double)items);
    hipLaunchKernelGGL(ssim_loss_stats_kernel, dim3(n * BPI), dim3(256), 0, s, y, t, partial, maps, plane, h, w, c,
                       max_val, gauss_win, coef);
    NIMG_CHECK_LAUNCH();
    hipLaunchKernelGGL(mean_final_kernel, dim3(1), dim3(64), 0, s, (const double*)partial, n * BPI,
                       1.0 / ((double)n * (double)items), -255.0, 255.0, loss);
    NIMG_CHECK_LAUNCH();
    if (grad_y) {
        hipLaunchKernelGGL(ssim_loss_grad_kernel, dim3(grid_for((long)n * h * w * c)), dim3(256), 0, s, y, t, maps, plane,
                           grad_y, n, h, w, c, gauss_win, grad_scale, accumulate, (const float*)nullptr);
        NIMG_CHECK_LAUNCH();
    }
    return NIMG_OK;
}

size_t nimg_ssim_planes_workspace_bytes(int n, int c) { return (size_t)2 * n * c * BPP * sizeof(double); }

int nimg_ssim_planes(const float* y, const float* t, int n, int h, int w, int c, float max_val, const float* gauss_win,
                     float* mean_ssim, float* mean_cs, float* maps, int which_maps, void* workspace,
                     size_t workspace_bytes, void* stream) {
    if (!y || !t || !gauss_win || !workspace || n <= 0 || c <= 0 || h < WIN || w < WIN || !(max_val > 0.f) ||
        which_maps < 0 || which_maps > 2 || (which_maps && !maps) || (long)n * c * BPP > 0x7fffffffL)
        return NIMG_ERR_ARG;
    if (workspace_bytes < nimg_ssim_planes_workspace_bytes(n, c)) return NIMG_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const long plane = (long)n * (h - WIN + 1) * (w - WIN + 1) * c;
    hipLaunchKernelGGL(ssim_planes_kernel, dim3(n * c * BPP), dim3(256), 0, s, y, t, (double*)workspace, maps, plane, h, w,
                       c, max_val, gauss_win, which_maps);
    NIMG_CHECK_LAUNCH();
    const int planes = n * c;
    hipLaunchKernelGGL(ssim_planes_final_kernel, dim3((planes + 63) / 64), dim3(64), 0, s, (const double*)workspace, planes,
                       1.0 / ((double)(h - WIN + 1) * (w - WIN + 1)), mean_ssim, mean_cs);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_msssim_combine(const float* values, const float* items, int scales, int planes, float* loss, float* coef,
                        void* stream) {
    if (!values || !items || !loss || scales < 1 || scales > 5 || planes <= 0) return NIMG_ERR_ARG;
    hipLaunchKernelGGL(msssim_combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, values, items, scales, planes,
                       loss, coef);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_ssim_maps_grad(const float* y, const float* t, const float* maps, const float* coef, float* grad_y, int n,
                        int h, int w, int c, const float* gauss_win, float grad_scale, int accumulate, void* stream) {
    if (!y || !t || !maps || !grad_y || !gauss_win || n <= 0 || c <= 0 || h < WIN || w < WIN) return NIMG_ERR_ARG;
    const long plane = (long)n * (h - WIN + 1) * (w - WIN + 1) * c;
    hipLaunchKernelGGL(ssim_loss_grad_kernel, dim3(grid_for((long)n * h * w * c)), dim3(256), 0, (hipStream_t)stream, y,
                       t, maps, plane, grad_y, n, h, w, c, gauss_win, grad_scale, accumulate, coef);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

}  // extern "C"
