// Learned-codec specific kernels (TwitterDCN, models/compression.py:197-279):
//   DiscreteLatent  = trainable scale * z -> Quantization('soft-codebook') + differentiable entropy
//                     models/layers.py:139-170, 183-203 ; helpers/tf_helpers.py:290-333
//   plus the small element-wise pieces the codec graph needs (affine, LeakyReLU, zero insertion for strided dgrad).
//
// The reference evaluates the kernel weights in FLOAT64 (layers.py:141, tf_helpers.py:306); so do these kernels - the
// t-Student kernel (1 + (gamma d)^2 / v)^(-(v+1)/2) with gamma = 25, v = 50 underflows float32 for |d| > ~4.
// The entropy is a BATCH-GLOBAL statistic: forward accumulates the soft histogram in per-workgroup float64 partials, a
// finalise kernel reduces them in a fixed order (deterministic), computes H and dH/dhist; the backward kernel needs only
// those 2^bpf numbers.  Under data parallelism the partial histograms are what gets all-reduced.
#include <stdlib.h>

#include <type_traits>
#include "common.h"

namespace {

using namespace nimg;

constexpr int MAXK = 256;      // up to 8 bits per feature (models/compression.py:53); the generic kernels keep K-sized arrays per thread

inline int grid_for(long items) {
    long g = (items + 255) / 256;
    return (int)(g > 1024 ? 1024 : (g < 1 ? 1 : g));
}

__global__ void affine_kernel(const float* __restrict__ x, float* __restrict__ y, long count, float a, float b) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        y[i] = a * x[i] + b;
}

__global__ void lrelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long count, float alpha) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        y[i] = lrelu(x[i], alpha);
}

// tanh activation of INet's gamma MLP (models/pipelines.py:283) and its derivative through the stored output
__global__ void tanh_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long count) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        y[i] = tanhf(x[i]);
}
__global__ void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                long count) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        dx[i] = dy[i] * (1.0f - y[i] * y[i]);
}
// y = clip(x, 0, 1); the gradient is straight-through (pipelines.py:287, :341)
__global__ void clip01_kernel(const float* __restrict__ x, float* __restrict__ y, long count) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        y[i] = fminf(fmaxf(x[i], 0.0f), 1.0f);
}
// tf.pad(x, [[0,0],[P,P],[P,P],[0,0]], mode) with mode 0 CONSTANT(0) | 1 SYMMETRIC | 2 REFLECT: x (n,h,w,c) -> y (n,h+2P,w+2P,c)
__global__ void pad2d_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int h, int w, int c, int P,
                             int mode) {
    const int hp = h + 2 * P, wp = w + 2 * P;
    const long total = (long)n * hp * wp * c;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        long r = i / c;
        int gx = (int)(r % wp) - P;
        r /= wp;
        int gy = (int)(r % hp) - P;
        const long im = r / hp;
        const bool ok = map_coord(gy, h, mode) && map_coord(gx, w, mode);
        y[i] = ok ? x[((im * h + gy) * w + gx) * c + ch] : 0.f;
    }
}

// out (n,2h,2w,c): out[2y,2x] = in[y,x], zero elsewhere
__global__ void zero_insert2_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int h, int w, int c) {
    const long total = (long)n * 4 * h * w * c;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        long r = i / c;
        const int X = (int)(r % (2 * w));
        r /= 2 * w;
        const int Y = (int)(r % (2 * h));
        const long im = r / (2 * h);
        out[i] = ((X | Y) & 1) ? 0.f : in[((im * h + (Y >> 1)) * w + (X >> 1)) * c + ch];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// A 5x5 stride-2 TF-SAME convolution over an even-sized image IS a 3x3 stride-1 SAME convolution over its space-to-depth image
// (block (by, bx) = input pixels (2by + pr, 2bx + pc), channel (pr, pc, ci)): output row oy reads input rows 2oy - 1 .. 2oy + 3 =
// blocks oy - 1 (phase 1), oy (phases 0, 1), oy + 1 (phases 0, 1), i.e. tap ky = 2 dy + pr - 1 of block offset dy in 0..2
// (the combination dy = 0, pr = 0 does not exist: a zero weight); the zero padding of the block image is TF's (1, 2) padding.
// The strided layers of the codec (models/compression.py:217-229) run through the stride-1 MFMA kernels this way - the first one
// (3 input channels: 12 of 16 block channels) in forward, weight gradient and input gradient, the others in the input gradient,
// which otherwise is a stride-1 correlation over a zero-stuffed gradient (4x the products and a 4x larger tensor).
//   s2d2_affine_bf16:      y[by][bx][(2 pr + pc) c + ci] = bf16(a x[2by + pr][2bx + pc][ci] + b), channels >= 4c zero
//   s2d_conv_weights:      w3[dy][dx][(2 pr + pc) c + ci][co] = w5[2dy + pr - 1][2dx + pc - 1][ci][co] (0 outside 0..4)
//   s2d_conv_weights_bwd:  the gather back, dw5[ky][kx][ci][co] (+)= dw3[...]
//   d2s2_scale:            x[2by + pr][2bx + pc][ci] = scale * xs[by][bx][(2 pr + pc) c + ci]
__global__ void s2d2_affine_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, int n, int h, int w, int c, int cp,
                                        float a, float b) {
    const int hb = h >> 1, wb = w >> 1;
    const long total = (long)n * hb * wb * cp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i % cp);
        long r = i / cp;
        const int bx = (int)(r % wb);
        r /= wb;
        const int by = (int)(r % hb);
        const long im = r / hb;
        float v = 0.f;
        if (q < 4 * c) {
            const int ph = q / c, ci = q % c;
            v = fmaf(a, x[((im * h + 2 * by + (ph >> 1)) * w + 2 * bx + (ph & 1)) * c + ci], b);
        }
        y[i] = (__bf16)v;
    }
}

// c == 3, cp == 16 (the RGB image in front of the codec's first layer): one thread per BLOCK pixel - the two image rows it covers
// are 6 contiguous floats each (8-byte loads), its 16 bf16 block channels (12 used, 4 zero) two 16-byte stores.  The per-element
// form ran at 1.2 TB/s on its 64-bit divisions and 2-byte stores.
__global__ void s2d2_affine3_bf16_kernel(const float* __restrict__ x, uint4* __restrict__ y, int npix, int hb, int wb, float a,
                                         float b) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const int bx = i % wb, r = i / wb, by = r % hb, im = r / hb;
        const float2* top = reinterpret_cast<const float2*>(x + (((long)im * 2 * hb + 2 * by) * (2L * wb) + 2 * bx) * 3);
        const float2* bot = reinterpret_cast<const float2*>(x + (((long)im * 2 * hb + 2 * by + 1) * (2L * wb) + 2 * bx) * 3);
        const float2 t0 = top[0], t1 = top[1], t2 = top[2], b0 = bot[0], b1 = bot[1], b2 = bot[2];
        const float v[12] = {t0.x, t0.y, t1.x, t1.y, t2.x, t2.y, b0.x, b0.y, b1.x, b1.y, b2.x, b2.y};
        unsigned pk[8];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const __bf16 lo = (__bf16)fmaf(a, v[2 * k], b), hi = (__bf16)fmaf(a, v[2 * k + 1], b);
            pk[k] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
        }
        pk[6] = pk[7] = 0u;
        y[2L * i] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        y[2L * i + 1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
    }
}

__global__ void s2d_conv_weights_kernel(const float* __restrict__ w5, float* __restrict__ w3, int c, int cp, int cout) {
    const int total = 9 * cp * cout;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i % cout, q = (i / cout) % cp, tap = i / (cout * cp);
        float v = 0.f;
        if (q < 4 * c) {
            const int ph = q / c, ci = q % c;
            const int ky = 2 * (tap / 3) + (ph >> 1) - 1, kx = 2 * (tap % 3) + (ph & 1) - 1;
            if (ky >= 0 && kx >= 0) v = w5[((ky * 5 + kx) * c + ci) * cout + co];
        }
        w3[i] = v;
    }
}

__global__ void s2d_conv_weights_bwd_kernel(const float* __restrict__ dw3, float* __restrict__ dw5, int c, int cp, int cout,
                                            int accumulate) {
    const int total = 25 * c * cout;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i % cout, ci = (i / cout) % c, kx = (i / (cout * c)) % 5, ky = i / (cout * c * 5);
        const int dy = (ky + 1) >> 1, pr = (ky + 1) & 1, dx = (kx + 1) >> 1, pc = (kx + 1) & 1;
        const float v = dw3[(((dy * 3 + dx) * cp) + (2 * pr + pc) * c + ci) * cout + co];
        dw5[i] = accumulate ? dw5[i] + v : v;
    }
}

__global__ void d2s2_scale_kernel(const float* __restrict__ xs, float* __restrict__ x, int n, int h, int w, int c, int cp,
                                  float scale) {
    const int hb = h >> 1, wb = w >> 1;
    const long total = (long)n * h * w * c;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % c);
        long r = i / c;
        const int px = (int)(r % w);
        r /= w;
        const int py = (int)(r % h);
        const long im = r / h;
        x[i] = scale * xs[((im * hb + (py >> 1)) * wb + (px >> 1)) * cp + (2 * (py & 1) + (px & 1)) * c + ci];
    }
}

// c == 3 out of cp == 16 block channels (the input gradient of the codec's first layer, models/compression.py:217): one thread per
// BLOCK pixel - its 12 used floats are three of four 16-byte loads and land as two runs of 6 contiguous floats (rows 2y, 2y + 1):
// 8-byte stores.  The per-element form spends its time on 64-bit divisions and 4-byte accesses (0.7 TB/s: 211 us per config-5 step).
__global__ void d2s2_scale3_kernel(const float4* __restrict__ xs, float* __restrict__ x, int npix, int hb, int wb, float scale) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const int bx = i % wb, r = i / wb, by = r % hb, im = r / hb;
        const float4 a = xs[4L * i], b = xs[4L * i + 1], c = xs[4L * i + 2];
        float2* top = reinterpret_cast<float2*>(x + (((long)im * 2 * hb + 2 * by) * (2L * wb) + 2 * bx) * 3);
        float2* bot = reinterpret_cast<float2*>(x + (((long)im * 2 * hb + 2 * by + 1) * (2L * wb) + 2 * bx) * 3);
        top[0] = make_float2(scale * a.x, scale * a.y); top[1] = make_float2(scale * a.z, scale * a.w);
        top[2] = make_float2(scale * b.x, scale * b.y);
        bot[0] = make_float2(scale * b.z, scale * b.w); bot[1] = make_float2(scale * c.x, scale * c.y);
        bot[2] = make_float2(scale * c.z, scale * c.w);
    }
}

// The latent's rounding when it is NOT the soft codebook (reference models/layers.py:118-134, Quantization.call), selected by bits 2-3
// of the kernels' `soft_codebook` flag word: 0 identity; 1 'soft' = tf.round forward (half to even), the sinusoidal approximation's
// derivative backward; 2 'sin' = x - sin(2 pi x) / (2 pi) forward and backward.  float32 like the reference's graph (the Python
// constant 2 * np.pi becomes a float32 tensor constant next to a float32 operand).
__device__ __forceinline__ float round_mode_fwd(float zs, int mode) {
    const float tp = 6.2831855f;
    if (mode == 0) return zs;
    const float xs = zs - sinf(tp * zs) / tp;
    // 'soft': stop_gradient(round(x) - x_) + x_ with x_ the sinusoidal approximation (layers.py:126-128) - evaluated as written,
    // in float32: the sum can sit one ulp off round(x), exactly as the soft-codebook branch's (hard - soft) + soft below
    if (mode == 1) return (rintf(zs) - xs) + xs;
    return xs;
}
__device__ __forceinline__ double round_mode_bwd(float zs, int mode) {
    return mode ? (double)(1.0f - cosf(6.2831855f * zs)) : 1.0;
}

// KB = array bucket of the generic kernels (64 | 128 | 256 centres): a per-thread `double w[256]` is 6 KB of scratch, four times
// what the K <= 64 codebooks (bpf <= 6) need - the bucket is picked per launch (launch_generic below)
template <int KB>
struct KernelW {
    double w[KB];
    double dw[KB];
    double S, dS;
};

// base^(-m/2) for an integer m >= 1 (the t-Student kernel's exponent -(v + 1)/2 with the reference's integer v = 50,
// tf_helpers.py:290,321): binary powering + one square root + one division, ~15 float64 operations where the generic pow()
// spends ~300 (two calls per centre and latent value made these kernels 15 % of the codec's training step).  Every step is
// rounded to 0.5 ulp: <= 1e-15 relative against pow(); an overflowing power gives 0 like pow()'s underflow would.
__device__ __forceinline__ double pow_neg_half_int(double base, int m) {
    double r = 1.0, b = base;
    for (int n = m >> 1; n > 0; n >>= 1) {
        if (n & 1) r *= b;
        b *= b;
    }
    if (m & 1) r *= sqrt(base);
    return 1.0 / r;
}

// kernel weights (+eps) and their derivative w.r.t. u, for all K centres
template <int KB>
__device__ __forceinline__ void eval_weights(double u, const float* __restrict__ cb, int K, double v, double gamma,
                                             KernelW<KB>& o, bool need_grad) {
    o.S = 0.0;
    o.dS = 0.0;
    const double m_real = v + 1.0;
    const int m_int = (v > 0.0 && m_real <= 1024.0 && m_real == floor(m_real)) ? (int)m_real : 0;      // 0: generic pow
    for (int k = 0; k < K; ++k) {
        const double t = gamma * (u - (double)cb[k]);
        double wk, dwk;
        if (v <= 0.0) {                               // Gaussian kernel: exp(-gamma d^2)
            const double d = u - (double)cb[k];
            wk = exp(-gamma * d * d);
            dwk = wk * (-2.0 * gamma * d);
        } else {
            const double base = 1.0 + t * t / v;
            wk = m_int ? pow_neg_half_int(base, m_int) : pow(base, -(v + 1.0) / 2.0);
            dwk = wk * (-(v + 1.0) / v) * gamma * t / base;
        }
        o.w[k] = wk + 1e-72;
        o.dw[k] = need_grad ? dwk : 0.0;
        o.S += o.w[k];
        o.dS += o.dw[k];
    }
}

// forward: latent = STE(hard, soft)(scale * z); hist partial (float64) of the normalised weights AT THE LATENT values
template <int KB>
__global__ __launch_bounds__(256) void soft_codebook_fwd_kernel(const float* __restrict__ z,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ cb, int K, double v,
                                                                double gamma, float* __restrict__ latent,
                                                                double* __restrict__ hist_partial, long count,
                                                                int soft_codebook) {
    __shared__ double sh[MAXK];
    if (threadIdx.x < MAXK) sh[threadIdx.x] = 0.0;
    __syncthreads();
    const float s = scale ? scale[0] : 1.0f;
    double hacc[KB];
    for (int k = 0; k < K; ++k) hacc[k] = 0.0;
    KernelW<KB> kw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float zs = z[i] * s;                                           // layers.py:197-198 (float32 product)
        float lat = round_mode_fwd(zs, soft_codebook >> 2);
        if (soft_codebook & 1) {
            eval_weights((double)zs, cb, K, v, gamma, kw, false);
            double soft = 0.0, best = -1.0;
            int arg = 0;
            for (int k = 0; k < K; ++k) {
                const double wn = kw.w[k] / kw.S;
                soft += wn * (double)cb[k];
                if (wn > best) { best = wn; arg = k; }                       // first maximum, like tf.argmax
            }
            const float softf = (float)soft, hard = cb[arg];
            lat = (hard - softf) + softf;                                    // stop_gradient(hard - soft) + soft
        }
        latent[i] = lat;
        eval_weights((double)lat, cb, K, v, gamma, kw, false);               // entropy(latent, codebook), layers.py:201
        for (int k = 0; k < K; ++k) hacc[k] += kw.w[k] / kw.S;
    }
    for (int k = 0; k < K; ++k) {
        const double t = wave_sum_d(hacc[k]);
        if ((threadIdx.x & 63) == 0) atomicAdd(&sh[k], t);                   // LDS float64 atomics, 4 waves
    }
    __syncthreads();
    if (threadIdx.x < K) hist_partial[(long)blockIdx.x * K + threadIdx.x] = sh[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------------------------
// The same two kernels for the case every trained codec uses (t-Student kernel with an integer v + 1, tf_helpers.py:290,321),
// with the codebook size as a template parameter.  The generic kernels above index `double w[64]` / `hacc[64]` with run-time
// loops, which puts 1.5 KB per thread into scratch memory, divide by S once per centre, and take a square root and a division
// per weight; they cost 0.99 + 0.90 ms per training step of the codec at B = 50 (16 % of it).  Here
//   * base^-(m/2) = r^m with r = base^-1/2 from v_rsq_f32 refined by two Newton steps in float64 (error ~1e-16 before the
//     powering, <= 1e-14 after it) and the powering a run of wave-uniform multiplies - no division, no square root;
//   * everything the outputs need is a running sum (S, sum w c, arg max; in the backward sum dH dw, sum dH w, sum c dw, sum c w),
//     the only arrays - the weights at the latent value and the per-thread histogram - are fully unrolled registers;
//   * one reciprocal of S per latent value.
__device__ __forceinline__ double rsqrt_refined(double x) {
    double r = (double)__builtin_amdgcn_rsqf((float)x);       // 1 ulp of float32; x >= 1 here, +inf gives 0 (weight 0)
    r = r * __builtin_fma(-0.5 * x, r * r, 1.5);
    r = r * __builtin_fma(-0.5 * x, r * r, 1.5);
    return r;
}
__device__ __forceinline__ double pow_int(double r, int m) {  // r^m, m wave-uniform, m >= 1
    double acc = (m & 1) ? r : 1.0, b = r;
    for (int n = m >> 1; n > 0; n >>= 1) {
        b *= b;
        if (n & 1) acc *= b;
    }
    return acc;
}

// The same run of multiplies for a COMPILE-TIME exponent (M = 51: the reference's v = 50, tf_helpers.py:290): the loop above
// unrolls into 5 squarings + 3 products with no bit tests / selects - in the kernels below the runtime loop was ~48 of the ~85
// instructions per weight.  Same operations in the same order, so bit-identical to pow_int(r, M).  M = 0: runtime exponent.
template <int M>
__device__ __forceinline__ double pow_fixed(double r, int m) {
    if (M == 0) return pow_int(r, m);
    double acc = (M & 1) ? r : 1.0, b = r;
#pragma unroll
    for (int n = M >> 1; n > 0; n >>= 1) {
        b *= b;
        if (n & 1) acc *= b;
    }
    return acc;
}

// UNIT-SPACED codebooks (cb[k] = cb[0] + k, the reference's: consecutive integers, models/layers.py:160-170) under gamma >= 25,
// v = 50: seen from a point u inside [cb[0] - 1/2, cb[K-1] + 1/2] the weight of a centre 2.5 or more away is below
// (4.125 / 79.1)^25.5 = 2e-33 of the nearest centre's - thirteen orders under the resolution of the float64 sums it would be added
// to.  The WIN kernels evaluate the five centres around the nearest one and nothing else (the full loop for points outside
// that range): every float64 sum they form differs from the full one by less than its last bit, at 5 / K of the work.  The per-value
// histogram terms go to the workgroup's LDS histogram by float64 atomics (dynamic centre index; the full kernels keep
// K per-thread accumulators in registers).
template <int M>
__device__ __forceinline__ double t_weight(double u, double c, double gamma, double inv_v, int m, double& r_out, double& t_out) {
    const double t = gamma * (u - c);
    const double r = rsqrt_refined(__builtin_fma(t * t, inv_v, 1.0));
    r_out = r; t_out = t;
    return pow_fixed<M>(r, m);
}

template <int K, int M>
__global__ __launch_bounds__(256) void soft_codebook_fwd_win_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                                    const float* __restrict__ cb, int m, double inv_v,
                                                                    double gamma, float* __restrict__ latent,
                                                                    double* __restrict__ hist_partial, long count,
                                                                    int soft_codebook) {
    __shared__ double sh[K];
    if (threadIdx.x < K) sh[threadIdx.x] = 0.0;
    __syncthreads();
    const float s = scale ? scale[0] : 1.0f;
    const float c0f = cb[0];
    const double c0 = (double)c0f, lo = c0 - 0.5, hi = c0 + (double)(K - 1) + 0.5;
    // the centres [ka, kb] that matter at u (all of them outside the range the bound above covers)
    auto window = [&](double u, int& ka, int& kb) {
        if (u >= lo && u <= hi) {
            int k0 = (int)rint(u - c0);
            k0 = k0 < 0 ? 0 : (k0 > K - 1 ? K - 1 : k0);
            ka = k0 - 2 < 0 ? 0 : k0 - 2;
            kb = k0 + 2 > K - 1 ? K - 1 : k0 + 2;
        } else {
            ka = 0; kb = K - 1;
        }
    };
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float zs = z[i] * s;                                           // layers.py:197-198 (float32 product)
        float lat = round_mode_fwd(zs, soft_codebook >> 2);
        int ka, kb;
        double r, t;
        if (soft_codebook & 1) {
            double S = 0.0, wc = 0.0, best = -1.0;
            float hard = 0.f;
            window((double)zs, ka, kb);
            for (int k = ka; k <= kb; ++k) {
                const float ckf = c0f + (float)k;                            // exact: the centres are cb[0] + k
                const double w = t_weight<M>((double)zs, (double)ckf, gamma, inv_v, m, r, t) + 1e-72;
                S += w;
                wc = __builtin_fma(w, (double)ckf, wc);
                if (w > best) { best = w; hard = ckf; }                      // first maximum, like tf.argmax
            }
            const float softf = (float)(wc / S), hardf = hard;
            lat = (hardf - softf) + softf;                                   // stop_gradient(hard - soft) + soft
        }
        latent[i] = lat;
        window((double)lat, ka, kb);                                         // entropy(latent, codebook), layers.py:201
        double S = 0.0;
        for (int k = ka; k <= kb; ++k) S += t_weight<M>((double)lat, c0 + (double)k, gamma, inv_v, m, r, t) + 1e-72;
        const double inv = 1.0 / S;
        for (int k = ka; k <= kb; ++k)
            atomicAdd(&sh[k], (t_weight<M>((double)lat, c0 + (double)k, gamma, inv_v, m, r, t) + 1e-72) * inv);
    }
    __syncthreads();
    if (threadIdx.x < K) hist_partial[(long)blockIdx.x * K + threadIdx.x] = sh[threadIdx.x];
}

template <int K, int M>
__global__ __launch_bounds__(256) void soft_codebook_bwd_win_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                                    const float* __restrict__ latent,
                                                                    const float* __restrict__ dlat,
                                                                    const double* __restrict__ dH_dsum, float coef,
                                                                    const float* __restrict__ cb, int m, double inv_v,
                                                                    double gamma, float* __restrict__ dz,
                                                                    double* __restrict__ dscale_partial, long count,
                                                                    int soft_codebook) {
    __shared__ double red[4];
    __shared__ double sdh[K];
    if (threadIdx.x < K) sdh[threadIdx.x] = dH_dsum[threadIdx.x];
    __syncthreads();
    const float s = scale ? scale[0] : 1.0f;
    const double dfac = -(double)m * inv_v * gamma;
    const double c0 = (double)cb[0], lo = c0 - 0.5, hi = c0 + (double)(K - 1) + 0.5;
    auto sums = [&](double u, bool use_dh) {
        int ka = 0, kb = K - 1;
        if (u >= lo && u <= hi) {
            int k0 = (int)rint(u - c0);
            k0 = k0 < 0 ? 0 : (k0 > K - 1 ? K - 1 : k0);
            ka = k0 - 2 < 0 ? 0 : k0 - 2;
            kb = k0 + 2 > K - 1 ? K - 1 : k0 + 2;
        }
        double S = 0.0, dS = 0.0, A = 0.0, B = 0.0;
        for (int k = ka; k <= kb; ++k) {
            const double ck = c0 + (double)k;
            const double qk = use_dh ? sdh[k] : ck;
            double r, t;
            const double w0 = t_weight<M>(u, ck, gamma, inv_v, m, r, t);
            const double dw = w0 * (dfac * t) * (r * r);
            const double w = w0 + 1e-72;
            S += w; dS += dw;
            A = __builtin_fma(qk, dw, A);
            B = __builtin_fma(qk, w, B);
        }
        const double inv = 1.0 / S;
        return (A - B * dS * inv) * inv;
    };
    double dsum = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        double g = dlat ? (double)dlat[i] : 0.0;
        if (coef != 0.f) g += (double)coef * sums((double)latent[i], true);
        const float zs = z[i] * s;
        const double dsoft = (soft_codebook & 1) ? sums((double)zs, false) : round_mode_bwd(zs, soft_codebook >> 2);
        const double gz = g * dsoft;                 // gradient w.r.t. zs = scale * z
        dz[i] = (float)(gz * (double)s);
        dsum += gz * (double)z[i];
    }
    dsum = wave_sum_d(dsum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) dscale_partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

template <int K, int M>
__global__ __launch_bounds__(256) void soft_codebook_fwd_fast_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                                     const float* __restrict__ cb, int m, double inv_v,
                                                                     double gamma, float* __restrict__ latent,
                                                                     double* __restrict__ hist_partial, long count,
                                                                     int soft_codebook) {
    __shared__ double sh[K];
    if (threadIdx.x < K) sh[threadIdx.x] = 0.0;
    __syncthreads();
    const float s = scale ? scale[0] : 1.0f;
    double hacc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) hacc[k] = 0.0;
    auto weight = [&](double u, int k) {                                     // cb[k]: wave-uniform, read through the scalar cache
        const double t = gamma * (u - (double)cb[k]);
        return pow_fixed<M>(rsqrt_refined(__builtin_fma(t * t, inv_v, 1.0)), m) + 1e-72;
    };
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float zs = z[i] * s;                                           // layers.py:197-198 (float32 product)
        float lat = round_mode_fwd(zs, soft_codebook >> 2);
        if (soft_codebook & 1) {
            double S = 0.0, wc = 0.0, best = -1.0;
            float hard = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double w = weight((double)zs, k);
                S += w;
                wc = __builtin_fma(w, (double)cb[k], wc);
                if (w > best) { best = w; hard = cb[k]; }                    // first maximum, like tf.argmax
            }
            const float softf = (float)(wc / S), hardf = hard;
            lat = (hardf - softf) + softf;                                   // stop_gradient(hard - soft) + soft
        }
        latent[i] = lat;
        double w[K], S = 0.0;                                                // entropy(latent, codebook), layers.py:201
#pragma unroll
        for (int k = 0; k < K; ++k) { w[k] = weight((double)lat, k); S += w[k]; }
        const double inv = 1.0 / S;
#pragma unroll
        for (int k = 0; k < K; ++k) hacc[k] = __builtin_fma(w[k], inv, hacc[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double t = wave_sum_d(hacc[k]);
        if ((threadIdx.x & 63) == 0) atomicAdd(&sh[k], t);                   // LDS float64 atomics, 4 waves
    }
    __syncthreads();
    if (threadIdx.x < K) hist_partial[(long)blockIdx.x * K + threadIdx.x] = sh[threadIdx.x];
}

template <int K, int M>
__global__ __launch_bounds__(256) void soft_codebook_bwd_fast_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                                     const float* __restrict__ latent,
                                                                     const float* __restrict__ dlat,
                                                                     const double* __restrict__ dH_dsum, float coef,
                                                                     const float* __restrict__ cb, int m, double inv_v,
                                                                     double gamma, float* __restrict__ dz,
                                                                     double* __restrict__ dscale_partial, long count,
                                                                     int soft_codebook) {
    __shared__ double red[4];
    const float s = scale ? scale[0] : 1.0f;
    const double dfac = -(double)m * inv_v * gamma;                          // d w / d u = w * dfac * t / base = w * dfac * t * r^2
    // sums over the centres at u: S = sum w, dS = sum dw, A = sum q dw, B = sum q w  ->  sum q d(w / S)/du = A / S - B dS / S^2
    // with q = dH/dsum (entropy term) or q = the centres (soft value); both tables are wave-uniform (scalar cache)
    auto sums = [&](double u, auto use_dh) {
        double S = 0.0, dS = 0.0, A = 0.0, B = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double ck = (double)cb[k];
            const double qk = decltype(use_dh)::value ? dH_dsum[k] : ck;
            const double t = gamma * (u - ck);
            const double r = rsqrt_refined(__builtin_fma(t * t, inv_v, 1.0));
            const double w0 = pow_fixed<M>(r, m);
            const double dw = w0 * (dfac * t) * (r * r);
            const double w = w0 + 1e-72;
            S += w; dS += dw;
            A = __builtin_fma(qk, dw, A);
            B = __builtin_fma(qk, w, B);
        }
        const double inv = 1.0 / S;
        return (A - B * dS * inv) * inv;
    };
    double dsum = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        double g = dlat ? (double)dlat[i] : 0.0;
        if (coef != 0.f) g += (double)coef * sums((double)latent[i], std::true_type{});
        const float zs = z[i] * s;
        const double dsoft = (soft_codebook & 1) ? sums((double)zs, std::false_type{}) : round_mode_bwd(zs, soft_codebook >> 2);
        const double gz = g * dsoft;                 // gradient w.r.t. zs = scale * z
        dz[i] = (float)(gz * (double)s);
        dsum += gz * (double)z[i];
    }
    dsum = wave_sum_d(dsum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) dscale_partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// the fast kernels serve power-of-two codebooks of 8 .. 32 centres (3 .. 5 bits per feature; the reference trains 5) under the
// t-Student kernel with an integer v + 1; anything else runs the generic kernels.  NIMG_LATENT_GENERIC=1: always generic (A/B)
inline int fast_exponent(int K, double v) {
    static const bool generic = getenv("NIMG_LATENT_GENERIC") != nullptr;
    const double m = v + 1.0;
    if (generic || !(K == 8 || K == 16 || K == 32) || !(v > 0.0) || m > 1024.0 || m != floor(m)) return 0;
    return (int)m;
}

// hist_sum[k] = sum_blocks partial (fixed order)
// 1024 threads = 16 segments x up to 64 centres: segment g adds the blocks g, g + 16, ... in order, then the 16 segment sums
// are added in order - the same result whatever the machine does, and 64 loads in a chain instead of 1024
__global__ __launch_bounds__(1024) void hist_reduce_kernel(const double* __restrict__ partial, int nblocks, int K,
                                                           double* __restrict__ hist_sum) {
    __shared__ double seg[16][MAXK];
    const int k0 = threadIdx.x & 63, g = threadIdx.x >> 6;
    for (int k = k0; k < K; k += 64) {                 // a lane serves centres k0, k0 + 64, ... (K <= 256)
        double s = 0.0;
        for (int b = g; b < nblocks; b += 16) s += partial[(long)b * K + k];
        seg[g][k] = s;
    }
    __syncthreads();
    if (g == 0)
        for (int k = k0; k < K; k += 64) {
            double t = 0.0;
            for (int j = 0; j < 16; ++j) t += seg[j][k];
            hist_sum[k] = t;
        }
}

// entropy (bits) and dH/d(hist_sum) from the global weight sums; tf_helpers.py:326-331
__global__ void entropy_finalize_kernel(const double* __restrict__ hist_sum, int K, double n_total,
                                        float* __restrict__ entropy, double* __restrict__ dH_dsum) {
    // one wave, lane l = centres l, l + 64, l + 128, l + 192 (K <= MAXK = 256): per-lane partial sums in that order, then butterfly
    // sums over the lanes - a fixed order.  (One thread looping over the centres spent 40 us on its 3 K dependent float64
    // logarithms and divisions.)
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    const int l = threadIdx.x;
    constexpr int NPL = MAXK / 64;
    double hc[NPL], lq[NPL], q[NPL];
    bool on[NPL], clipped[NPL];
    double part = 0.0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const int k = l + 64 * j;
        on[j] = k < K;
        const double h = on[j] ? hist_sum[k] / n_total : 1.0;
        clipped[j] = h < 1e-9;
        hc[j] = on[j] ? (clipped[j] ? 1e-9 : h) : 0.0;
        part += hc[j];
    }
    const double T = wave_sum_d(part);
    double hp = 0.0, mp = 0.0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        q[j] = on[j] ? hc[j] / T : 1.0;
        lq[j] = log(q[j]);                                      // 0 on the idle slots
        hp += on[j] ? q[j] * lq[j] : 0.0;
        mp += on[j] ? q[j] * (lq[j] + 1.0) : 0.0;
    }
    const double H = -wave_sum_d(hp);
    if (l == 0) entropy[0] = (float)(H / 0.6931);
    // dH/dhc_k = -(log q_k + 1)/T + (sum_j q_j (log q_j + 1))/T ;  then through the clip and the 1/n_total mean
    const double mean = wave_sum_d(mp);
#pragma unroll
    for (int j = 0; j < NPL; ++j)
        if (on[j]) {
            const double d = (-(lq[j] + 1.0) + mean) / T / 0.6931;
            dH_dsum[l + 64 * j] = clipped[j] ? 0.0 : d / n_total;
        }
}

// backward: dz = scale * dsoft/du(zs) * [ dlat + coef * sum_k dH_dsum[k] * dwn_k/du(lat) ];  dscale partial = sum z * (...)
template <int KB>
__global__ __launch_bounds__(256) void soft_codebook_bwd_kernel(const float* __restrict__ z,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ latent,
                                                                const float* __restrict__ dlat,
                                                                const double* __restrict__ dH_dsum, float coef,
                                                                const float* __restrict__ cb, int K, double v,
                                                                double gamma, float* __restrict__ dz,
                                                                double* __restrict__ dscale_partial, long count,
                                                                int soft_codebook) {
    __shared__ double red[4];
    const float s = scale ? scale[0] : 1.0f;
    double dsum = 0.0;
    KernelW<KB> kw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        double g = dlat ? (double)dlat[i] : 0.0;
        if (coef != 0.f) {
            eval_weights((double)latent[i], cb, K, v, gamma, kw, true);
            double e = 0.0;
            for (int k = 0; k < K; ++k)
                e += dH_dsum[k] * (kw.dw[k] * kw.S - kw.w[k] * kw.dS) / (kw.S * kw.S);
            g += (double)coef * e;
        }
        const float zs = z[i] * s;
        double dsoft = round_mode_bwd(zs, soft_codebook >> 2);
        if (soft_codebook & 1) {
            eval_weights((double)zs, cb, K, v, gamma, kw, true);
            dsoft = 0.0;
            for (int k = 0; k < K; ++k)
                dsoft += (double)cb[k] * (kw.dw[k] * kw.S - kw.w[k] * kw.dS) / (kw.S * kw.S);
        }
        const double gz = g * dsoft;                 // gradient w.r.t. zs = scale * z
        dz[i] = (float)(gz * (double)s);
        dsum += gz * (double)z[i];
    }
    dsum = wave_sum_d(dsum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) dscale_partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void dscale_final_kernel(const double* __restrict__ partial, int nblocks, float* __restrict__ dscale,
                                    int accumulate) {
    __shared__ double seg[64];
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += partial[b];          // 64 interleaved chains, then a fixed-order finish
    seg[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double t = 0.0;
        for (int j = 0; j < 64; ++j) t += seg[j];
        dscale[0] = accumulate ? dscale[0] + (float)t : (float)t;
    }
}

// l2 loss: sum((a-b)^2)/2 and its gradient (a - b) * gscale (tf.nn.l2_loss, models/compression.py:92-93)
__global__ __launch_bounds__(256) void l2_loss_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* grad_b, double* __restrict__ partial, long count,
                                                      float gscale, int accumulate) {
    __shared__ double red[4];
    double s = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float d = b[i] - a[i];                                  // d/db of (a-b)^2/2 = (b - a)
        s += 0.5 * (double)d * (double)d;
        if (grad_b) grad_b[i] = accumulate ? grad_b[i] + gscale * d : gscale * d;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// out = sum of the per-block partial sums, in a fixed order: lane l of one wave adds partials l, l + 64, ... (independent
// loads), then a butterfly over the lanes.  (One thread adding 1024 partials through 1024 dependent loads took 56 us.)
__global__ void sum_final_kernel(const double* __restrict__ partial, int nblocks, float* out) {
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    double s = 0.0;
    for (int k = threadIdx.x; k < nblocks; k += 64) s += partial[k];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) out[0] = (float)s;
}

}  // namespace

extern "C" {

int nimg_affine(const float* x, float* y, long count, float a, float b, void* stream) {
    if (!x || !y || count < 0) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    hipLaunchKernelGGL(affine_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, x, y, count, a, b);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_lrelu_fwd(const float* x, float* y, long count, float alpha, void* stream) {
    if (!x || !y || count < 0) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    hipLaunchKernelGGL(lrelu_fwd_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, x, y, count, alpha);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_tanh_fwd(const float* x, float* y, long count, void* stream) {
    if (!x || !y || count < 0) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    hipLaunchKernelGGL(tanh_fwd_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, x, y, count);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_tanh_bwd(const float* dy, const float* y, float* dx, long count, void* stream) {
    if (!dy || !y || !dx || count < 0) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, count);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_clip01(const float* x, float* y, long count, void* stream) {
    if (!x || !y || count < 0) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    hipLaunchKernelGGL(clip01_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, x, y, count);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_pad2d(const float* x, float* y, int n, int h, int w, int c, int pad, int pad_mode, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!x || !y || n < 0 || h <= 0 || w <= 0 || c <= 0 || pad < 0 || pad_mode < 0 || pad_mode > 2) return NIMG_ERR_ARG;
    if ((pad_mode == 1 && (pad > h || pad > w)) || (pad_mode == 2 && (pad >= h || pad >= w))) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    hipLaunchKernelGGL(pad2d_kernel, dim3(grid_for((long)n * (h + 2 * pad) * (w + 2 * pad) * c)), dim3(256), 0,
                       (hipStream_t)stream, x, y, n, h, w, c, pad, pad_mode);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_s2d2_affine_bf16(const float* x, void* y, int n, int h, int w, int c, int cp, float a, float b, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!x || !y || n < 0 || h < 2 || w < 2 || (h & 1) || (w & 1) || c < 1 || cp < 4 * c) return NIMG_ERR_ARG;
    if (c == 3 && cp == 16 && (long)n * h * w < (1L << 31)) {
        const int npix = n * (h >> 1) * (w >> 1);
        hipLaunchKernelGGL(s2d2_affine3_bf16_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, x, (uint4*)y, npix,
                           h >> 1, w >> 1, a, b);
        NIMG_CHECK_LAUNCH();
        return NIMG_OK;
    }
    hipLaunchKernelGGL(s2d2_affine_bf16_kernel, dim3(grid_for((long)n * (h / 2) * (w / 2) * cp)), dim3(256), 0, (hipStream_t)stream,
                       x, (__bf16*)y, n, h, w, c, cp, a, b);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_s2d_conv_weights(const float* w5, float* w3, int cin, int cp, int cout, void* stream) {
    if (!w5 || !w3 || cin < 1 || cp < 4 * cin || cout < 1) return NIMG_ERR_ARG;
    hipLaunchKernelGGL(s2d_conv_weights_kernel, dim3(grid_for(9L * cp * cout)), dim3(256), 0, (hipStream_t)stream, w5, w3, cin, cp,
                       cout);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_s2d_conv_weights_bwd(const float* dw3, float* dw5, int cin, int cp, int cout, int accumulate, void* stream) {
    if (!dw3 || !dw5 || cin < 1 || cp < 4 * cin || cout < 1) return NIMG_ERR_ARG;
    hipLaunchKernelGGL(s2d_conv_weights_bwd_kernel, dim3(grid_for(25L * cin * cout)), dim3(256), 0, (hipStream_t)stream, dw3, dw5,
                       cin, cp, cout, accumulate);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_d2s2_scale(const float* xs, float* x, int n, int h, int w, int c, int cp, float scale, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!xs || !x || n < 0 || h < 2 || w < 2 || (h & 1) || (w & 1) || c < 1 || cp < 4 * c) return NIMG_ERR_ARG;
    if (c == 3 && cp == 16 && (long)n * h * w < (1L << 31)) {
        const int npix = n * (h >> 1) * (w >> 1);
        hipLaunchKernelGGL(d2s2_scale3_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, (const float4*)xs, x, npix,
                           h >> 1, w >> 1, scale);
        NIMG_CHECK_LAUNCH();
        return NIMG_OK;
    }
    hipLaunchKernelGGL(d2s2_scale_kernel, dim3(grid_for((long)n * h * w * c)), dim3(256), 0, (hipStream_t)stream, xs, x, n, h, w, c,
                       cp, scale);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_zero_insert2(const float* in, float* out, int n, int h, int w, int c, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!in || !out || n < 0 || h <= 0 || w <= 0 || c <= 0) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    hipLaunchKernelGGL(zero_insert2_kernel, dim3(grid_for((long)n * 4 * h * w * c)), dim3(256), 0,
                       (hipStream_t)stream, in, out, n, h, w, c);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

size_t nimg_latent_workspace_bytes(int codebook_size) {
    // [1024 blocks][K] hist partials + [K] hist sums + [K] dH/dsum + [1024] dscale partials, all float64
    return (size_t)(1024 * (size_t)codebook_size + 2 * (size_t)codebook_size + 1024) * sizeof(double);
}

/* diagnostic switch: NIMG_LATENT_GENERIC_POW=1 nudges v off the integers, which sends the kernels down the generic pow() path */
static double kernel_v(float v) {
    static const bool generic = getenv("NIMG_LATENT_GENERIC_POW") != nullptr;
    return generic ? (double)v + 1e-9 : (double)v;
}

int nimg_latent_fwd(const float* z, const float* scale, const float* codebook, int codebook_size, float v,
                    float gamma, int soft_codebook, float* latent, float* entropy, long count, long count_global,
                    void* workspace, size_t workspace_bytes, int finalize, void* stream) {
    if (!z || !codebook || !latent || !entropy || !workspace || count <= 0 || codebook_size < 2 || codebook_size > MAXK)
        return NIMG_ERR_ARG;
    if (workspace_bytes < nimg_latent_workspace_bytes(codebook_size)) return NIMG_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int K = codebook_size, grid = grid_for(count);
    double* part = (double*)workspace;
    double* hsum = part + 1024 * (size_t)K;
    double* dH = hsum + K;
    const double vd = kernel_v(v);
    const int m = fast_exponent(K, vd);
#define NIMG_SCB_FWD(KK, MM)                                                                                              \
    hipLaunchKernelGGL((soft_codebook_fwd_fast_kernel<KK, MM>), dim3(grid), dim3(256), 0, s, z, scale, codebook, m, 1.0 / vd, \
                       (double)gamma, latent, part, count, soft_codebook)
    // soft_codebook bit 1 (a promise of the caller): unit-spaced codebook -> the windowed kernels (gamma >= 25, v = 50 only)
    static const bool no_win = getenv("NIMG_LATENT_NO_WINDOW") != nullptr;
    const bool win = (soft_codebook & 2) && m == 51 && gamma >= 25.0f && !no_win;
    soft_codebook &= ~2;                       // bit 0: soft codebook; bits 2-3: rounding mode of the other branch
#define NIMG_SCB_FWD_WIN(KK)                                                                                              \
    hipLaunchKernelGGL((soft_codebook_fwd_win_kernel<KK, 51>), dim3(grid), dim3(256), 0, s, z, scale, codebook, m, 1.0 / vd, \
                       (double)gamma, latent, part, count, soft_codebook)
    if (win && K == 32) NIMG_SCB_FWD_WIN(32);
    else if (win && K == 16) NIMG_SCB_FWD_WIN(16);
    else if (win && K == 8) NIMG_SCB_FWD_WIN(8);
    else if (m == 51 && K == 32) NIMG_SCB_FWD(32, 51);
    else if (m == 51 && K == 16) NIMG_SCB_FWD(16, 51);
    else if (m == 51 && K == 8) NIMG_SCB_FWD(8, 51);
    else if (m && K == 32) NIMG_SCB_FWD(32, 0);
    else if (m && K == 16) NIMG_SCB_FWD(16, 0);
    else if (m && K == 8) NIMG_SCB_FWD(8, 0);
#undef NIMG_SCB_FWD
    else if (K <= 64)
        hipLaunchKernelGGL(soft_codebook_fwd_kernel<64>, dim3(grid), dim3(256), 0, s, z, scale, codebook, K, vd, (double)gamma,
                           latent, part, count, soft_codebook);
    else if (K <= 128)
        hipLaunchKernelGGL(soft_codebook_fwd_kernel<128>, dim3(grid), dim3(256), 0, s, z, scale, codebook, K, vd, (double)gamma,
                           latent, part, count, soft_codebook);
    else
        hipLaunchKernelGGL(soft_codebook_fwd_kernel<MAXK>, dim3(grid), dim3(256), 0, s, z, scale, codebook, K, vd, (double)gamma,
                           latent, part, count, soft_codebook);
    NIMG_CHECK_LAUNCH();
    hipLaunchKernelGGL(hist_reduce_kernel, dim3(1), dim3(1024), 0, s, (const double*)part, grid, K, hsum);
    NIMG_CHECK_LAUNCH();
    if (finalize) {       // single process: the local histogram is the global one
        hipLaunchKernelGGL(entropy_finalize_kernel, dim3(1), dim3(64), 0, s, (const double*)hsum, K,
                           (double)(count_global > 0 ? count_global : count), entropy, dH);
        NIMG_CHECK_LAUNCH();
    }
    return NIMG_OK;
}

/* after an all-reduce(sum) of the K float64 histogram sums (workspace + 1024*K doubles) under data parallelism */
int nimg_latent_entropy_finalize(int codebook_size, long count_global, float* entropy, void* workspace, void* stream) {
    if (!entropy || !workspace || codebook_size < 2 || codebook_size > MAXK || count_global <= 0) return NIMG_ERR_ARG;
    const int K = codebook_size;
    double* hsum = (double*)workspace + 1024 * (size_t)K;
    hipLaunchKernelGGL(entropy_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)hsum, K,
                       (double)count_global, entropy, hsum + K);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_latent_bwd(const float* z, const float* scale, const float* latent, const float* dlatent,
                    float entropy_coef, const float* codebook, int codebook_size, float v, float gamma,
                    int soft_codebook, float* dz, float* dscale, int accumulate_dscale, long count, void* workspace,
                    size_t workspace_bytes, void* stream) {
    if (!z || !latent || !codebook || !dz || !workspace || count <= 0 || codebook_size < 2 || codebook_size > MAXK)
        return NIMG_ERR_ARG;
    if (workspace_bytes < nimg_latent_workspace_bytes(codebook_size)) return NIMG_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int K = codebook_size, grid = grid_for(count);
    double* part = (double*)workspace;
    double* dH = part + 1024 * (size_t)K + K;
    double* dsp = dH + K;
    const double vd = kernel_v(v);
    const int m = fast_exponent(K, vd);
#define NIMG_SCB_BWD(KK, MM)                                                                                          \
    hipLaunchKernelGGL((soft_codebook_bwd_fast_kernel<KK, MM>), dim3(grid), dim3(256), 0, s, z, scale, latent, dlatent,   \
                       (const double*)dH, entropy_coef, codebook, m, 1.0 / vd, (double)gamma, dz, dsp, count, soft_codebook)
    static const bool no_win = getenv("NIMG_LATENT_NO_WINDOW") != nullptr;
    const bool win = (soft_codebook & 2) && m == 51 && gamma >= 25.0f && !no_win;
    soft_codebook &= ~2;                       // bit 0: soft codebook; bits 2-3: rounding mode of the other branch
#define NIMG_SCB_BWD_WIN(KK)                                                                                          \
    hipLaunchKernelGGL((soft_codebook_bwd_win_kernel<KK, 51>), dim3(grid), dim3(256), 0, s, z, scale, latent, dlatent,    \
                       (const double*)dH, entropy_coef, codebook, m, 1.0 / vd, (double)gamma, dz, dsp, count, soft_codebook)
    if (win && K == 32) NIMG_SCB_BWD_WIN(32);
    else if (win && K == 16) NIMG_SCB_BWD_WIN(16);
    else if (win && K == 8) NIMG_SCB_BWD_WIN(8);
    else if (m == 51 && K == 32) NIMG_SCB_BWD(32, 51);
    else if (m == 51 && K == 16) NIMG_SCB_BWD(16, 51);
    else if (m == 51 && K == 8) NIMG_SCB_BWD(8, 51);
    else if (m && K == 32) NIMG_SCB_BWD(32, 0);
    else if (m && K == 16) NIMG_SCB_BWD(16, 0);
    else if (m && K == 8) NIMG_SCB_BWD(8, 0);
#undef NIMG_SCB_BWD
    else if (K <= 64)
        hipLaunchKernelGGL(soft_codebook_bwd_kernel<64>, dim3(grid), dim3(256), 0, s, z, scale, latent, dlatent,
                           (const double*)dH, entropy_coef, codebook, K, vd, (double)gamma, dz, dsp, count, soft_codebook);
    else if (K <= 128)
        hipLaunchKernelGGL(soft_codebook_bwd_kernel<128>, dim3(grid), dim3(256), 0, s, z, scale, latent, dlatent,
                           (const double*)dH, entropy_coef, codebook, K, vd, (double)gamma, dz, dsp, count, soft_codebook);
    else
        hipLaunchKernelGGL(soft_codebook_bwd_kernel<MAXK>, dim3(grid), dim3(256), 0, s, z, scale, latent, dlatent,
                           (const double*)dH, entropy_coef, codebook, K, vd, (double)gamma, dz, dsp, count, soft_codebook);
    NIMG_CHECK_LAUNCH();
    if (dscale) {
        hipLaunchKernelGGL(dscale_final_kernel, dim3(1), dim3(64), 0, s, (const double*)dsp, grid, dscale,
                           accumulate_dscale);
        NIMG_CHECK_LAUNCH();
    }
    return NIMG_OK;
}

size_t nimg_l2_loss_workspace_bytes(void) { return 1024 * sizeof(double); }

int nimg_l2_loss(const float* target, const float* y, float* loss, float* grad_y, long count, float grad_scale,
                 int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (!target || !y || !loss || !workspace || count <= 0) return NIMG_ERR_ARG;
    if (workspace_bytes < nimg_l2_loss_workspace_bytes()) return NIMG_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for(count);
    hipLaunchKernelGGL(l2_loss_kernel, dim3(grid), dim3(256), 0, s, target, y, grad_y, (double*)workspace, count,
                       grad_scale, accumulate);
    NIMG_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, grid, loss);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

}  // extern "C"
