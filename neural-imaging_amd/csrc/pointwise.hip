// HBM-bound element-wise / pooling / reduction kernels of the imaging channel (gfx950).
// float4 along the NHWC channel axis wherever C % 4 == 0, grid-stride loops capped at 2048 workgroups.
//
//   max-pool 2x2 fwd / bwd(+skip add, +LeakyReLU')        models/pipelines.py:197, models/forensics.py:70
//   Conv2DTranspose 2x2 s2 fwd                            models/pipelines.py:205
//   depth_to_space (DCR) + straight-through clip fwd/bwd  models/pipelines.py:218-223, models/compression.py:248-271
//   mse on 255-scaled images (loss + gradient)            helpers/tf_helpers.py:31-32
//   GAP + Dense + softmax + sparse CE head, fwd + bwd     models/forensics.py:80-94
//   Keras Adam over a flat parameter buffer               tf.keras.optimizers.Adam (pipelines.py:51 etc.)
#include <stdlib.h>
#include <mutex>

#include "common.h"

namespace {

using namespace nimg;

inline int grid_for(long items) {
    long g = (items + 255) / 256;
    return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

// ---------------------------------------------------------------------------------------------------------------
template <int V>
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int h, int w, int c) {
    const int ho = h / 2, wo = w / 2, cv = c / V;
    const long total = (long)n * ho * wo * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cv) * V;
        long r = i / cv;
        const int ox = (int)(r % wo);
        r /= wo;
        const int oy = (int)(r % ho), im = (int)(r / ho);
        const float* p00 = x + (((long)im * h + 2 * oy) * w + 2 * ox) * c + cc;
        const float* p10 = p00 + (long)w * c;
        float* q = y + (((long)im * ho + oy) * wo + ox) * c + cc;
        if (V == 4) {
            const float4 a = *reinterpret_cast<const float4*>(p00), b = *reinterpret_cast<const float4*>(p00 + c);
            const float4 d = *reinterpret_cast<const float4*>(p10), e = *reinterpret_cast<const float4*>(p10 + c);
            *reinterpret_cast<float4*>(q) = make_float4(fmaxf(fmaxf(a.x, b.x), fmaxf(d.x, e.x)), fmaxf(fmaxf(a.y, b.y), fmaxf(d.y, e.y)),
                                                        fmaxf(fmaxf(a.z, b.z), fmaxf(d.z, e.z)), fmaxf(fmaxf(a.w, b.w), fmaxf(d.w, e.w)));
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) q[k] = fmaxf(fmaxf(p00[k], p00[c + k]), fmaxf(p10[k], p10[c + k]));
        }
    }
}

// dz[n,y,x,c] = (first arg-max of the window ? dp : 0) [+ add] ) * [lrelu'(y)]
// Backward of the fused conv -> LeakyReLU -> MaxPool2D epilogue (nimg_conv2d_pool_fwd): the full-resolution activation
// was never stored, so the pre-activation gradient is rebuilt from the pooled tensor, its argmax byte and the
// upstream gradient: dz[window position] = (position == argmax) ? dp * lrelu'(pooled) : 0.  (The sign of the window
// maximum is the sign of the pooled value, which is all lrelu' needs.)
typedef __bf16 nimg_bf16x4 __attribute__((ext_vector_type(4)));

// bf16 in, bf16 out, no activation mask (what the FAN's backward runs in throughput mode): 8 channels per thread - one
// 16-byte load of the pooled gradient, 8 arg-max bytes, four 16-byte stores - and the routing is done on the packed bf16
// pairs with byte-wise equality masks (no float conversions).
__device__ __forceinline__ unsigned unpool_eq_bytes(unsigned k, unsigned pos) {      // 0xFF in every byte of k equal to pos (0..3)
    const unsigned x = k ^ (pos * 0x01010101u);
    return (((x | (x >> 1)) & 0x01010101u) ^ 0x01010101u) * 0xFFu;
}

__global__ void maxpool2_unpool_bf16x8_kernel(const uint4* __restrict__ dp, const uint2* __restrict__ idx, uint4* __restrict__ dz,
                                              int n, int ho, int wo, int c8) {
    const long total = (long)n * ho * wo * c8;
    const long row = (long)2 * wo * c8;                    // uint4 entries per full-resolution row
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c8);
        long r = i / c8;
        const int ox = (int)(r % wo);
        r /= wo;                                            // r = im * ho + oy
        const uint4 g = dp[i];
        const uint2 k = idx[i];
        const long base = (2 * r * (long)(2 * wo) + 2 * ox) * c8 + cc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned m0 = unpool_eq_bytes(k.x, q), m1 = unpool_eq_bytes(k.y, q);
            uint4 o;
            o.x = g.x & __builtin_amdgcn_perm(m0, m0, 0x01010000u);
            o.y = g.y & __builtin_amdgcn_perm(m0, m0, 0x03030202u);
            o.z = g.z & __builtin_amdgcn_perm(m1, m1, 0x01010000u);
            o.w = g.w & __builtin_amdgcn_perm(m1, m1, 0x03030202u);
            dz[base + (q >> 1) * row + (q & 1) * c8] = o;
        }
    }
}

template <bool IN_BF16, bool OUT_BF16>
__global__ void maxpool2_unpool_kernel(const float* __restrict__ dp, const unsigned char* __restrict__ idx,
                                       const float* __restrict__ pooled, float* __restrict__ dz, int n, int ho, int wo,
                                       int c, int apply_mask, float alpha) {
    const int cv = c / 4, w = 2 * wo;
    const long total = (long)n * ho * wo * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cv) * 4;
        long r = i / cv;
        const int ox = (int)(r % wo);
        r /= wo;
        const int oy = (int)(r % ho), im = (int)(r / ho);
        const long po = (((long)im * ho + oy) * wo + ox) * c + cc;
        float4 g;
        if (IN_BF16) {
            const nimg_bf16x4 gb = *reinterpret_cast<const nimg_bf16x4*>(reinterpret_cast<const __bf16*>(dp) + po);
            g = make_float4((float)gb[0], (float)gb[1], (float)gb[2], (float)gb[3]);
        } else {
            g = *reinterpret_cast<const float4*>(dp + po);
        }
        const uchar4 k = *reinterpret_cast<const uchar4*>(idx + po);
        if (apply_mask) {
            const float4 pv = *reinterpret_cast<const float4*>(pooled + po);
            g.x *= pv.x > 0.f ? 1.0f : alpha; g.y *= pv.y > 0.f ? 1.0f : alpha;
            g.z *= pv.z > 0.f ? 1.0f : alpha; g.w *= pv.w > 0.f ? 1.0f : alpha;
        }
        const long base = (((long)im * 2 * ho + 2 * oy) * w + 2 * ox) * c + cc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 o = make_float4(k.x == q ? g.x : 0.f, k.y == q ? g.y : 0.f, k.z == q ? g.z : 0.f,
                                         k.w == q ? g.w : 0.f);
            const long off = base + (long)(q >> 1) * w * c + (long)(q & 1) * c;
            if (OUT_BF16) {
                nimg_bf16x4 ob;
                ob[0] = (__bf16)o.x; ob[1] = (__bf16)o.y; ob[2] = (__bf16)o.z; ob[3] = (__bf16)o.w;
                *reinterpret_cast<nimg_bf16x4*>(reinterpret_cast<__bf16*>(dz) + off) = ob;
            } else {
                *reinterpret_cast<float4*>(dz + off) = o;
            }
        }
    }
}

// bf16-stored activations (UNet in throughput mode): 8 channels = 16 bytes per lane.  Rounding to bf16 is monotonic, so
// the maximum of the rounded values is the rounded maximum - pooling bf16 tensors changes nothing for their bf16 consumers.
typedef __bf16 nimg_bf16x8 __attribute__((ext_vector_type(8)));
__global__ void maxpool2_fwd_bf16_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y, int n, int h, int w, int c) {
    const int ho = h / 2, wo = w / 2, cv = c / 8;
    const long total = (long)n * ho * wo * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cv) * 8;
        long r = i / cv;
        const int ox = (int)(r % wo);
        r /= wo;
        const int oy = (int)(r % ho), im = (int)(r / ho);
        const __bf16* p00 = x + (((long)im * h + 2 * oy) * w + 2 * ox) * c + cc;
        const __bf16* p10 = p00 + (long)w * c;
        const nimg_bf16x8 a = *reinterpret_cast<const nimg_bf16x8*>(p00), b = *reinterpret_cast<const nimg_bf16x8*>(p00 + c);
        const nimg_bf16x8 d = *reinterpret_cast<const nimg_bf16x8*>(p10), e = *reinterpret_cast<const nimg_bf16x8*>(p10 + c);
        nimg_bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            o[k] = (__bf16)fmaxf(fmaxf((float)a[k], (float)b[k]), fmaxf((float)d[k], (float)e[k]));
        *reinterpret_cast<nimg_bf16x8*>(y + (((long)im * ho + oy) * wo + ox) * c + cc) = o;
    }
}

// maxpool2_bwd_kernel on bf16-stored tensors (dp, yact, add, dz all bf16): same first-maximum routing, skip-gradient add
// and LeakyReLU' factor, float32 arithmetic per element, one rounding at the store.
__global__ void maxpool2_bwd_bf16_kernel(const __bf16* __restrict__ dp, const __bf16* __restrict__ yact, const __bf16* add,
                                         __bf16* dz, int n, int h, int w, int c, int apply_mask, float alpha) {
    const int ho = h / 2, wo = w / 2, cv = c / 8;
    const long total = (long)n * ho * wo * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cv) * 8;
        long r = i / cv;
        const int ox = (int)(r % wo);
        r /= wo;
        const int oy = (int)(r % ho), im = (int)(r / ho);
        const long base = (((long)im * h + 2 * oy) * w + 2 * ox) * c + cc;
        const long offs[4] = {0, (long)c, (long)w * c, (long)w * c + c};
        nimg_bf16x8 v[4], a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] = *reinterpret_cast<const nimg_bf16x8*>(yact + base + offs[q]);
            if (add) a[q] = *reinterpret_cast<const nimg_bf16x8*>(add + base + offs[q]);
        }
        const nimg_bf16x8 g = *reinterpret_cast<const nimg_bf16x8*>(dp + (((long)im * ho + oy) * wo + ox) * c + cc);
        nimg_bf16x8 o[4];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v0 = (float)v[0][k], v1 = (float)v[1][k], v2 = (float)v[2][k], v3 = (float)v[3][k];
            const float m = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
            const int sel = v0 == m ? 0 : (v1 == m ? 1 : (v2 == m ? 2 : 3));
            const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t = (q == sel) ? (float)g[k] : 0.f;
                if (add) t += (float)a[q][k];
                if (apply_mask) t *= (vv[q] > 0.f ? 1.0f : alpha);
                o[q][k] = (__bf16)t;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<nimg_bf16x8*>(dz + base + offs[q]) = o[q];
    }
}

// V channels per thread; V == 4 moves float4s (16 B per lane, coalesced along the NHWC channel axis)
template <int V>
__global__ void maxpool2_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ yact,
                                    const float* add, float* dz, int n, int h, int w, int c, int apply_mask,
                                    float alpha) {
    const int ho = h / 2, wo = w / 2, cv = c / V;
    const long total = (long)n * ho * wo * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cv) * V;
        long r = i / cv;
        const int ox = (int)(r % wo);
        r /= wo;
        const int oy = (int)(r % ho), im = (int)(r / ho);
        const long base = (((long)im * h + 2 * oy) * w + 2 * ox) * c + cc;
        const long offs[4] = {0, (long)c, (long)w * c, (long)w * c + c};
        float v[4][V], a[4][V], g[V], o[4][V];
        if (V == 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(yact + base + offs[q]);
                v[q][0] = t.x; v[q][1] = t.y; v[q][2] = t.z; v[q][3] = t.w;
                if (add) {
                    const float4 u = *reinterpret_cast<const float4*>(add + base + offs[q]);
                    a[q][0] = u.x; a[q][1] = u.y; a[q][2] = u.z; a[q][3] = u.w;
                }
            }
            const float4 t = *reinterpret_cast<const float4*>(dp + (((long)im * ho + oy) * wo + ox) * c + cc);
            g[0] = t.x; g[1] = t.y; g[2] = t.z; g[3] = t.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    v[q][k] = yact[base + offs[q] + k];
                    if (add) a[q][k] = add[base + offs[q] + k];
                }
#pragma unroll
            for (int k = 0; k < V; ++k) g[k] = dp[(((long)im * ho + oy) * wo + ox) * c + cc + k];
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float m = fmaxf(fmaxf(v[0][k], v[1][k]), fmaxf(v[2][k], v[3][k]));
            int sel = 3;
            if (v[0][k] == m) sel = 0; else if (v[1][k] == m) sel = 1; else if (v[2][k] == m) sel = 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t = (q == sel) ? g[k] : 0.f;
                if (add) t += a[q][k];
                if (apply_mask) t *= (v[q][k] > 0.f ? 1.0f : alpha);
                o[q][k] = t;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (V == 4) *reinterpret_cast<float4*>(dz + base + offs[q]) = make_float4(o[q][0], o[q][1], o[q][2], o[q][3]);
            else
#pragma unroll
                for (int k = 0; k < V; ++k) dz[base + offs[q] + k] = o[q][k];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Conv2DTranspose(k=2,s=2): out[n,2y+i,2x+j,co] = sum_ci in[n,y,x,ci] * w[i,j,co,ci] + b[co]
// One workgroup: 16 input pixels x 64 output channels x the 4 taps, Cin streamed through LDS in chunks of 32.
__global__ __launch_bounds__(256) void convt2x2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           long npix, int h, int wd, int cin, int cout) {
    __shared__ float sx[16][33];
    __shared__ float sw[4][64][33];
    const int tid = threadIdx.x;
    const int cot = (cout + 63) / 64;
    const int co0 = (blockIdx.x % cot) * 64;
    const long p0 = (long)(blockIdx.x / cot) * 16;
    const int co = tid & 63, pg = tid >> 6;          // thread: channel co, pixels pg*4..pg*4+3, all 4 taps
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    for (int c0 = 0; c0 < cin; c0 += 32) {
        __syncthreads();
        for (int it = tid; it < 16 * 32; it += 256) {
            const int pp = it / 32, k = it % 32;
            sx[pp][k] = (p0 + pp < npix && c0 + k < cin) ? x[(p0 + pp) * cin + c0 + k] : 0.f;
        }
        for (int it = tid; it < 4 * 64 * 32; it += 256) {
            const int k = it % 32, oc = (it / 32) % 64, t = it / (32 * 64);
            sw[t][oc][k] = (co0 + oc < cout && c0 + k < cin) ? w[((long)t * cout + co0 + oc) * cin + c0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < 32; ++k) {
            float xv[4], wv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) xv[a] = sx[pg * 4 + a][k];
#pragma unroll
            for (int t = 0; t < 4; ++t) wv[t] = sw[t][co][k];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[a][t] = fmaf(xv[a], wv[t], acc[a][t]);
        }
    }
    if (co0 + co >= cout) return;
    const float bv = bias ? bias[co0 + co] : 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const long pp = p0 + pg * 4 + a;
        if (pp >= npix) continue;
        const int xx = (int)(pp % wd);
        const long rest = pp / wd;
        const int yy = (int)(rest % h);
        const long im = rest / h;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            y[(((im * 2 * h) + 2 * yy + (t >> 1)) * (2L * wd) + 2 * xx + (t & 1)) * cout + co0 + co] = acc[a][t] + bv;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// depth_to_space(DCR, block 2) with optional affine + straight-through clip: y = clip(scale*d2s(x)+shift, 0, 1)
__global__ void d2s_clip_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int h, int w, int co,
                                    float scale, float shift, int clip) {
    const long total = (long)n * h * w * 4 * co;        // output elements
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % co);
        long r = i / co;
        const int X = (int)(r % (2 * w));
        r /= 2 * w;
        const int Y = (int)(r % (2 * h)), im = (int)(r / (2 * h));
        const float v = x[(((long)im * h + (Y >> 1)) * w + (X >> 1)) * (4 * co) + ((Y & 1) * 2 + (X & 1)) * co + c];
        float o = scale * v + shift;
        if (clip) o = fminf(fmaxf(o, 0.f), 1.f);
        y[i] = o;
    }
}
// co == 3 (every NIP of the reference ends in depth_to_space of 12 channels): one thread per INPUT pixel - its 12 floats are
// three 16-byte loads, and they land as two runs of 6 contiguous floats (output rows 2y and 2y + 1, pixels 2x and 2x + 1):
// 8-byte stores.  The per-element form above spends its time on 64-bit divisions and 4-byte accesses (2.5 TB/s).
__global__ void d2s_clip3_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long npix, int h, int w, float scale,
                                     float shift, int clip) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % w);
        const long r = i / w;
        const int yy = (int)(r % h);
        const long im = r / h;
        const float4* src = reinterpret_cast<const float4*>(x + i * 12);
        const float4 a = src[0], b = src[1], c = src[2];
        float v[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            float o = scale * v[k] + shift;
            if (clip) o = fminf(fmaxf(o, 0.f), 1.f);
            v[k] = o;
        }
        float2* top = reinterpret_cast<float2*>(y + ((im * 2 * h + 2 * yy) * (2L * w) + 2 * xx) * 3);
        float2* bot = reinterpret_cast<float2*>(y + ((im * 2 * h + 2 * yy + 1) * (2L * w) + 2 * xx) * 3);
        top[0] = make_float2(v[0], v[1]); top[1] = make_float2(v[2], v[3]); top[2] = make_float2(v[4], v[5]);
        bot[0] = make_float2(v[6], v[7]); bot[1] = make_float2(v[8], v[9]); bot[2] = make_float2(v[10], v[11]);
    }
}
__global__ void d2s_clip3_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long npix, int h, int w, float scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % w);
        const long r = i / w;
        const int yy = (int)(r % h);
        const long im = r / h;
        const float2* top = reinterpret_cast<const float2*>(dy + ((im * 2 * h + 2 * yy) * (2L * w) + 2 * xx) * 3);
        const float2* bot = reinterpret_cast<const float2*>(dy + ((im * 2 * h + 2 * yy + 1) * (2L * w) + 2 * xx) * 3);
        const float2 t0 = top[0], t1 = top[1], t2 = top[2], b0 = bot[0], b1 = bot[1], b2 = bot[2];
        float4* dst = reinterpret_cast<float4*>(dx + i * 12);
        dst[0] = make_float4(scale * t0.x, scale * t0.y, scale * t1.x, scale * t1.y);
        dst[1] = make_float4(scale * t2.x, scale * t2.y, scale * b0.x, scale * b0.y);
        dst[2] = make_float4(scale * b1.x, scale * b1.y, scale * b2.x, scale * b2.y);
    }
}
// co % 4 == 0 (the codec's 128- and 64-channel layers, models/compression.py:233,245): 16-byte granules never straddle a
// channel block, 32-bit index arithmetic - the per-element forms run at 1.5 TB/s on their 64-bit divisions.
// Forward: one thread per OUTPUT granule (stores coalesced; loads are runs of co floats).  Backward: per INPUT-gradient granule.
__global__ void d2s_clip4_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ y, unsigned total4, unsigned h, unsigned w,
                                     unsigned c4, float scale, float shift, int clip) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
        const unsigned c = i % c4;
        unsigned r = i / c4;
        const unsigned X = r % (2 * w);
        r /= 2 * w;
        const unsigned Y = r % (2 * h), im = r / (2 * h);
        const float4 v = x[((im * h + (Y >> 1)) * w + (X >> 1)) * (4 * c4) + ((Y & 1) * 2 + (X & 1)) * c4 + c];
        float4 o = make_float4(scale * v.x + shift, scale * v.y + shift, scale * v.z + shift, scale * v.w + shift);
        if (clip) {
            o.x = fminf(fmaxf(o.x, 0.f), 1.f); o.y = fminf(fmaxf(o.y, 0.f), 1.f);
            o.z = fminf(fmaxf(o.z, 0.f), 1.f); o.w = fminf(fmaxf(o.w, 0.f), 1.f);
        }
        y[i] = o;
    }
}
__global__ void d2s_clip4_bwd_kernel(const float4* __restrict__ dy, float4* __restrict__ dx, unsigned total4, unsigned h,
                                     unsigned w, unsigned c4, float scale) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
        const unsigned ch = i % (4 * c4);
        unsigned r = i / (4 * c4);
        const unsigned xx = r % w;
        r /= w;
        const unsigned yy = r % h, im = r / h;
        const unsigned blk = ch / c4, c = ch - blk * c4;
        const float4 v = dy[((im * 2 * h + 2 * yy + (blk >> 1)) * (2 * w) + 2 * xx + (blk & 1)) * c4 + c];
        dx[i] = make_float4(scale * v.x, scale * v.y, scale * v.z, scale * v.w);
    }
}
// gradient: straight-through (identity through the clip), dx = scale * space_to_depth(dy)
__global__ void d2s_clip_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int n, int h, int w,
                                    int co, float scale) {
    const long total = (long)n * h * w * 4 * co;        // input elements
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % (4 * co));
        long r = i / (4 * co);
        const int xx = (int)(r % w);
        r /= w;
        const int yy = (int)(r % h), im = (int)(r / h);
        const int blk = ch / co, c = ch % co;
        dx[i] = scale * dy[(((long)im * 2 * h + 2 * yy + (blk >> 1)) * (2L * w) + 2 * xx + (blk & 1)) * co + c];
    }
}

// dz = dy * lrelu'(y)
__global__ void lrelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ yact, float* dz, long count,
                                 float alpha) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        dz[i] = dy[i] * (yact[i] > 0.f ? 1.0f : alpha);
}

// out = a + b (residual adds; out may alias a)
__global__ void add_kernel(const float* a, const float* __restrict__ b, float* out, long count) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        out[i] = a[i] + b[i];
}

// out = a0 + a1 + ... (2..6 float32 tensors, 16-byte accesses; out may alias any input): the manipulation gradients of the
// channel are summed onto the native branch in ONE pass instead of one add launch each
__global__ void add_n_kernel(const float4* a0, const float4* a1, const float4* a2, const float4* a3, const float4* a4,
                             const float4* a5, float4* out, long count4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count4; i += (long)gridDim.x * blockDim.x) {
        float4 s = a0[i];
        const float4 b = a1[i];
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        if (a2) { const float4 c = a2[i]; s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w; }
        if (a3) { const float4 c = a3[i]; s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w; }
        if (a4) { const float4 c = a4[i]; s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w; }
        if (a5) { const float4 c = a5[i]; s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w; }
        out[i] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// mse on 255-scaled images: loss = mean((255a-255b)^2); grad_a (+)= gscale * 2*255^2*(a-b)/count
__global__ __launch_bounds__(256) void mse255_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     float* grad_a, double* __restrict__ partial, long count,
                                                     float gscale, int accumulate) {
    __shared__ double red[4];
    double s = 0.0;
    const float gk = gscale * 2.0f * 255.0f * 255.0f / (float)count;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i];
        const float e = 255.0f * d;
        s += (double)e * (double)e;
        if (grad_a) grad_a[i] = accumulate ? grad_a[i] + gk * d : gk * d;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void int_words_kernel(int* __restrict__ dst, const int* __restrict__ src, long n, int value, int mode) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = mode == 0 ? value : max(dst[i], src[i]);
}
__global__ void float_fill_kernel(float* __restrict__ dst, long n, float value) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = value;
}
// The head of the UNet's backward pass in the workflow, one pass instead of three (add_n -> mse255 with accumulate ->
// d2s_clip3_bwd): element (n, y, x, ch) of the result's depth_to_space image = parts[0] + parts[1] + ... + gk (a - b), the
// additions in that order with the same single rounding per step as the three kernels (the last one contracted to an fma, as
// mse255_kernel's accumulate form compiles) - written as (n, h, w, 12); the loss partials as mse255_kernel.
struct SumS2dParts { const float* p[6]; };
__global__ __launch_bounds__(256) void mse255_sum_s2d3_kernel(SumS2dParts parts, int n_parts, const float* __restrict__ a,
                                                              const float* __restrict__ b, float* __restrict__ dz,
                                                              double* __restrict__ partial, long npix, int h, int w, float gscale) {
    __shared__ double red[4];
    double s = 0.0;
    const long count = npix * 12;
    const float gk = gscale * 2.0f * 255.0f * 255.0f / (float)count;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % w);
        const long r = i / w;
        const int yy = (int)(r % h);
        const long im = r / h;
        const long top = ((im * 2 * h + 2 * yy) * (2L * w) + 2 * xx) * 3, bot = top + 2L * w * 3;
        float v[12];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const long o = half ? bot : top;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float2 acc = reinterpret_cast<const float2*>(parts.p[0] + o)[q];
#pragma unroll
                for (int k = 1; k < 6; ++k)
                    if (k < n_parts) {
                        const float2 t = reinterpret_cast<const float2*>(parts.p[k] + o)[q];
                        acc.x += t.x; acc.y += t.y;
                    }
                const float2 va = reinterpret_cast<const float2*>(a + o)[q], vb = reinterpret_cast<const float2*>(b + o)[q];
                const float d0 = va.x - vb.x, d1 = va.y - vb.y;
                const float e0 = 255.0f * d0, e1 = 255.0f * d1;
                s += (double)e0 * (double)e0;
                s += (double)e1 * (double)e1;
                v[half * 6 + 2 * q] = fmaf(gk, d0, acc.x);
                v[half * 6 + 2 * q + 1] = fmaf(gk, d1, acc.y);
            }
        }
        float4* dst = reinterpret_cast<float4*>(dz + i * 12);
        dst[0] = make_float4(v[0], v[1], v[2], v[3]);
        dst[1] = make_float4(v[4], v[5], v[6], v[7]);
        dst[2] = make_float4(v[8], v[9], v[10], v[11]);
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// Row form of the kernel above (w even, w <= 512): a workgroup takes one OUTPUT row = two input rows of 6 w floats.  The pixel-pair
// form reads 24-byte runs at 8 bytes per lane - every 64-byte line of the seven input tensors is requested three times by
// neighbouring lanes (counters: 1.04 GB fetched per launch for 0.4 GB of operands, 150 us at B = 64); here the two rows are read
// as linear 16-byte items, summed in the same order (parts[0] + parts[1] + ..., then the fma with gk (a - b)), turned around
// through LDS and written as linear 16-byte items of the (n, h, w, 12) row.  Same values, same bits; the loss partials are per
// workgroup as before.
__global__ __launch_bounds__(256) void mse255_sum_s2d3_rows_kernel(SumS2dParts parts, int n_parts, const float* __restrict__ a,
                                                                   const float* __restrict__ b, float* __restrict__ dz,
                                                                   double* __restrict__ partial, long nrows, int h, int w,
                                                                   float gscale) {
    __shared__ double red[4];
    __shared__ __attribute__((aligned(16))) float sh[2 * 6 * 512];            // [top | bottom][6 w]
    double s = 0.0;
    const long count = nrows * w * 12;
    const float gk = gscale * 2.0f * 255.0f * 255.0f / (float)count;
    const int rq = (6 * w) >> 2;                                               // float4 items of one input row (w even)
    const int tid = threadIdx.x;
    for (long row = blockIdx.x; row < nrows; row += gridDim.x) {
        const long im = row / h;
        const int yy = (int)(row - im * h);
        const long top = ((im * 2 * h + 2 * yy) * (2L * w)) * 3;              // float offset of the upper input row (the lower one follows)
        __syncthreads();
        for (int item = tid; item < 2 * rq; item += 256) {
            const long o = top + 4L * item;
            float4 acc = *reinterpret_cast<const float4*>(parts.p[0] + o);
#pragma unroll
            for (int k = 1; k < 6; ++k)
                if (k < n_parts) {
                    const float4 t = *reinterpret_cast<const float4*>(parts.p[k] + o);
                    acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
                }
            const float4 va = *reinterpret_cast<const float4*>(a + o), vb = *reinterpret_cast<const float4*>(b + o);
            const float d0 = va.x - vb.x, d1 = va.y - vb.y, d2 = va.z - vb.z, d3 = va.w - vb.w;
            const float e0 = 255.0f * d0, e1 = 255.0f * d1, e2 = 255.0f * d2, e3 = 255.0f * d3;
            s += (double)e0 * (double)e0;
            s += (double)e1 * (double)e1;
            s += (double)e2 * (double)e2;
            s += (double)e3 * (double)e3;
            *reinterpret_cast<float4*>(sh + 4 * item) = make_float4(fmaf(gk, d0, acc.x), fmaf(gk, d1, acc.y), fmaf(gk, d2, acc.z),
                                                                     fmaf(gk, d3, acc.w));
        }
        __syncthreads();
        // output pixel px = [top 6 px .. + 5 | bottom 6 px .. + 5]: item j = (px, third t) -> t 0: top 0..3; 1: top 4, 5 + bottom 0, 1;
        // 2: bottom 2..5
        const float* st = sh;
        const float* sb = sh + 6 * w;
        float4* dst = reinterpret_cast<float4*>(dz + row * (long)w * 12);
        for (int j = tid; j < 3 * w; j += 256) {
            const int px = j / 3, third = j - 3 * px;
            float f[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * third + e;                                   // channel of the (n, h, w, 12) pixel
                f[e] = k < 6 ? st[6 * px + k] : sb[6 * px + k - 6];
            }
            dst[j] = make_float4(f[0], f[1], f[2], f[3]);
        }
    }
    s = wave_sum_d(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void mse255_final_kernel(const double* __restrict__ partial, int nblocks, long count, float* loss) {
    double s = 0.0;                                      // one wave: strided partials, then a fixed-shape butterfly
    for (int k = threadIdx.x; k < nblocks; k += 64) s += partial[k];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) loss[0] = (float)(s / (double)count);
}

// ---------------------------------------------------------------------------------------------------------------
// FAN head.  gap[n][c] = mean_p act[n][p][c]
__global__ __launch_bounds__(256) void gap_fwd_kernel(const float* __restrict__ act, float* __restrict__ gap, int hw,
                                                      int c) {
    const int n = blockIdx.x, tid = threadIdx.x;
    const int cq = c >> 2;
    if ((c & 3) == 0 && cq <= 256 && 256 % cq == 0) {    // float4 columns x pixel phases, fixed-order LDS finish
        __shared__ float4 red4[256];
        const int parts = 256 / cq, col = tid % cq, part = tid / cq;
        const float4* src = reinterpret_cast<const float4*>(act + (long)n * hw * c) + col;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = part; q < hw; q += parts) {
            const float4 v = src[(long)q * cq];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        red4[tid] = a;
        __syncthreads();
        if (part == 0) {
            for (int k = 1; k < parts; ++k) {
                const float4 v = red4[k * cq + col];
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            const float inv = 1.0f / (float)hw;
            reinterpret_cast<float4*>(gap + (long)n * c)[col] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
        }
        return;
    }
    for (int ch = tid; ch < c; ch += blockDim.x) {
        float sacc = 0.f;
        const float* p = act + (long)n * hw * c + ch;
        for (int q = 0; q < hw; ++q) sacc += p[(long)q * c];
        gap[(long)n * c + ch] = sacc / (float)hw;
    }
}

// one wave per image: logits = gap W + b (lanes over the features, butterfly sums), softmax, Keras sparse CE on
// probabilities (clip 1e-7, renormalise), per-image loss and d loss / d logits (already scaled by loss_scale = 1 / batch)
__global__ __launch_bounds__(64) void dense_softmax_ce_kernel(const float* __restrict__ gap, const float* __restrict__ w,
                                        const float* __restrict__ b, const int* __restrict__ labels,
                                        float* __restrict__ probs, float* __restrict__ loss_per,
                                        float* __restrict__ dlogits, int n, int c, int k, float loss_scale) {
    const int im = blockIdx.x, lane = threadIdx.x;
    if (im >= n) return;
    float z[16], pr[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) z[j] = 0.f;
    for (int ch = lane; ch < c; ch += 64) {
        const float g = gap[(long)im * c + ch];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < k) z[j] = fmaf(g, w[ch * k + j], z[j]);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (j < k) z[j] = wave_sum(z[j]) + b[j];
    if (lane != 0) return;
    float m = z[0];
    for (int j = 1; j < k; ++j) m = fmaxf(m, z[j]);
    float s = 0.f;
    for (int j = 0; j < k; ++j) { pr[j] = expf(z[j] - m); s += pr[j]; }
    for (int j = 0; j < k; ++j) { pr[j] /= s; probs[(long)im * k + j] = pr[j]; }
    if (!labels) return;
    // keras backend sparse_categorical_crossentropy(from_logits=False), eager path: clip, log, softmax-CE of log p
    const float eps = 1e-7f;
    const int lab = labels[im];
    float pc[16], S = 0.f;
    for (int j = 0; j < k; ++j) { pc[j] = fminf(fmaxf(pr[j], eps), 1.0f - eps); S += pc[j]; }
    loss_per[im] = -logf(pc[lab]) + logf(S);
    // dL/dp_j = [p_j inside clip range] * (1/S - delta_jl / pc_l);   dL/dz = p * (g - sum_j g_j p_j)
    float g[16], dot = 0.f;
    for (int j = 0; j < k; ++j) {
        const bool inside = pr[j] >= eps && pr[j] <= 1.0f - eps;
        g[j] = inside ? (1.0f / S - (j == lab ? 1.0f / pc[lab] : 0.f)) : 0.f;
        dot += g[j] * pr[j];
    }
    for (int j = 0; j < k; ++j) dlogits[(long)im * k + j] = loss_scale * pr[j] * (g[j] - dot);
}

// The same head for 16 < k <= 256 classes (models/forensics.py:37 allows n_classes up to 256): the lanes of the image's wave own
// the classes (lane j: classes j, j + 64, ...), the softmax / clip / renormalise sums are wave reductions.
__global__ __launch_bounds__(64) void dense_softmax_ce_wide_kernel(const float* __restrict__ gap, const float* __restrict__ w,
                                        const float* __restrict__ b, const int* __restrict__ labels,
                                        float* __restrict__ probs, float* __restrict__ loss_per,
                                        float* __restrict__ dlogits, int n, int c, int k, float loss_scale) {
    const int im = blockIdx.x, lane = threadIdx.x;
    if (im >= n) return;
    float z[4], pr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) z[q] = 0.f;
    for (int ch = 0; ch < c; ++ch) {
        const float g = gap[(long)im * c + ch];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = lane + 64 * q;
            if (j < k) z[q] = fmaf(g, w[ch * k + j], z[q]);
        }
    }
    float m = -3.0e38f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = lane + 64 * q;
        if (j < k) { z[q] += b[j]; m = fmaxf(m, z[q]); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = lane + 64 * q;
        pr[q] = j < k ? expf(z[q] - m) : 0.f;
        s += pr[q];
    }
    s = wave_sum(s);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = lane + 64 * q;
        pr[q] /= s;
        if (j < k) probs[(long)im * k + j] = pr[q];
    }
    if (!labels) return;
    const float eps = 1e-7f;
    const int lab = labels[im];
    float pc[4], S = 0.f, pl = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = lane + 64 * q;
        pc[q] = j < k ? fminf(fmaxf(pr[q], eps), 1.0f - eps) : 0.f;
        S += pc[q];
        if (j == lab) pl = pc[q];
    }
    S = wave_sum(S);
    pl = wave_sum(pl);                         // exactly one lane holds the label's clipped probability
    if (lane == 0) loss_per[im] = -logf(pl) + logf(S);
    float g[4], dot = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = lane + 64 * q;
        const bool inside = j < k && pr[q] >= eps && pr[q] <= 1.0f - eps;
        g[q] = inside ? (1.0f / S - (j == lab ? 1.0f / pl : 0.f)) : 0.f;
        dot += g[q] * pr[q];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = lane + 64 * q;
        if (j < k) dlogits[(long)im * k + j] = loss_scale * pr[q] * (g[q] - dot);
    }
}

// ... and its parameter gradients: one wave per feature channel (block c: bias, block c + 1: the loss), lanes over the classes,
// the images summed in order
__global__ __launch_bounds__(64) void dense_bwd_params_wide_kernel(const float* __restrict__ gap,
                                        const float* __restrict__ dlogits, const float* __restrict__ loss_per,
                                        float* __restrict__ dw, float* __restrict__ db, float* __restrict__ loss, int n, int c,
                                        int k, float loss_scale) {
    const int ch = blockIdx.x, lane = threadIdx.x;
    if (ch == c + 1) {
        double sd = 0.0;
        for (int im = lane; im < n; im += 64) sd += (double)loss_per[im];
        sd = wave_sum_d(sd);
        if (lane == 0) loss[0] = (float)(sd * (double)loss_scale);
        return;
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int im = 0; im < n; ++im) {
        const float g = ch < c ? gap[(long)im * c + ch] : 1.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = lane + 64 * q;
            if (j < k) acc[q] = fmaf(g, dlogits[(long)im * k + j], acc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = lane + 64 * q;
        if (j < k) {
            if (ch < c) dw[ch * k + j] = acc[q];
            else db[j] = acc[q];
        }
    }
}

// dW[c][j] = sum_n gap[n][c] dlogits[n][j];  db[j] = sum_n dlogits[n][j];  loss = scale * sum loss_per
// one wave per feature channel (block c: bias gradient, block c + 1: the loss); lanes over the images
__global__ __launch_bounds__(64) void dense_bwd_params_kernel(const float* __restrict__ gap,
                                        const float* __restrict__ dlogits,
                                        const float* __restrict__ loss_per, float* __restrict__ dw,
                                        float* __restrict__ db, float* __restrict__ loss, int n, int c, int k,
                                        float loss_scale) {
    const int ch = blockIdx.x, lane = threadIdx.x;
    if (ch == c + 1) {
        double sd = 0.0;
        for (int im = lane; im < n; im += 64) sd += (double)loss_per[im];
        sd = wave_sum_d(sd);
        if (lane == 0) loss[0] = (float)(sd * (double)loss_scale);
        return;
    }
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    for (int im = lane; im < n; im += 64) {
        const float g = ch < c ? gap[(long)im * c + ch] : 1.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < k) acc[j] = fmaf(g, dlogits[(long)im * k + j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (j < k) {
            const float t = wave_sum(acc[j]);
            if (lane == 0) {
                if (ch < c) dw[ch * k + j] = t;
                else db[j] = t;
            }
        }
}

// d act[n][p][ch] = (sum_j dlogits[n][j] W[ch][j]) / hw * lrelu'(act).  Workgroup = one image: a thread owns 4 adjacent
// channels, computes their k-term dot products ONCE, then streams the image's positions with 16-byte loads / stores (the
// first version recomputed the dot product - and two 64-bit divisions - for every element: 12 % of the HBM rate).
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ w,
                                                      const float* __restrict__ act, float* __restrict__ dact,
                                                      int hw, int c, int k, float alpha, int parts) {
    const int n = blockIdx.x / parts, part = blockIdx.x % parts;
    const int c4 = c >> 2, tid = threadIdx.x;
    const float inv = 1.0f / (float)hw;
    if ((c & 3) == 0 && c4 <= 256 && 256 % c4 == 0) {
        const int cg = tid % c4, prow = tid / c4, ppi = 256 / c4;             // positions per iteration
        float g[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = 0.f;
            for (int j = 0; j < k; ++j) a = fmaf(dlogits[(long)n * k + j], w[(cg * 4 + e) * k + j], a);
            g[e] = a / (float)hw;
        }
        const float4* a4 = reinterpret_cast<const float4*>(act + (long)n * hw * c);
        float4* d4 = reinterpret_cast<float4*>(dact + (long)n * hw * c);
        for (int p0 = part * ppi + prow; p0 < hw; p0 += parts * ppi) {
            const float4 a = a4[(long)p0 * c4 + cg];
            d4[(long)p0 * c4 + cg] = make_float4(g[0] * (a.x > 0.f ? 1.0f : alpha), g[1] * (a.y > 0.f ? 1.0f : alpha),
                                                 g[2] * (a.z > 0.f ? 1.0f : alpha), g[3] * (a.w > 0.f ? 1.0f : alpha));
        }
        return;
    }
    (void)inv;
    for (int i = part * 256 + tid; i < hw * c; i += parts * 256) {           // general shapes: one element per thread
        const int ch = i % c;
        float g = 0.f;
        for (int j = 0; j < k; ++j) g = fmaf(dlogits[(long)n * k + j], w[ch * k + j], g);
        g /= (float)hw;
        const long o = (long)n * hw * c + i;
        dact[o] = g * (act[o] > 0.f ? 1.0f : alpha);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// SSIM (Wang et al. 2004), per image = mean over channels and VALID window positions.
//   mode 0: skimage.metrics.structural_similarity as helpers/metrics.py:9-25 calls it (7x7 uniform window, sample
//           covariance N/(N-1), K1 .01, K2 .03, interior crop == VALID positions);
//   mode 1: tf.image.ssim as models/compression.py:89 calls it (11x11 Gaussian sigma 1.5, population moments).
// Separable windows (uniform: 1/win x 1/win; tf's Gaussian: normalised outer product, its 1-D factor = the row sums of the
// 2-D table), moments in double.  A workgroup walks 16 x 16 tiles of window positions of one channel: the (16 + win - 1)^2 input
// patch of both images is staged in LDS, the horizontal pass writes the five moment rows (u_a, u_b, u_aa, u_bb, u_ab) to LDS
// in double, the vertical pass finishes one position per thread: 2 x 5 x win fused multiply-adds per position instead of
// 5 x win^2 products with 3 global loads each (it was one thread per position straight from global memory: 403 us for 16 images
// of 256 x 256 x 3, every training step of the codec returns the SSIM - models/compression.py:129).  Per-workgroup partial
// sums, fixed-order finish.
template <int WIN>
__global__ __launch_bounds__(256) void ssim_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           double* __restrict__ partial, int h, int w, int c, int mode,
                                                           float max_val, const float* __restrict__ gk, int blocks_per_image) {
    constexpr int T = 16, HT = T + WIN - 1;
    __shared__ float sa[HT * HT], sb[HT * HT];
    __shared__ double hm[5][HT * T];
    __shared__ double red[256];
    __shared__ double wt1[WIN];
    const int tid = threadIdx.x;
    const int n = blockIdx.x / blocks_per_image, blk = blockIdx.x % blocks_per_image;
    const int ho = h - WIN + 1, wo = w - WIN + 1;
    const int tiles_x = (wo + T - 1) / T, tiles_y = (ho + T - 1) / T, ntiles = tiles_y * tiles_x * c;
    const double c1 = (0.01 * max_val) * (0.01 * max_val), c2 = (0.03 * max_val) * (0.03 * max_val);
    const double np_ = (double)WIN * WIN, covn = mode == 0 ? np_ / (np_ - 1.0) : 1.0;
    if (tid < WIN) {
        double s = 0.0;
        if (mode == 0) s = 1.0 / WIN;
        else
            for (int j = 0; j < WIN; ++j) s += (double)gk[tid * WIN + j];
        wt1[tid] = s;
    }
    __syncthreads();
    double wt[WIN];
#pragma unroll
    for (int k = 0; k < WIN; ++k) wt[k] = wt1[k];
    double sum = 0.0;
    for (int t = blk; t < ntiles; t += blocks_per_image) {
        const int ch = t % c, tile = t / c, ty0 = (tile / tiles_x) * T, tx0 = (tile % tiles_x) * T;
        __syncthreads();                                       // the previous tile's moment rows are consumed
        for (int i = tid; i < HT * HT; i += 256) {
            const int gy = ty0 + i / HT, gx = tx0 + i % HT;
            const bool ok = gy < h && gx < w;
            const long o = (((long)n * h + (ok ? gy : 0)) * w + (ok ? gx : 0)) * c + ch;
            sa[i] = ok ? a[o] : 0.f;
            sb[i] = ok ? b[o] : 0.f;
        }
        __syncthreads();
        for (int i = tid; i < HT * T; i += 256) {
            const int r = i / T, cc = i % T;
            double ux = 0, uy = 0, uxx = 0, uyy = 0, uxy = 0;
#pragma unroll
            for (int dx = 0; dx < WIN; ++dx) {
                const double va = sa[r * HT + cc + dx], vb = sb[r * HT + cc + dx], wv = wt[dx];
                const double wa = wv * va, wb = wv * vb;
                ux += wa; uy += wb; uxx += wa * va; uyy += wb * vb; uxy += wa * vb;
            }
            hm[0][i] = ux; hm[1][i] = uy; hm[2][i] = uxx; hm[3][i] = uyy; hm[4][i] = uxy;
        }
        __syncthreads();
        const int oy = tid / T, ox = tid % T;
        if (ty0 + oy < ho && tx0 + ox < wo) {
            double ux = 0, uy = 0, uxx = 0, uyy = 0, uxy = 0;
#pragma unroll
            for (int dy = 0; dy < WIN; ++dy) {
                const int i = (oy + dy) * T + ox;
                const double wv = wt[dy];
                ux += wv * hm[0][i]; uy += wv * hm[1][i]; uxx += wv * hm[2][i]; uyy += wv * hm[3][i]; uxy += wv * hm[4][i];
            }
            const double vx = covn * (uxx - ux * ux), vy = covn * (uyy - uy * uy), vxy = covn * (uxy - ux * uy);
            sum += ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2));
        }
    }
    red[tid] = sum;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) partial[blockIdx.x] = red[0];
}

__global__ void ssim_final_kernel(const double* __restrict__ partial, float* __restrict__ out, int blocks_per_image,
                                  double inv_items) {
    const int n = blockIdx.x;
    double s = 0.0;
    for (int k = threadIdx.x; k < blocks_per_image; k += 64) s += partial[(long)n * blocks_per_image + k];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) out[n] = (float)(s * inv_items);
}

// ---------------------------------------------------------------------------------------------------------------
// Keras Adam: theta -= lr_t * m / (sqrt(v) + eps), lr_t = lr * sqrt(1-b2^t)/(1-b1^t); optional grad pre-scale
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long count, float lr_t, float b1, float b2, float eps,
                            float gscale, const int* __restrict__ skip_flag, const float* __restrict__ lr_t_dev) {
    if (skip_flag && skip_flag[0]) return;      // NaN gradients: leave the model untouched (reference raises first)
    if (lr_t_dev) lr_t = lr_t_dev[0];           // captured step: the bias-corrected rate of THIS replay lives on the device
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float gi = gscale * g[i];
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// flag[0] = 1 if any element is NaN (workflows/manipulation_classification.py:281-282, kept on device)
__global__ void nan_flag_kernel(const float* __restrict__ g, long count, int* flag) {
    bool bad = false;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        bad |= (g[i] != g[i]);
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

}  // namespace

extern "C" {

int nimg_maxpool2_fwd(const float* x, float* y, int n, int h, int w, int c, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!x || !y || n < 0 || h < 2 || w < 2 || c <= 0) return NIMG_ERR_ARG;      /* odd sizes: VALID (the last row / column is dropped) */
    if (n == 0) return NIMG_OK;
    hipStream_t s = (hipStream_t)stream;
    if (c % 4 == 0)
        hipLaunchKernelGGL(maxpool2_fwd_kernel<4>, dim3(grid_for((long)n * (h / 2) * (w / 2) * (c / 4))), dim3(256),
                           0, s, x, y, n, h, w, c);
    else
        hipLaunchKernelGGL(maxpool2_fwd_kernel<1>, dim3(grid_for((long)n * (h / 2) * (w / 2) * c)), dim3(256), 0, s,
                           x, y, n, h, w, c);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_maxpool2_bwd(const float* dp, const float* yact, const float* add, float* dz, int n, int h, int w, int c,
                      int apply_lrelu_mask, float alpha, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!dp || !yact || !dz || n < 0 || h < 2 || w < 2 || c <= 0) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    hipStream_t s = (hipStream_t)stream;
    if ((h & 1) || (w & 1)) {            /* VALID pooling of an odd size: the dropped last row / column gets no gradient */
        if (add && add != dz) return NIMG_ERR_ARG;
        if (!add && hipMemsetAsync(dz, 0, (size_t)n * h * w * c * sizeof(float), s) != hipSuccess) return NIMG_ERR_LAUNCH;
    }
    if (c % 4 == 0)
        hipLaunchKernelGGL(maxpool2_bwd_kernel<4>, dim3(grid_for((long)n * (h / 2) * (w / 2) * (c / 4))), dim3(256),
                           0, s, dp, yact, add, dz, n, h, w, c, apply_lrelu_mask, alpha);
    else
        hipLaunchKernelGGL(maxpool2_bwd_kernel<1>, dim3(grid_for((long)n * (h / 2) * (w / 2) * c)), dim3(256), 0, s,
                           dp, yact, add, dz, n, h, w, c, apply_lrelu_mask, alpha);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

/* MaxPool2D(2) forward / backward on bf16-stored NHWC tensors (even h, w; c % 8 == 0): see nimg_maxpool2_fwd / _bwd */
int nimg_maxpool2_fwd_bf16(const void* x, void* y, int n, int h, int w, int c, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!x || !y || n < 0 || h < 2 || w < 2 || c <= 0 || (h & 1) || (w & 1) || (c & 7)) return NIMG_ERR_ARG;
    hipLaunchKernelGGL(maxpool2_fwd_bf16_kernel, dim3(grid_for((long)n * (h / 2) * (w / 2) * (c / 8))), dim3(256), 0,
                       (hipStream_t)stream, (const __bf16*)x, (__bf16*)y, n, h, w, c);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_maxpool2_bwd_bf16(const void* dp, const void* yact, const void* add, void* dz, int n, int h, int w, int c,
                           int apply_lrelu_mask, float alpha, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!dp || !yact || !dz || n < 0 || h < 2 || w < 2 || c <= 0 || (h & 1) || (w & 1) || (c & 7)) return NIMG_ERR_ARG;
    hipLaunchKernelGGL(maxpool2_bwd_bf16_kernel, dim3(grid_for((long)n * (h / 2) * (w / 2) * (c / 8))), dim3(256), 0,
                       (hipStream_t)stream, (const __bf16*)dp, (const __bf16*)yact, (const __bf16*)add, (__bf16*)dz, n, h, w, c,
                       apply_lrelu_mask, alpha);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_maxpool2_unpool(const float* dp, const unsigned char* idx, const float* pooled, float* dz, int n, int ho, int wo,
                         int c, int apply_lrelu_mask, float alpha, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!dp || !idx || !dz || n < 0 || ho <= 0 || wo <= 0 || c <= 0 || (c & 3)) return NIMG_ERR_ARG;
    if (apply_lrelu_mask && !pooled) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    return nimg_maxpool2_unpool_ex(dp, idx, pooled, dz, n, ho, wo, c, apply_lrelu_mask, alpha, 0, stream);
}

int nimg_maxpool2_unpool_ex(const float* dp, const unsigned char* idx, const float* pooled, float* dz, int n, int ho,
                            int wo, int c, int apply_lrelu_mask, float alpha, int flags, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!dp || !idx || !dz || n < 0 || ho <= 0 || wo <= 0 || c <= 0 || (c & 3)) return NIMG_ERR_ARG;
    if (apply_lrelu_mask && !pooled) return NIMG_ERR_ARG;
    const dim3 grid(grid_for((long)n * ho * wo * (c / 4)));
    hipStream_t s = (hipStream_t)stream;
    if ((flags & NIMG_BF16_IN) && (flags & NIMG_BF16_OUT) && !apply_lrelu_mask && (c & 7) == 0) {
        hipLaunchKernelGGL(maxpool2_unpool_bf16x8_kernel, dim3(grid_for((long)n * ho * wo * (c / 8))), dim3(256), 0, s,
                           (const uint4*)dp, (const uint2*)idx, (uint4*)dz, n, ho, wo, c / 8);
        NIMG_CHECK_LAUNCH();
        return NIMG_OK;
    }
#define NIMG_UNPOOL(A_, B_) hipLaunchKernelGGL((maxpool2_unpool_kernel<A_, B_>), grid, dim3(256), 0, s, dp, idx, pooled, dz, \
                                               n, ho, wo, c, apply_lrelu_mask, alpha)
    if (flags & NIMG_BF16_IN) { if (flags & NIMG_BF16_OUT) NIMG_UNPOOL(true, true); else NIMG_UNPOOL(true, false); }
    else { if (flags & NIMG_BF16_OUT) NIMG_UNPOOL(false, true); else NIMG_UNPOOL(false, false); }
#undef NIMG_UNPOOL
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_convt2x2_fwd(const float* x, const float* w, const float* bias, float* y, int n, int h, int wd, int cin,
                      int cout, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!x || !w || !y || n < 0 || h <= 0 || wd <= 0 || cin <= 0 || cout <= 0) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    const long npix = (long)n * h * wd;
    const long blocks = ((npix + 15) / 16) * ((cout + 63) / 64);
    hipLaunchKernelGGL(convt2x2_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, w, bias, y,
                       npix, h, wd, cin, cout);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_d2s_clip_fwd(const float* x, float* y, int n, int h, int w, int cout, float scale, float shift, int clip,
                      void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!x || !y || n < 0 || h <= 0 || w <= 0 || cout <= 0) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    if (cout == 3) {
        hipLaunchKernelGGL(d2s_clip3_fwd_kernel, dim3(grid_for((long)n * h * w)), dim3(256), 0, (hipStream_t)stream, x, y,
                           (long)n * h * w, h, w, scale, shift, clip);
        NIMG_CHECK_LAUNCH();
        return NIMG_OK;
    }
    if ((cout & 3) == 0 && (long)n * h * w * cout < (1L << 32)) {
        hipLaunchKernelGGL(d2s_clip4_fwd_kernel, dim3(grid_for((long)n * h * w * cout)), dim3(256), 0, (hipStream_t)stream,
                           (const float4*)x, (float4*)y, (unsigned)((long)n * h * w * cout), (unsigned)h, (unsigned)w,
                           (unsigned)(cout >> 2), scale, shift, clip);
        NIMG_CHECK_LAUNCH();
        return NIMG_OK;
    }
    hipLaunchKernelGGL(d2s_clip_fwd_kernel, dim3(grid_for((long)n * h * w * 4 * cout)), dim3(256), 0,
                       (hipStream_t)stream, x, y, n, h, w, cout, scale, shift, clip);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_d2s_clip_bwd(const float* dy, float* dx, int n, int h, int w, int cout, float scale, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!dy || !dx || n < 0 || h <= 0 || w <= 0 || cout <= 0) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    if (cout == 3) {
        hipLaunchKernelGGL(d2s_clip3_bwd_kernel, dim3(grid_for((long)n * h * w)), dim3(256), 0, (hipStream_t)stream, dy, dx,
                           (long)n * h * w, h, w, scale);
        NIMG_CHECK_LAUNCH();
        return NIMG_OK;
    }
    if ((cout & 3) == 0 && (long)n * h * w * cout < (1L << 32)) {
        hipLaunchKernelGGL(d2s_clip4_bwd_kernel, dim3(grid_for((long)n * h * w * cout)), dim3(256), 0, (hipStream_t)stream,
                           (const float4*)dy, (float4*)dx, (unsigned)((long)n * h * w * cout), (unsigned)h, (unsigned)w,
                           (unsigned)(cout >> 2), scale);
        NIMG_CHECK_LAUNCH();
        return NIMG_OK;
    }
    hipLaunchKernelGGL(d2s_clip_bwd_kernel, dim3(grid_for((long)n * h * w * 4 * cout)), dim3(256), 0,
                       (hipStream_t)stream, dy, dx, n, h, w, cout, scale);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_lrelu_bwd(const float* dy, const float* yact, float* dz, long count, float alpha, void* stream) {
    if (!dy || !yact || !dz || count < 0) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, dy, yact, dz,
                       count, alpha);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_add(const float* a, const float* b, float* out, long count, void* stream) {
    if (!a || !b || !out || count < 0) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, a, b, out, count);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_add_n(const float* const* inputs, int n_inputs, float* out, long count, void* stream) {
    if (!inputs || !out || n_inputs < 2 || n_inputs > 6 || count < 0 || (count & 3)) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    const float4* a[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < n_inputs; ++i) {
        if (!inputs[i]) return NIMG_ERR_ARG;
        a[i] = reinterpret_cast<const float4*>(inputs[i]);
    }
    hipLaunchKernelGGL(add_n_kernel, dim3(grid_for(count / 4)), dim3(256), 0, (hipStream_t)stream, a[0], a[1], a[2], a[3], a[4],
                       a[5], reinterpret_cast<float4*>(out), count / 4);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

size_t nimg_mse255_workspace_bytes(void) { return 2048 * sizeof(double); }

int nimg_mse255(const float* a, const float* b, float* loss, float* grad_a, long count, float grad_scale,
                int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (!a || !b || !loss || count <= 0 || !workspace) return NIMG_ERR_ARG;
    if (workspace_bytes < nimg_mse255_workspace_bytes()) return NIMG_ERR_WORKSPACE;
    const int grid = grid_for(count);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mse255_kernel, dim3(grid), dim3(256), 0, s, a, b, grad_a, (double*)workspace, count,
                       grad_scale, accumulate);
    NIMG_CHECK_LAUNCH();
    hipLaunchKernelGGL(mse255_final_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, grid, count, loss);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_mse255_sum_s2d3(const float* const* parts, int n_parts, const float* y, const float* target, float* loss, float* dz,
                         int n, int h, int w, float grad_scale, void* workspace, size_t workspace_bytes, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!parts || n_parts < 1 || n_parts > 6 || !y || !target || !loss || !dz || !workspace || n < 0 || h <= 0 || w <= 0)
        return NIMG_ERR_ARG;
    if (workspace_bytes < nimg_mse255_workspace_bytes()) return NIMG_ERR_WORKSPACE;
    SumS2dParts sp;
    for (int i = 0; i < 6; ++i) {
        sp.p[i] = i < n_parts ? parts[i] : nullptr;
        if (i < n_parts && (!parts[i] || ((size_t)parts[i] & 7))) return NIMG_ERR_ARG;
    }
    if (((size_t)y & 7) || ((size_t)target & 7) || ((size_t)dz & 15)) return NIMG_ERR_ARG;
    const long npix = (long)n * h * w;
    int grid = grid_for(npix);
    hipStream_t s = (hipStream_t)stream;
    bool a16 = (((size_t)y | (size_t)target) & 15) == 0;
    for (int i = 0; i < n_parts; ++i) a16 = a16 && ((size_t)parts[i] & 15) == 0;
    static const bool no_rows = getenv("NIMG_NO_S2D3_ROWS") != nullptr;
    if (!no_rows && a16 && (w & 1) == 0 && w <= 512) {          // row form: linear 16-byte items on both sides
        const long nrows = (long)n * h;
        grid = (int)(nrows < 2048 ? nrows : 2048);
        hipLaunchKernelGGL(mse255_sum_s2d3_rows_kernel, dim3(grid), dim3(256), 0, s, sp, n_parts, y, target, dz, (double*)workspace,
                           nrows, h, w, grad_scale);
    } else {
        hipLaunchKernelGGL(mse255_sum_s2d3_kernel, dim3(grid), dim3(256), 0, s, sp, n_parts, y, target, dz, (double*)workspace, npix,
                           h, w, grad_scale);
    }
    NIMG_CHECK_LAUNCH();
    hipLaunchKernelGGL(mse255_final_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, grid, npix * 12, loss);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

size_t nimg_ssim_workspace_bytes(int n) { return (size_t)n * 64 * sizeof(double); }

int nimg_ssim(const float* a, const float* b, float* out, int n, int h, int w, int c, int mode, float max_val,
              const float* gauss_win, void* workspace, size_t workspace_bytes, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    const int win = mode == 0 ? 7 : 11;
    if (!a || !b || !out || !workspace || n < 0 || c <= 0 || h < win || w < win || mode < 0 || mode > 1) return NIMG_ERR_ARG;
    if (mode == 1 && !gauss_win) return NIMG_ERR_ARG;
    if (workspace_bytes < nimg_ssim_workspace_bytes(n)) return NIMG_ERR_WORKSPACE;
    if (n == 0) return NIMG_OK;
    hipStream_t s = (hipStream_t)stream;
    const int bpi = 64;
    if (mode == 0)
        hipLaunchKernelGGL(ssim_partial_kernel<7>, dim3(n * bpi), dim3(256), 0, s, a, b, (double*)workspace, h, w, c, mode, max_val,
                           gauss_win, bpi);
    else
        hipLaunchKernelGGL(ssim_partial_kernel<11>, dim3(n * bpi), dim3(256), 0, s, a, b, (double*)workspace, h, w, c, mode, max_val,
                           gauss_win, bpi);
    NIMG_CHECK_LAUNCH();
    const double items = (double)(h - win + 1) * (w - win + 1) * c;
    hipLaunchKernelGGL(ssim_final_kernel, dim3(n), dim3(64), 0, s, (const double*)workspace, out, bpi, 1.0 / items);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_fan_head_fwd(const float* act, const float* w, const float* b, const int* labels, float* gap, float* probs,
                      float* loss_per, float* dlogits, int n, int hw, int c, int k, float loss_scale, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!act || !w || !b || !gap || !probs || n < 0 || hw <= 0 || c <= 0 || k <= 0 || k > 256) return NIMG_ERR_ARG;
    if (labels && (!loss_per || !dlogits)) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gap_fwd_kernel, dim3(n), dim3(256), 0, s, act, gap, hw, c);
    NIMG_CHECK_LAUNCH();
    if (k > 16)
        hipLaunchKernelGGL(dense_softmax_ce_wide_kernel, dim3(n), dim3(64), 0, s, gap, w, b, labels, probs, loss_per, dlogits, n,
                           c, k, loss_scale);
    else
    hipLaunchKernelGGL(dense_softmax_ce_kernel, dim3(n), dim3(64), 0, s, gap, w, b, labels, probs, loss_per, dlogits, n, c,
                       k, loss_scale);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_fan_head_bwd(const float* act, const float* gap, const float* w, const float* dlogits,
                      const float* loss_per, float* dact, float* dw, float* db, float* loss, int n, int hw, int c,
                      int k, float loss_scale, float alpha, void* stream) {
    if (!act || !gap || !w || !dlogits || !loss_per || !dact || !dw || !db || !loss) return NIMG_ERR_ARG;
    if (n <= 0 || hw <= 0 || c <= 0 || k <= 0 || k > 256) return NIMG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (k > 16)
        hipLaunchKernelGGL(dense_bwd_params_wide_kernel, dim3(c + 2), dim3(64), 0, s, gap, dlogits, loss_per, dw, db, loss, n, c,
                           k, loss_scale);
    else
    hipLaunchKernelGGL(dense_bwd_params_kernel, dim3(c + 2), dim3(64), 0, s, gap, dlogits, loss_per, dw, db, loss, n, c, k,
                       loss_scale);
    NIMG_CHECK_LAUNCH();
    const int parts = n >= 1024 ? 1 : (n >= 256 ? 4 : 16);                   // workgroups per image: enough to fill 256 CUs
    hipLaunchKernelGGL(gap_bwd_kernel, dim3((unsigned)(n * parts)), dim3(256), 0, s, dlogits, w, act, dact, hw, c, k, alpha,
                       parts);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_fan_dense_fwd(const float* gap, const float* w, const float* b, const int* labels, float* probs, float* loss_per,
                       float* dlogits, int n, int c, int k, float loss_scale, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!gap || !w || !b || !probs || n < 0 || c <= 0 || k <= 0 || k > 256) return NIMG_ERR_ARG;
    if (labels && (!loss_per || !dlogits)) return NIMG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (k > 16)
        hipLaunchKernelGGL(dense_softmax_ce_wide_kernel, dim3(n), dim3(64), 0, s, gap, w, b, labels, probs, loss_per, dlogits, n,
                           c, k, loss_scale);
    else
        hipLaunchKernelGGL(dense_softmax_ce_kernel, dim3(n), dim3(64), 0, s, gap, w, b, labels, probs, loss_per, dlogits, n, c,
                           k, loss_scale);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_fan_dense_bwd(const float* gap, const float* dlogits, const float* loss_per, float* dw, float* db, float* loss, int n,
                       int c, int k, float loss_scale, void* stream) {
    if (!gap || !dlogits || !loss_per || !dw || !db || !loss || n <= 0 || c <= 0 || k <= 0 || k > 256) return NIMG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (k > 16)
        hipLaunchKernelGGL(dense_bwd_params_wide_kernel, dim3(c + 2), dim3(64), 0, s, gap, dlogits, loss_per, dw, db, loss, n, c,
                           k, loss_scale);
    else
        hipLaunchKernelGGL(dense_bwd_params_kernel, dim3(c + 2), dim3(64), 0, s, gap, dlogits, loss_per, dw, db, loss, n, c, k,
                           loss_scale);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_adam_step(float* params, const float* grads, float* m, float* v, long count, float lr, float beta1,
                   float beta2, float eps, int step, float grad_scale, const int* skip_flag, void* stream) {
    if (!params || !grads || !m || !v || count < 0 || step < 1) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    const double lr_t = (double)lr * __builtin_sqrt(1.0 - __builtin_pow((double)beta2, (double)step)) /
                        (1.0 - __builtin_pow((double)beta1, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, params, grads, m, v,
                       count, (float)lr_t, beta1, beta2, eps, grad_scale, skip_flag, (const float*)nullptr);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_adam_step_dev(float* params, const float* grads, float* m, float* v, long count, const float* lr_t, float beta1,
                       float beta2, float eps, float grad_scale, const int* skip_flag, void* stream) {
    if (!params || !grads || !m || !v || !lr_t || count < 0) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, params, grads, m, v,
                       count, 0.f, beta1, beta2, eps, grad_scale, skip_flag, lr_t);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

/* The step's flag / scalar bookkeeping as kernels of this library (no framework-native launch inside a training step):
 * mode 0: dst[0 .. n) = value;  mode 1: dst[i] = max(dst[i], src[i]) (the workflow's "NaN seen since the last check" word) */
int nimg_int_words(int* dst, const int* src, long n, int value, int mode, void* stream) {
    if (!dst || n < 0 || mode < 0 || mode > 1 || (mode == 1 && !src)) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    hipLaunchKernelGGL(int_words_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dst, src, n, value, mode);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

// ---- arrival counters of the in-kernel split-K finish (common.h ticket_finish): caller-owned zeroed device memory per stream
namespace {
struct TicketBinding { hipStream_t stream; unsigned* buf; size_t words; bool used; };
TicketBinding g_tickets[NIMG_TICKET_STREAMS];
std::mutex g_tickets_mutex;
}  // namespace

int nimg_bind_tickets(void* stream, void* buf, size_t bytes) {
    if ((buf && (bytes < 4 || (bytes & 3))) || ((uintptr_t)buf & 3)) return NIMG_ERR_ARG;
    std::lock_guard<std::mutex> lock(g_tickets_mutex);
    int free_slot = -1;
    for (int i = 0; i < NIMG_TICKET_STREAMS; ++i) {
        if (g_tickets[i].used && g_tickets[i].stream == (hipStream_t)stream) {
            if (buf) { g_tickets[i].buf = (unsigned*)buf; g_tickets[i].words = bytes / 4; }
            else g_tickets[i].used = false;
            return NIMG_OK;
        }
        if (!g_tickets[i].used && free_slot < 0) free_slot = i;
    }
    if (!buf) return NIMG_OK;
    if (free_slot < 0) return NIMG_ERR_ARG;
    g_tickets[free_slot] = TicketBinding{(hipStream_t)stream, (unsigned*)buf, bytes / 4, true};
    return NIMG_OK;
}

int nimg_stream_create_cu_mask(int n_cus, void** stream) {
    if (!stream || n_cus < 1 || n_cus > 1024) return NIMG_ERR_ARG;
    uint32_t mask[32] = {0};
    for (int i = 0; i < n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)((n_cus + 31) / 32), mask) != hipSuccess) return NIMG_ERR_LAUNCH;
    *stream = (void*)s;
    return NIMG_OK;
}

int nimg_stream_destroy(void* stream) {
    if (!stream) return NIMG_ERR_ARG;
    return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? NIMG_OK : NIMG_ERR_LAUNCH;
}

unsigned* nimg_internal_tickets(hipStream_t stream, size_t words) {
    static const bool off = getenv("NIMG_NO_TICKETS") != nullptr;
    if (off) return nullptr;
    std::lock_guard<std::mutex> lock(g_tickets_mutex);
    for (int i = 0; i < NIMG_TICKET_STREAMS; ++i)
        if (g_tickets[i].used && g_tickets[i].stream == stream) return g_tickets[i].words >= words ? g_tickets[i].buf : nullptr;
    return nullptr;
}

int nimg_float_fill(float* dst, long n, float value, void* stream) {
    if (!dst || n < 0) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    hipLaunchKernelGGL(float_fill_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dst, n, value);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_nan_flag(const float* g, long count, int* flag, void* stream) {
    if (!g || !flag || count < 0) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    hipLaunchKernelGGL(nan_flag_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, g, count, flag);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

}  // extern "C"
