// Photo-manipulation stencils of the channel (HBM-bound, 3-channel NHWC images), forward and backward:
//   manipulation_sharpen   helpers/tf_helpers.py:156-184  (SYMMETRIC pad, rgb->hsv, 3x3 filter incl. the S-channel
//                                                          corner-tap quirk of :167-169, hsv->rgb, hard clip)
//   manipulation_gaussian  helpers/tf_helpers.py:113-125  (REFLECT pad, 5x5 depthwise gaussian, hard clip)
//   manipulation_resample  helpers/tf_helpers.py:68-76    (bilinear down + up == one banded linear operator per axis,
//                                                          applied as a sparse row gather; the backward uses its transpose)
//   fold of a padded-domain gradient back onto the image (backward of tf.pad SYMMETRIC / REFLECT; used by the
//   ConstrainedConv2D input gradient, models/layers.py:56)
// One thread per pixel; neighbour reads are served by L1/L2 (a 3-channel image row is 3 KB).
#include "common.h"

namespace {

using namespace nimg;

inline int grid_for(long items) {
    long g = (items + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

// padded-domain indices that map onto image index y (own + at most one mirrored), pad P, mode 1 SYMMETRIC / 2 REFLECT
__device__ __forceinline__ int pad_sources(int y, int size, int P, int mode, int (&out)[2]) {
    out[0] = y + P;
    int cnt = 1;
    if (mode == 1) {
        if (y < P) out[cnt++] = P - 1 - y;
        else if (y >= size - P) out[cnt++] = 2 * size + P - 1 - y;
    } else if (mode == 2) {
        if (y >= 1 && y <= P) out[cnt++] = P - y;
        else if (y >= size - 1 - P && y <= size - 2) out[cnt++] = 2 * size + P - 2 - y;
    }
    return cnt;
}

__device__ __forceinline__ void rgb2hsv(float r, float g, float b, float& h, float& s, float& v) {
    // tf.image.rgb_to_hsv; max/min selection order R, G, B (matches the branch order of the hue)
    v = (r >= g && r >= b) ? r : (g >= b ? g : b);
    const float mn = (r <= g && r <= b) ? r : (g <= b ? g : b);
    const float rng = v - mn;
    s = v > 0.f ? rng / v : 0.f;
    const float norm = 1.0f / (6.0f * (rng > 0.f ? rng : 1.0f));
    float hh = (r == v) ? norm * (g - b) : ((g == v) ? norm * (b - r) + 2.0f / 6.0f : norm * (r - g) + 4.0f / 6.0f);
    hh = rng > 0.f ? hh : 0.f;
    h = hh < 0.f ? hh + 1.0f : hh;
}

__device__ __forceinline__ void hsv2rgb(float h, float s, float v, float& r, float& g, float& b) {
    const float dh = h * 6.0f;
    const float dr = fminf(fmaxf(fabsf(dh - 3.0f) - 1.0f, 0.f), 1.f);
    const float dg = fminf(fmaxf(2.0f - fabsf(dh - 2.0f), 0.f), 1.f);
    const float db = fminf(fmaxf(2.0f - fabsf(dh - 4.0f), 0.f), 1.f);
    const float one_s = 1.0f - s;
    r = (one_s + s * dr) * v;
    g = (one_s + s * dg) * v;
    b = (one_s + s * db) * v;
}

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// ------------------------------------------------------------------------------------------------------------------
__global__ void gaussian_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ mask,
                                    const float* __restrict__ gk, int n, int h, int w, int clip) {
    const long total = (long)n * h * w;
    float g[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) g[k] = gk[k];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int px = (int)(i % w), py = (int)((i / w) % h);
        const long im = i / ((long)w * h);
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 5; ++ky) {
            int yy = py + ky - 2;
            map_coord(yy, h, 2);
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                int xx = px + kx - 2;
                map_coord(xx, w, 2);
                const float* p = x + ((im * h + yy) * w + xx) * 3;
                const float wv = g[ky * 5 + kx];
                acc[0] = fmaf(p[0], wv, acc[0]);
                acc[1] = fmaf(p[1], wv, acc[1]);
                acc[2] = fmaf(p[2], wv, acc[2]);
            }
        }
        uint32_t m = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            m |= (acc[c] >= 0.f && acc[c] <= 1.f) ? (1u << c) : 0u;
            y[i * 3 + c] = clip ? fminf(fmaxf(acc[c], 0.f), 1.f) : acc[c];
        }
        if (mask) mask[i] = clip ? (uint8_t)m : (uint8_t)7;
    }
}

// LDS-tiled forward (images of at least 16 x 16): a 16 x 16 tile + its 2-pixel ring (REFLECT-mapped while staging) is read
// once as 20 x 20 float4 instead of 25 x 3 scalar loads per output pixel; same tap order (ky, then kx) and the same fmaf chain
// as gaussian_fwd_kernel, so the results are identical.
__global__ __launch_bounds__(256) void gaussian_fwd_tiled_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 uint8_t* __restrict__ mask, const float* __restrict__ gk,
                                                                 int n, int h, int w, int clip, int tiles_y, int tiles_x) {
    __shared__ float4 sx[20 * 21];
    __shared__ float sg[25];
    const int tid = threadIdx.x;
    if (tid < 25) sg[tid] = gk[tid];
    const int tiles = tiles_y * tiles_x;
    const long im = blockIdx.x / tiles;
    const int tile = blockIdx.x % tiles, y0 = (tile / tiles_x) * 16, x0 = (tile % tiles_x) * 16;
    for (int i = tid; i < 400; i += 256) {
        const int r = i / 20, c = i % 20;
        int gy = y0 - 2 + r, gx = x0 - 2 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        // rows / columns beyond the reflected ring of a partial tile are never read by a valid output pixel
        if (gy < h + 2 && gx < w + 2) {
            map_coord(gy, h, 2);
            map_coord(gx, w, 2);
            const float* p = x + ((im * h + gy) * w + gx) * 3;
            v = make_float4(p[0], p[1], p[2], 0.f);
        }
        sx[r * 21 + c] = v;
    }
    __syncthreads();
    const int ly = tid >> 4, lx = tid & 15, py = y0 + ly, px = x0 + lx;
    if (py >= h || px >= w) return;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
            const float4 v = sx[(ly + ky) * 21 + lx + kx];
            const float wv = sg[ky * 5 + kx];
            acc[0] = fmaf(v.x, wv, acc[0]);
            acc[1] = fmaf(v.y, wv, acc[1]);
            acc[2] = fmaf(v.z, wv, acc[2]);
        }
    const long i = (im * h + py) * w + px;
    uint32_t m = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        m |= (acc[c] >= 0.f && acc[c] <= 1.f) ? (1u << c) : 0u;
        y[i * 3 + c] = clip ? fminf(fmaxf(acc[c], 0.f), 1.f) : acc[c];
    }
    if (mask) mask[i] = clip ? (uint8_t)m : (uint8_t)7;
}

// Wide form (h % 16 == 0, w % 64 == 0: the bench images): a 16 x 64 tile, the thread owns FOUR consecutive pixels of a row.
// The tile + its 2-pixel ring lives in LDS as the flat rows of the image (pixel c of the ring-extended row at floats 3c ..
// 3c + 2): the 64 interior pixels of a row are 48 aligned 16-byte global loads (the 12-byte pixel accesses of the 16 x 16
// form needed 3 dword loads per pixel), only the 2 + 2 ring pixels are mapped (REFLECT) one by one; a thread's 8-pixel
// window of a tile row is six aligned ds_read_b128 shared by its four outputs (30 LDS reads per 4 pixels instead of 100),
// its 12 results leave as three 16-byte stores and its four mask bytes as one dword.  Per output the same fmaf chain (ky, then
// kx) as gaussian_fwd_kernel: identical results.
constexpr int GW_RS = 208;            // floats per LDS row: 68 pixels x 3 = 204, padded to a multiple of 4
__global__ __launch_bounds__(256) void gaussian_fwd_wide_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                uint8_t* __restrict__ mask, const float* __restrict__ gk,
                                                                int n, int h, int w, int clip, int tiles_y, int tiles_x) {
    __shared__ __attribute__((aligned(16))) float sx[20 * GW_RS];
    const int tid = threadIdx.x;
    const int tiles = tiles_y * tiles_x;
    const long im = blockIdx.x / tiles;
    const int tile = blockIdx.x % tiles, y0 = (tile / tiles_x) * 16, x0 = (tile % tiles_x) * 64;
    for (int i = tid; i < 20 * 48; i += 256) {                   // interior columns: 48 float4 per staged row
        const int r = i / 48, q = i % 48;
        int gy = y0 - 2 + r;
        map_coord(gy, h, 2);
        const float4 v = reinterpret_cast<const float4*>(x + ((im * h + gy) * w + x0) * 3)[q];
        float2* d = reinterpret_cast<float2*>(sx + r * GW_RS + 6 + 4 * q);       // 8-byte aligned (the ring shifts by 6 floats)
        d[0] = make_float2(v.x, v.y);
        d[1] = make_float2(v.z, v.w);
    }
    if (tid < 240) {                                             // ring columns: 4 pixels x 3 floats per staged row
        const int r = tid / 12, j = tid % 12, c = j < 6 ? j / 3 : 64 + j / 3, ch = j % 3;
        int gy = y0 - 2 + r, gx = x0 - 2 + c;
        map_coord(gy, h, 2);
        map_coord(gx, w, 2);
        sx[r * GW_RS + c * 3 + ch] = x[((im * h + gy) * w + gx) * 3 + ch];
    }
    __syncthreads();
    const int ly = tid >> 4, lq = tid & 15;                      // pixels 4 lq .. 4 lq + 3 of tile row ly
    float acc[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j][0] = acc[j][1] = acc[j][2] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
        float win[24];
        const float4* src = reinterpret_cast<const float4*>(sx + (ly + ky) * GW_RS + 12 * lq);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const float4 v = src[q];
            win[4 * q] = v.x; win[4 * q + 1] = v.y; win[4 * q + 2] = v.z; win[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
            const float wv = gk[ky * 5 + kx];                    // uniform: scalar loads
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[j][c] = fmaf(win[(j + kx) * 3 + c], wv, acc[j][c]);
        }
    }
    const long i0 = (im * h + y0 + ly) * w + x0 + 4 * lq;
    uint32_t mbits = 0;
    float o[12];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a = acc[j][c];
            mbits |= (a >= 0.f && a <= 1.f) ? (1u << (8 * j + c)) : 0u;
            o[j * 3 + c] = clip ? fminf(fmaxf(a, 0.f), 1.f) : a;
        }
    float4* dst = reinterpret_cast<float4*>(y + i0 * 3);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]);
    dst[1] = make_float4(o[4], o[5], o[6], o[7]);
    dst[2] = make_float4(o[8], o[9], o[10], o[11]);
    if (mask) *reinterpret_cast<uint32_t*>(mask + i0) = clip ? mbits : 0x07070707u;
}

__global__ void gaussian_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ mask,
                                    float* __restrict__ dx, const float* __restrict__ gk, int n, int h, int w) {
    const long total = (long)n * h * w;
    float g[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) g[k] = gk[k];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int px = (int)(i % w), py = (int)((i / w) % h);
        const long im = i / ((long)w * h);
        int ys[2], xs[2];
        const int ny = pad_sources(py, h, 2, 2, ys), nx = pad_sources(px, w, 2, 2, xs);
        float acc[3] = {0.f, 0.f, 0.f};
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) {
#pragma unroll
                for (int ky = 0; ky < 5; ++ky) {
                    const int oy = ys[a] - ky;           // output row that read padded row ys[a] with tap ky
                    if (oy < 0 || oy >= h) continue;
#pragma unroll
                    for (int kx = 0; kx < 5; ++kx) {
                        const int ox = xs[b] - kx;
                        if (ox < 0 || ox >= w) continue;
                        const long o = (im * h + oy) * w + ox;
                        const uint32_t m = mask ? mask[o] : 7u;
                        const float wv = g[ky * 5 + kx];
                        const float* p = dy + o * 3;
                        if (m & 1u) acc[0] = fmaf(p[0], wv, acc[0]);
                        if (m & 2u) acc[1] = fmaf(p[1], wv, acc[1]);
                        if (m & 4u) acc[2] = fmaf(p[2], wv, acc[2]);
                    }
                }
            }
        dx[i * 3 + 0] = acc[0];
        dx[i * 3 + 1] = acc[1];
        dx[i * 3 + 2] = acc[2];
    }
}

// LDS-tiled variant (used when the image holds at least one 16x16 tile): the masked dy of a 16x16 tile + its 2-pixel
// ring is staged once (20 x 20 x float4) instead of 25 global mask + pixel reads per output pixel; same summation order
// as gaussian_bwd_kernel (sources, then ky, then kx), so the results are identical.
__global__ __launch_bounds__(256) void gaussian_bwd_tiled_kernel(const float* __restrict__ dy,
                                                                 const uint8_t* __restrict__ mask,
                                                                 float* __restrict__ dx, const float* __restrict__ gk,
                                                                 int n, int h, int w, int tiles_y, int tiles_x) {
    __shared__ float4 sd[20 * 21];
    __shared__ float sg[25];
    const int tid = threadIdx.x;
    if (tid < 25) sg[tid] = gk[tid];
    const int tiles = tiles_y * tiles_x;
    const long im = blockIdx.x / tiles;
    const int tile = blockIdx.x % tiles, y0 = (tile / tiles_x) * 16, x0 = (tile % tiles_x) * 16;
    for (int i = tid; i < 400; i += 256) {
        const int r = i / 20, c = i % 20, gy = y0 - 2 + r, gx = x0 - 2 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < h && gx >= 0 && gx < w) {
            const long o = (im * h + gy) * w + gx;
            const uint32_t m = mask ? mask[o] : 7u;
            const float* p = dy + o * 3;
            const float p0 = p[0], p1 = p[1], p2 = p[2];      // unconditional loads, then selects (no branch per channel)
            v = make_float4((m & 1u) ? p0 : 0.f, (m & 2u) ? p1 : 0.f, (m & 4u) ? p2 : 0.f, 0.f);
        }
        sd[r * 21 + c] = v;
    }
    __syncthreads();
    const int py = y0 + (tid >> 4), px = x0 + (tid & 15);
    if (py >= h || px >= w) return;
    int ys[2], xs[2];
    const int ny = pad_sources(py, h, 2, 2, ys), nx = pad_sources(px, w, 2, 2, xs);
    float acc[3] = {0.f, 0.f, 0.f};
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) {
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                const int oy = ys[a] - ky;
                if (oy < 0 || oy >= h) continue;
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
                    const int ox = xs[b] - kx;
                    if (ox < 0 || ox >= w) continue;
                    const float4 v = sd[(oy - y0 + 2) * 21 + (ox - x0 + 2)];
                    const float wv = sg[ky * 5 + kx];
                    acc[0] = fmaf(v.x, wv, acc[0]);
                    acc[1] = fmaf(v.y, wv, acc[1]);
                    acc[2] = fmaf(v.z, wv, acc[2]);
                }
            }
        }
    const long i = (im * h + py) * w + px;
    dx[i * 3 + 0] = acc[0];
    dx[i * 3 + 1] = acc[1];
    dx[i * 3 + 2] = acc[2];
}

// Wide form of the backward pass (h % 16 == 0, w % 64 == 0), laid out like gaussian_fwd_wide_kernel: the masked gradient of a
// 16 x 64 tile + ring as flat rows (zeros outside the image), four pixels per thread.  Every pixel first takes the term of its
// own position in the padded domain - 25 taps over the shared register window, order (ky, kx) - then, for the pixels within two
// of a border, the terms of the mirrored positions in the order of gaussian_bwd_kernel's loops: the same fmaf chain, identical
// results.
__global__ __launch_bounds__(256) void gaussian_bwd_wide_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ mask,
                                                                float* __restrict__ dx, const float* __restrict__ gk,
                                                                int n, int h, int w, int tiles_y, int tiles_x) {
    __shared__ __attribute__((aligned(16))) float sd[20 * GW_RS];
    const int tid = threadIdx.x;
    const int tiles = tiles_y * tiles_x;
    const long im = blockIdx.x / tiles;
    const int tile = blockIdx.x % tiles, y0 = (tile / tiles_x) * 16, x0 = (tile % tiles_x) * 64;
    for (int i = tid; i < 20 * 48; i += 256) {
        const int r = i / 48, q = i % 48, gy = y0 - 2 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < h) {
            const long o = (im * h + gy) * w + x0;
            v = reinterpret_cast<const float4*>(dy + o * 3)[q];
            if (mask) {                                          // floats 4q .. 4q + 3 of the row belong to pixels p0 and p0 + 1
                const int p0 = (4 * q) / 3, c0 = (4 * q) % 3;
                const uint32_t m0 = mask[o + p0], m1 = mask[o + (p0 + 1 < 64 ? p0 + 1 : p0)];
                const uint32_t bits = (m0 >> c0) | (m1 << (3 - c0));       // bit e = keep float 4q + e
                v.x = (bits & 1u) ? v.x : 0.f; v.y = (bits & 2u) ? v.y : 0.f;
                v.z = (bits & 4u) ? v.z : 0.f; v.w = (bits & 8u) ? v.w : 0.f;
            }
        }
        float2* d = reinterpret_cast<float2*>(sd + r * GW_RS + 6 + 4 * q);
        d[0] = make_float2(v.x, v.y);
        d[1] = make_float2(v.z, v.w);
    }
    if (tid < 240) {
        const int r = tid / 12, j = tid % 12, c = j < 6 ? j / 3 : 64 + j / 3, ch = j % 3;
        const int gy = y0 - 2 + r, gx = x0 - 2 + c;
        float v = 0.f;
        if (gy >= 0 && gy < h && gx >= 0 && gx < w) {
            const long o = (im * h + gy) * w + gx;
            const uint32_t m = mask ? mask[o] : 7u;
            v = ((m >> ch) & 1u) ? dy[o * 3 + ch] : 0.f;
        }
        sd[r * GW_RS + c * 3 + ch] = v;
    }
    __syncthreads();
    const int ly = tid >> 4, lq = tid & 15, py = y0 + ly;
    float acc[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j][0] = acc[j][1] = acc[j][2] = 0.f;
    // own position: dx[py][px] += w[ky][kx] d[py + 2 - ky][px + 2 - kx]  ->  tile row ly + 4 - ky, window pixel j + 4 - kx
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
        float win[24];
        const float4* src = reinterpret_cast<const float4*>(sd + (ly + 4 - ky) * GW_RS + 12 * lq);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const float4 v = src[q];
            win[4 * q] = v.x; win[4 * q + 1] = v.y; win[4 * q + 2] = v.z; win[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
            const float wv = gk[ky * 5 + kx];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[j][c] = fmaf(win[(j + 4 - kx) * 3 + c], wv, acc[j][c]);
        }
    }
    // mirrored positions (REFLECT, P = 2): only rows / columns 1, 2 and size - 3, size - 2 have one
    int ys[2];
    const int ny = pad_sources(py, h, 2, 2, ys);
    if (ny == 1) {
        // rows away from the top / bottom border: only the first / last four pixels of an image row have a mirrored column.
        // Column 1 mirrors to padded column 1 (taps kx = 0, 1 reach columns 1, 0), column 2 to padded column 0 (kx = 0 reaches
        // column 0); on the right w - 3 -> w + 3 (kx = 4 reaches w - 1), w - 2 -> w + 2 (kx = 3, 4 reach w - 1, w - 2).  Those
        // columns are window pixels 2, 3 (left) and 4, 5 (right) of the thread that owns the border pixels.
        const bool lb = x0 + 4 * lq == 0, rb = x0 + 4 * lq + 4 == w;
        if (lb || rb) {
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                const float4* src = reinterpret_cast<const float4*>(sd + (ly + 4 - ky) * GW_RS + 12 * lq);
                const float4 q1 = src[1], q2 = src[2], q3 = src[3], q4 = src[4];
                const float win[16] = {q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, q4.z, q4.w};
                // win[f - 4] = float f of the window: pixel 2 = floats 6..8, 3 = 9..11, 4 = 12..14, 5 = 15..17
                const float g0 = gk[ky * 5], g1 = gk[ky * 5 + 1], g3 = gk[ky * 5 + 3], g4 = gk[ky * 5 + 4];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (lb) {
                        acc[1][c] = fmaf(win[9 - 4 + c], g0, acc[1][c]);
                        acc[1][c] = fmaf(win[6 - 4 + c], g1, acc[1][c]);
                        acc[2][c] = fmaf(win[6 - 4 + c], g0, acc[2][c]);
                    } else {
                        acc[1][c] = fmaf(win[15 - 4 + c], g4, acc[1][c]);
                        acc[2][c] = fmaf(win[15 - 4 + c], g3, acc[2][c]);
                        acc[2][c] = fmaf(win[12 - 4 + c], g4, acc[2][c]);
                    }
                }
            }
        }
    } else {
#pragma unroll                                                   // (acc[j] must stay in registers)
        for (int j = 0; j < 4; ++j) {
            const int px = x0 + 4 * lq + j;
            int xs[2];
            const int nx = pad_sources(px, w, 2, 2, xs);
            for (int a = 0; a < ny; ++a)
                for (int b = 0; b < nx; ++b) {
                    if (a == 0 && b == 0) continue;              // the own position, summed above
                    for (int ky = 0; ky < 5; ++ky) {
                        const int oy = ys[a] - ky;
                        if (oy < 0 || oy >= h) continue;
                        for (int kx = 0; kx < 5; ++kx) {
                            const int ox = xs[b] - kx;
                            if (ox < 0 || ox >= w) continue;
                            const float* v = sd + (oy - y0 + 2) * GW_RS + (ox - x0 + 2) * 3;
                            const float wv = gk[ky * 5 + kx];
                            acc[j][0] = fmaf(v[0], wv, acc[j][0]);
                            acc[j][1] = fmaf(v[1], wv, acc[j][1]);
                            acc[j][2] = fmaf(v[2], wv, acc[j][2]);
                        }
                    }
                }
        }
    }
    float4* dst = reinterpret_cast<float4*>(dx + ((im * h + py) * w + x0 + 4 * lq) * 3);
    dst[0] = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[1][0]);
    dst[1] = make_float4(acc[1][1], acc[1][2], acc[2][0], acc[2][1]);
    dst[2] = make_float4(acc[2][2], acc[3][0], acc[3][1], acc[3][2]);
}

// ------------------------------------------------------------------------------------------------------------------
// sharpen forward: y = clip(hsv2rgb(filter(rgb2hsv(sympad(x))))); aux = filtered hsv (needed by the backward)
__global__ void sharpen_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ aux,
                                   uint8_t* __restrict__ mask, const float* __restrict__ gk9, int n, int h, int w) {
    const long total = (long)n * h * w;
    float g[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k] = gk9[k];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int px = (int)(i % w), py = (int)((i / w) % h);
        const long im = i / ((long)w * h);
        float ho = 0.f, vo = 0.f, so = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            int yy = py + ky - 1;
            map_coord(yy, h, 1);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                int xx = px + kx - 1;
                map_coord(xx, w, 1);
                const float* p = x + ((im * h + yy) * w + xx) * 3;
                float hh, ss, vv;
                rgb2hsv(p[0], p[1], p[2], hh, ss, vv);
                ho = fmaf(hh, g[ky * 3 + kx], ho);
                vo = fmaf(vv, g[ky * 3 + kx], vo);
                if (ky == 2 && kx == 2) so = ss;          // S channel: single corner tap [2,2] = 1
            }
        }
        float r, gg, b;
        hsv2rgb(ho, so, vo, r, gg, b);
        const float o[3] = {r, gg, b};
        uint32_t m = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            m |= (o[c] >= 0.f && o[c] <= 1.f) ? (1u << c) : 0u;
            y[i * 3 + c] = fminf(fmaxf(o[c], 0.f), 1.f);
        }
        if (aux) { aux[i * 3] = ho; aux[i * 3 + 1] = so; aux[i * 3 + 2] = vo; }
        if (mask) mask[i] = (uint8_t)m;
    }
}

// LDS-tiled sharpen forward (images of at least 16 x 16): rgb2hsv (two divisions) is evaluated ONCE per pixel of the 18 x 18
// SYMMETRIC-mapped neighbourhood of a 16 x 16 tile instead of nine times per output pixel; the 3 x 3 filter then reads the
// staged (h, s, v) triples.  Same per-pixel functions and the same tap order as sharpen_fwd_kernel: identical results.
__global__ __launch_bounds__(256) void sharpen_fwd_tiled_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                float* __restrict__ aux, uint8_t* __restrict__ mask,
                                                                const float* __restrict__ gk9, int n, int h, int w,
                                                                int tiles_y, int tiles_x) {
    __shared__ float4 sh[18 * 19];
    __shared__ float sg[9];
    const int tid = threadIdx.x;
    if (tid < 9) sg[tid] = gk9[tid];
    const int tiles = tiles_y * tiles_x;
    const long im = blockIdx.x / tiles;
    const int tile = blockIdx.x % tiles, y0 = (tile / tiles_x) * 16, x0 = (tile % tiles_x) * 16;
    for (int i = tid; i < 324; i += 256) {
        const int r = i / 18, c = i % 18;
        int gy = y0 - 1 + r, gx = x0 - 1 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy < h + 1 && gx < w + 1) {                    // beyond the mirrored ring of a partial tile: never read
            map_coord(gy, h, 1);
            map_coord(gx, w, 1);
            const float* p = x + ((im * h + gy) * w + gx) * 3;
            rgb2hsv(p[0], p[1], p[2], v.x, v.y, v.z);
        }
        sh[r * 19 + c] = v;
    }
    __syncthreads();
    const int ly = tid >> 4, lx = tid & 15, py = y0 + ly, px = x0 + lx;
    if (py >= h || px >= w) return;
    float ho = 0.f, vo = 0.f, so = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float4 t = sh[(ly + ky) * 19 + lx + kx];
            ho = fmaf(t.x, sg[ky * 3 + kx], ho);
            vo = fmaf(t.z, sg[ky * 3 + kx], vo);
            if (ky == 2 && kx == 2) so = t.y;             // S channel: single corner tap [2,2] = 1
        }
    float r, gg, b;
    hsv2rgb(ho, so, vo, r, gg, b);
    const float o[3] = {r, gg, b};
    const long i = (im * h + py) * w + px;
    uint32_t m = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        m |= (o[c] >= 0.f && o[c] <= 1.f) ? (1u << c) : 0u;
        y[i * 3 + c] = fminf(fmaxf(o[c], 0.f), 1.f);
    }
    if (aux) { aux[i * 3] = ho; aux[i * 3 + 1] = so; aux[i * 3 + 2] = vo; }
    if (mask) mask[i] = (uint8_t)m;
}

// backward stage A (per output pixel): d(filtered hsv) = J_hsv2rgb^T (dy * clipmask); written in place of aux
__global__ void sharpen_bwd_a_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ mask, float* aux,
                                     long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const float h = aux[i * 3], s = aux[i * 3 + 1], v = aux[i * 3 + 2];
        const uint32_t m = mask[i];
        const float gr = (m & 1u) ? dy[i * 3] : 0.f, gg = (m & 2u) ? dy[i * 3 + 1] : 0.f,
                    gb = (m & 4u) ? dy[i * 3 + 2] : 0.f;
        const float dh = h * 6.0f;
        const float ar = fabsf(dh - 3.0f) - 1.0f, ag = 2.0f - fabsf(dh - 2.0f), ab = 2.0f - fabsf(dh - 4.0f);
        const float dr = fminf(fmaxf(ar, 0.f), 1.f), dg = fminf(fmaxf(ag, 0.f), 1.f), db = fminf(fmaxf(ab, 0.f), 1.f);
        // clamp passes gradient on the closed interval [0,1]; d|x| = sign(x) with sign(0) = 0
        const float ddr = (ar >= 0.f && ar <= 1.f) ? 6.0f * sgn(dh - 3.0f) : 0.f;
        const float ddg = (ag >= 0.f && ag <= 1.f) ? -6.0f * sgn(dh - 2.0f) : 0.f;
        const float ddb = (ab >= 0.f && ab <= 1.f) ? -6.0f * sgn(dh - 4.0f) : 0.f;
        const float one_s = 1.0f - s;
        aux[i * 3] = s * v * (gr * ddr + gg * ddg + gb * ddb);
        aux[i * 3 + 1] = v * (gr * (dr - 1.0f) + gg * (dg - 1.0f) + gb * (db - 1.0f));
        aux[i * 3 + 2] = gr * (one_s + s * dr) + gg * (one_s + s * dg) + gb * (one_s + s * db);
    }
}

// rgb_to_hsv backward at one pixel: (gh, gs, gv) = gradient w.r.t. its (h, s, v) -> d = gradient w.r.t. (r, g, b)
__device__ __forceinline__ void hsv_backward(float r, float gg, float bb, float gh, float gs, float gv, float (&d)[3]) {
    const int imax = (r >= gg && r >= bb) ? 0 : (gg >= bb ? 1 : 2);
    const int imin = (r <= gg && r <= bb) ? 0 : (gg <= bb ? 1 : 2);
    const float c3[3] = {r, gg, bb};
    const float v = c3[imax], mn = c3[imin], rng = v - mn;
    d[0] = d[1] = d[2] = 0.f;
    float g_rng = 0.f, g_v = gv;
    if (v > 0.f) { g_rng += gs / v; g_v -= gs * rng / (v * v); }
    if (rng > 0.f) {
        const float norm = 1.0f / (6.0f * rng);
        // hue branch (same order as the forward): operands (a - b)
        int ia, ib;
        if (r == v) { ia = 1; ib = 2; } else if (gg == v) { ia = 2; ib = 0; } else { ia = 0; ib = 1; }
        d[ia] += gh * norm;
        d[ib] -= gh * norm;
        g_rng -= gh * norm * (c3[ia] - c3[ib]) / rng;
    }
    d[imax] += g_v + g_rng;
    d[imin] -= g_rng;
}

// LDS-tiled stage B (images of at least 16 x 16): the 18 x 18 neighbourhood of d(filtered hsv) is staged once (zeros outside
// the image); same source / tap order as sharpen_bwd_b_kernel, identical results.
__global__ __launch_bounds__(256) void sharpen_bwd_b_tiled_kernel(const float* __restrict__ x, const float* __restrict__ dhsv,
                                                                  float* __restrict__ dx, const float* __restrict__ gk9,
                                                                  int n, int h, int w, int tiles_y, int tiles_x) {
    __shared__ float4 sd[18 * 19];
    __shared__ float sg[9];
    const int tid = threadIdx.x;
    if (tid < 9) sg[tid] = gk9[tid];
    const int tiles = tiles_y * tiles_x;
    const long im = blockIdx.x / tiles;
    const int tile = blockIdx.x % tiles, y0 = (tile / tiles_x) * 16, x0 = (tile % tiles_x) * 16;
    for (int i = tid; i < 324; i += 256) {
        const int r = i / 18, c = i % 18, gy = y0 - 1 + r, gx = x0 - 1 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < h && gx >= 0 && gx < w) {
            const float* p = dhsv + ((im * h + gy) * w + gx) * 3;
            v = make_float4(p[0], p[1], p[2], 0.f);
        }
        sd[r * 19 + c] = v;
    }
    __syncthreads();
    const int py = y0 + (tid >> 4), px = x0 + (tid & 15);
    if (py >= h || px >= w) return;
    int ys[2], xs[2];
    const int ny = pad_sources(py, h, 1, 1, ys), nx = pad_sources(px, w, 1, 1, xs);
    float gh = 0.f, gs = 0.f, gv = 0.f;
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int oy = ys[a] - ky;
                if (oy < 0 || oy >= h) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ox = xs[b] - kx;
                    if (ox < 0 || ox >= w) continue;
                    const float4 t = sd[(oy - y0 + 1) * 19 + (ox - x0 + 1)];
                    gh = fmaf(t.x, sg[ky * 3 + kx], gh);
                    gv = fmaf(t.z, sg[ky * 3 + kx], gv);
                    if (ky == 2 && kx == 2) gs += t.y;
                }
            }
        }
    const long i = (im * h + py) * w + px;
    float d[3];
    hsv_backward(x[i * 3], x[i * 3 + 1], x[i * 3 + 2], gh, gs, gv, d);
    dx[i * 3] = d[0];
    dx[i * 3 + 1] = d[1];
    dx[i * 3 + 2] = d[2];
}

// backward stage B (per input pixel): gather the filter transpose + pad fold, then J_rgb2hsv^T
__global__ void sharpen_bwd_b_kernel(const float* __restrict__ x, const float* __restrict__ dhsv,
                                     float* __restrict__ dx, const float* __restrict__ gk9, int n, int h, int w) {
    const long total = (long)n * h * w;
    float g[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k] = gk9[k];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int px = (int)(i % w), py = (int)((i / w) % h);
        const long im = i / ((long)w * h);
        int ys[2], xs[2];
        const int ny = pad_sources(py, h, 1, 1, ys), nx = pad_sources(px, w, 1, 1, xs);
        float gh = 0.f, gs = 0.f, gv = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int oy = ys[a] - ky;
                    if (oy < 0 || oy >= h) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int ox = xs[b] - kx;
                        if (ox < 0 || ox >= w) continue;
                        const float* p = dhsv + ((im * h + oy) * w + ox) * 3;
                        gh = fmaf(p[0], g[ky * 3 + kx], gh);
                        gv = fmaf(p[2], g[ky * 3 + kx], gv);
                        if (ky == 2 && kx == 2) gs += p[1];
                    }
                }
            }
        float d[3];
        hsv_backward(x[i * 3], x[i * 3 + 1], x[i * 3 + 2], gh, gs, gv, d);
        dx[i * 3] = d[0];
        dx[i * 3 + 1] = d[1];
        dx[i * 3 + 2] = d[2];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// out[n, o, x, c] (axis 0) or out[n, y, o, c] (axis 1) = sum_e val[e] * in[.., col[e], ..],  e in [rowptr[o], rowptr[o+1])
__global__ void sparse_axis_kernel(const float* __restrict__ in, float* __restrict__ out,
                                   const int* __restrict__ rowptr, const int* __restrict__ col,
                                   const float* __restrict__ val, int n, int hin, int win, int hout, int wout, int c,
                                   int axis) {
    const long total = (long)n * hout * wout * c;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        long r = i / c;
        const int ox = (int)(r % wout);
        r /= wout;
        const int oy = (int)(r % hout);
        const long im = r / hout;
        const int o = axis == 0 ? oy : ox;
        float acc = 0.f;
        for (int e = rowptr[o]; e < rowptr[o + 1]; ++e) {
            const int src = col[e];
            const long idx = axis == 0 ? ((im * hin + src) * win + ox) * c + ch : ((im * hin + oy) * win + src) * c + ch;
            acc = fmaf(val[e], in[idx], acc);
        }
        out[i] = acc;
    }
}

// 3-channel images (every caller on the hot path): one thread per output PIXEL, so the CSR row (pointer, columns,
// values) and the 32-bit index arithmetic are shared by the channels
__global__ void sparse_axis3_kernel(const float* __restrict__ in, float* __restrict__ out,
                                    const int* __restrict__ rowptr, const int* __restrict__ col,
                                    const float* __restrict__ val, int n, int hin, int win, int hout, int wout,
                                    int axis) {
    const int total = n * hout * wout;                      // < 2^31 (checked by the entry point)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ox = i % wout, r = i / wout, oy = r % hout, im = r / hout;
        const int o = axis == 0 ? oy : ox;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        const int e1 = rowptr[o + 1];
        for (int e = rowptr[o]; e < e1; ++e) {
            const int src = col[e];
            const float v = val[e];
            const float* p = in + (axis == 0 ? ((long)(im * hin + src) * win + ox) : ((long)(im * hin + oy) * win + src)) * 3;
            a0 = fmaf(v, p[0], a0);
            a1 = fmaf(v, p[1], a1);
            a2 = fmaf(v, p[2], a2);
        }
        float* q = out + (long)i * 3;
        q[0] = a0; q[1] = a1; q[2] = a2;
    }
}

// 3-channel images whose flat rows are whole float4s (win % 4 == 0), operator along the ROWS (axis 0): the output row is a
// weighted sum of whole input rows, so a thread owns one 16-byte granule of the flat output row - 16-byte loads and stores
// instead of three dwords at a 12-byte stride; the same fmaf chain per float (CSR order).  (Along the columns the per-pixel
// form stays: staging the input rows in LDS and gathering four pixels per thread ran at half its speed.)
__global__ void sparse_axis3_rows_kernel(const float4* __restrict__ in, float4* __restrict__ out, const int* __restrict__ rowptr,
                                         const int* __restrict__ col, const float* __restrict__ val, int n, int hin, int hout,
                                         int row4) {
    const int total = n * hout * row4;                      // < 2^31 (checked by the entry point)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int f = i % row4, r = i / row4, oy = r % hout, im = r / hout;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        const int e1 = rowptr[oy + 1];
        for (int e = rowptr[oy]; e < e1; ++e) {
            const float v = val[e];
            const float4 p = in[(long)(im * hin + col[e]) * row4 + f];
            a.x = fmaf(v, p.x, a.x); a.y = fmaf(v, p.y, a.y); a.z = fmaf(v, p.z, a.z); a.w = fmaf(v, p.w, a.w);
        }
        out[i] = a;
    }
}

// fold a gradient defined on the padded domain (h+2P, w+2P) back onto the image
__global__ void fold_pad_kernel(const float* __restrict__ dpad, float* __restrict__ dx, int n, int h, int w, int c,
                                int P, int mode) {
    const long total = (long)n * h * w * c;
    const int hp = h + 2 * P, wp = w + 2 * P;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        long r = i / c;
        const int px = (int)(r % w);
        r /= w;
        const int py = (int)(r % h);
        const long im = r / h;
        int ys[2], xs[2];
        const int ny = pad_sources(py, h, P, mode, ys), nx = pad_sources(px, w, P, mode, xs);
        float acc = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) acc += dpad[((im * hp + ys[a]) * wp + xs[b]) * c + ch];
        dx[i] = acc;
    }
}

__global__ void fold_pad3_kernel(const float* __restrict__ dpad, float* __restrict__ dx, int n, int h, int w, int P,
                                 int mode) {
    const int total = n * h * w;
    const int hp = h + 2 * P, wp = w + 2 * P;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int px = i % w, r = i / w, py = r % h, im = r / h;
        int ys[2], xs[2];
        const int ny = pad_sources(py, h, P, mode, ys), nx = pad_sources(px, w, P, mode, xs);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) {
                const float* p = dpad + ((long)(im * hp + ys[a]) * wp + xs[b]) * 3;
                a0 += p[0]; a1 += p[1]; a2 += p[2];
            }
        float* q = dx + (long)i * 3;
        q[0] = a0; q[1] = a1; q[2] = a2;
    }
}

// 2x2 (factor f) average pooling fwd / bwd  (workflows/manipulation_classification.py:235 'pool' downsampling)
__global__ void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int h, int w, int c,
                                   int f) {
    const int ho = h / f, wo = w / f;
    const long total = (long)n * ho * wo * c;
    const float inv = 1.0f / (float)(f * f);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        long r = i / c;
        const int ox = (int)(r % wo);
        r /= wo;
        const int oy = (int)(r % ho);
        const long im = r / ho;
        float s = 0.f;
        for (int a = 0; a < f; ++a)
            for (int b = 0; b < f; ++b) s += x[((im * h + oy * f + a) * w + ox * f + b) * c + ch];
        y[i] = s * inv;
    }
}
__global__ void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int n, int h, int w, int c,
                                   int f) {
    const int ho = h / f, wo = w / f;
    const long total = (long)n * h * w * c;
    const float inv = 1.0f / (float)(f * f);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        long r = i / c;
        const int px = (int)(r % w);
        r /= w;
        const int py = (int)(r % h);
        const long im = r / h;
        dx[i] = (py / f < ho && px / f < wo) ? inv * dy[((im * ho + py / f) * wo + px / f) * c + ch] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Any odd k x k per-channel (diagonal) filter with a mirrored border and an optional hard clip: manipulation_gaussian with a
// kernel other than 5 (tf_helpers.py:113-125, REFLECT), manipulation_sharpen(hsv=False) (:156-184, SYMMETRIC) and
// residual(hsv=False) (:127-154, REFLECT, no clip).  Off the training hot path (the workflow uses the 5x5 / hsv kernels
// above): one thread per pixel, taps in LDS, tap order (ky, then kx) as the 5x5 kernel.
__global__ void dwfilter_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ mask,
                                    const float* __restrict__ gk, int k, int mode, int n, int h, int w, int clip) {
    extern __shared__ float taps[];
    for (int i = threadIdx.x; i < k * k; i += blockDim.x) taps[i] = gk[i];
    __syncthreads();
    const long total = (long)n * h * w;
    const int P = k / 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int px = (int)(i % w), py = (int)((i / w) % h);
        const long im = i / ((long)w * h);
        float acc[3] = {0.f, 0.f, 0.f};
        for (int ky = 0; ky < k; ++ky) {
            int yy = py + ky - P;
            map_coord(yy, h, mode);
            for (int kx = 0; kx < k; ++kx) {
                int xx = px + kx - P;
                map_coord(xx, w, mode);
                const float* p = x + ((im * h + yy) * w + xx) * 3;
                const float wv = taps[ky * k + kx];
                acc[0] = fmaf(p[0], wv, acc[0]);
                acc[1] = fmaf(p[1], wv, acc[1]);
                acc[2] = fmaf(p[2], wv, acc[2]);
            }
        }
        uint32_t m = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            m |= (acc[c] >= 0.f && acc[c] <= 1.f) ? (1u << c) : 0u;
            y[i * 3 + c] = clip ? fminf(fmaxf(acc[c], 0.f), 1.f) : acc[c];
        }
        if (mask) mask[i] = clip ? (uint8_t)m : (uint8_t)7;
    }
}

__global__ void dwfilter_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ mask,
                                    float* __restrict__ dx, const float* __restrict__ gk, int k, int mode, int n, int h,
                                    int w) {
    extern __shared__ float taps[];
    for (int i = threadIdx.x; i < k * k; i += blockDim.x) taps[i] = gk[i];
    __syncthreads();
    const long total = (long)n * h * w;
    const int P = k / 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int px = (int)(i % w), py = (int)((i / w) % h);
        const long im = i / ((long)w * h);
        int ys[2], xs[2];
        const int ny = pad_sources(py, h, P, mode, ys), nx = pad_sources(px, w, P, mode, xs);
        float acc[3] = {0.f, 0.f, 0.f};
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b)
                for (int ky = 0; ky < k; ++ky) {
                    const int oy = ys[a] - ky;           // output row that read padded row ys[a] with tap ky
                    if (oy < 0 || oy >= h) continue;
                    for (int kx = 0; kx < k; ++kx) {
                        const int ox = xs[b] - kx;
                        if (ox < 0 || ox >= w) continue;
                        const long o = (im * h + oy) * w + ox;
                        const uint32_t m = mask ? mask[o] : 7u;
                        const float wv = taps[ky * k + kx];
                        const float* p = dy + o * 3;
                        if (m & 1u) acc[0] = fmaf(p[0], wv, acc[0]);
                        if (m & 2u) acc[1] = fmaf(p[1], wv, acc[1]);
                        if (m & 4u) acc[2] = fmaf(p[2], wv, acc[2]);
                    }
                }
        dx[i * 3 + 0] = acc[0];
        dx[i * 3 + 1] = acc[1];
        dx[i * 3 + 2] = acc[2];
    }
}

}  // namespace

extern "C" {

int nimg_gaussian_fwd(const float* x, float* y, uint8_t* mask, const float* gk25, int n, int h, int w, int clip,
                      void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!x || !y || !gk25 || n < 0 || h < 5 || w < 5) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    const int ty = (h + 15) / 16, tx = (w + 15) / 16;
    static const bool narrow = getenv("NIMG_GAUSS_NARROW") != nullptr;            // A/B switch: the 16 x 16 form
    if (!narrow && h % 16 == 0 && w % 64 == 0 && (long)n * h * w * 3 < (1L << 31))
        hipLaunchKernelGGL(gaussian_fwd_wide_kernel, dim3((unsigned)((long)n * ty * (w / 64))), dim3(256), 0, (hipStream_t)stream,
                           x, y, mask, gk25, n, h, w, clip, ty, w / 64);
    else if (h >= 16 && w >= 16 && (long)n * ty * tx < (1L << 31))
        hipLaunchKernelGGL(gaussian_fwd_tiled_kernel, dim3((unsigned)((long)n * ty * tx)), dim3(256), 0, (hipStream_t)stream,
                           x, y, mask, gk25, n, h, w, clip, ty, tx);
    else
        hipLaunchKernelGGL(gaussian_fwd_kernel, dim3(grid_for((long)n * h * w)), dim3(256), 0, (hipStream_t)stream, x, y,
                           mask, gk25, n, h, w, clip);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_gaussian_bwd(const float* dy, const uint8_t* mask, float* dx, const float* gk25, int n, int h, int w,
                      void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!dy || !dx || !gk25 || n < 0 || h < 5 || w < 5) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    const int ty = (h + 15) / 16, tx = (w + 15) / 16;
    static const bool narrow = getenv("NIMG_GAUSS_NARROW") != nullptr;
    if (!narrow && h % 16 == 0 && w % 64 == 0 && (long)n * h * w * 3 < (1L << 31))
        hipLaunchKernelGGL(gaussian_bwd_wide_kernel, dim3((unsigned)((long)n * ty * (w / 64))), dim3(256), 0, (hipStream_t)stream,
                           dy, mask, dx, gk25, n, h, w, ty, w / 64);
    else if (h >= 16 && w >= 16 && (long)n * ty * tx < (1L << 31))
        hipLaunchKernelGGL(gaussian_bwd_tiled_kernel, dim3((unsigned)((long)n * ty * tx)), dim3(256), 0,
                           (hipStream_t)stream, dy, mask, dx, gk25, n, h, w, ty, tx);
    else
        hipLaunchKernelGGL(gaussian_bwd_kernel, dim3(grid_for((long)n * h * w)), dim3(256), 0, (hipStream_t)stream, dy,
                           mask, dx, gk25, n, h, w);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_dwfilter_fwd(const float* x, float* y, uint8_t* mask, const float* taps, int k, int pad_mode, int n, int h, int w,
                      int clip, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    /* a mirrored border of k/2 pixels needs k/2 (SYMMETRIC) or k/2 + 1 (REFLECT) pixels to mirror */
    if (!x || !y || !taps || n < 0 || k < 1 || k > 31 || !(k & 1) || (pad_mode != 1 && pad_mode != 2)) return NIMG_ERR_ARG;
    if (h < k / 2 + (pad_mode == 2) || w < k / 2 + (pad_mode == 2) || h < 1 || w < 1) return NIMG_ERR_ARG;
    hipLaunchKernelGGL(dwfilter_fwd_kernel, dim3(grid_for((long)n * h * w)), dim3(256), (size_t)k * k * sizeof(float),
                       (hipStream_t)stream, x, y, mask, taps, k, pad_mode, n, h, w, clip);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_dwfilter_bwd(const float* dy, const uint8_t* mask, float* dx, const float* taps, int k, int pad_mode, int n, int h,
                      int w, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!dy || !dx || !taps || n < 0 || k < 1 || k > 31 || !(k & 1) || (pad_mode != 1 && pad_mode != 2)) return NIMG_ERR_ARG;
    /* the fold of the padded-domain gradient assumes at most ONE mirrored source per pixel and axis */
    if (h < 2 * (k / 2) + 1 || w < 2 * (k / 2) + 1) return NIMG_ERR_ARG;
    hipLaunchKernelGGL(dwfilter_bwd_kernel, dim3(grid_for((long)n * h * w)), dim3(256), (size_t)k * k * sizeof(float),
                       (hipStream_t)stream, dy, mask, dx, taps, k, pad_mode, n, h, w);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_sharpen_fwd(const float* x, float* y, float* aux_hsv, uint8_t* mask, const float* gk9, int n, int h, int w,
                     void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!x || !y || !gk9 || n < 0 || h < 3 || w < 3) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    const int ty = (h + 15) / 16, tx = (w + 15) / 16;
    if (h >= 16 && w >= 16 && (long)n * ty * tx < (1L << 31))
        hipLaunchKernelGGL(sharpen_fwd_tiled_kernel, dim3((unsigned)((long)n * ty * tx)), dim3(256), 0, (hipStream_t)stream,
                           x, y, aux_hsv, mask, gk9, n, h, w, ty, tx);
    else
        hipLaunchKernelGGL(sharpen_fwd_kernel, dim3(grid_for((long)n * h * w)), dim3(256), 0, (hipStream_t)stream, x, y,
                           aux_hsv, mask, gk9, n, h, w);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_sharpen_bwd(const float* x, const float* dy, float* aux_hsv, const uint8_t* mask, float* dx,
                     const float* gk9, int n, int h, int w, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!x || !dy || !aux_hsv || !mask || !dx || !gk9 || n < 0 || h < 3 || w < 3) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)n * h * w;
    hipLaunchKernelGGL(sharpen_bwd_a_kernel, dim3(grid_for(total)), dim3(256), 0, s, dy, mask, aux_hsv, total);
    NIMG_CHECK_LAUNCH();
    const int ty = (h + 15) / 16, tx = (w + 15) / 16;
    if (h >= 16 && w >= 16 && (long)n * ty * tx < (1L << 31))
        hipLaunchKernelGGL(sharpen_bwd_b_tiled_kernel, dim3((unsigned)((long)n * ty * tx)), dim3(256), 0, s, x,
                           (const float*)aux_hsv, dx, gk9, n, h, w, ty, tx);
    else
        hipLaunchKernelGGL(sharpen_bwd_b_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, (const float*)aux_hsv, dx,
                           gk9, n, h, w);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_sparse_axis_apply(const float* in, float* out, const int* rowptr, const int* col, const float* val, int n,
                           int hin, int win, int c, int axis, int out_size, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!in || !out || !rowptr || !col || !val || n < 0 || hin <= 0 || win <= 0 || c <= 0 || out_size <= 0)
        return NIMG_ERR_ARG;
    if (axis != 0 && axis != 1) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    const int hout = axis == 0 ? out_size : hin, wout = axis == 1 ? out_size : win;
    static const bool scalar_form = getenv("NIMG_SPARSE_AXIS_SCALAR") != nullptr;            // A/B switch
    const bool small = (long)n * hout * wout < (1L << 29) && (long)n * hin * win < (1L << 29);
    if (c == 3 && !scalar_form && small && axis == 0 && win % 4 == 0) {
        const int row4 = win * 3 / 4;
        hipLaunchKernelGGL(sparse_axis3_rows_kernel, dim3(grid_for((long)n * hout * row4)), dim3(256), 0, (hipStream_t)stream,
                           (const float4*)in, (float4*)out, rowptr, col, val, n, hin, hout, row4);
        NIMG_CHECK_LAUNCH();
        return NIMG_OK;
    }
    if (c == 3 && (long)n * hout * wout < (1L << 31) && (long)n * hin * win < (1L << 31))
        hipLaunchKernelGGL(sparse_axis3_kernel, dim3(grid_for((long)n * hout * wout)), dim3(256), 0, (hipStream_t)stream,
                           in, out, rowptr, col, val, n, hin, win, hout, wout, axis);
    else
        hipLaunchKernelGGL(sparse_axis_kernel, dim3(grid_for((long)n * hout * wout * c)), dim3(256), 0,
                           (hipStream_t)stream, in, out, rowptr, col, val, n, hin, win, hout, wout, c, axis);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_fold_pad(const float* dpad, float* dx, int n, int h, int w, int c, int pad, int pad_mode, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!dpad || !dx || n < 0 || h <= 2 * pad || w <= 2 * pad || c <= 0 || pad < 0 || pad_mode < 1 || pad_mode > 2)
        return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    if (c == 3 && (long)n * (h + 2 * pad) * (w + 2 * pad) < (1L << 31))
        hipLaunchKernelGGL(fold_pad3_kernel, dim3(grid_for((long)n * h * w)), dim3(256), 0, (hipStream_t)stream, dpad, dx, n,
                           h, w, pad, pad_mode);
    else
        hipLaunchKernelGGL(fold_pad_kernel, dim3(grid_for((long)n * h * w * c)), dim3(256), 0, (hipStream_t)stream, dpad,
                           dx, n, h, w, c, pad, pad_mode);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_avgpool_fwd(const float* x, float* y, int n, int h, int w, int c, int factor, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!x || !y || n < 0 || h <= 0 || w <= 0 || c <= 0 || factor < 1 || h % factor || w % factor) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_for((long)n * (h / factor) * (w / factor) * c)), dim3(256), 0,
                       (hipStream_t)stream, x, y, n, h, w, c, factor);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_avgpool_bwd(const float* dy, float* dx, int n, int h, int w, int c, int factor, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!dy || !dx || n < 0 || h <= 0 || w <= 0 || c <= 0 || factor < 1 || h % factor || w % factor) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for((long)n * h * w * c)), dim3(256), 0, (hipStream_t)stream, dy,
                       dx, n, h, w, c, factor);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// awgn / gamma / median manipulations (helpers/tf_helpers.py:79-110); soft_quantization = round in the forward pass,
// 1 - cos(2 pi x) in the backward pass (tf_helpers.py:271-277).
namespace {

// y = clip(softq(x + s * noise), 0, 1);   element-wise, mask byte per ELEMENT (1 = gradient passes the clip)
__global__ void awgn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ noise, float* __restrict__ y,
                                uint8_t* __restrict__ mask, long count, float strength) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float q = rintf(255.0f * (x[i] + strength * noise[i])) / 255.0f;
        y[i] = fminf(fmaxf(q, 0.f), 1.f);
        if (mask) mask[i] = (q >= 0.f && q <= 1.f) ? 1 : 0;
    }
}
__global__ void awgn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                const float* __restrict__ dy, const uint8_t* __restrict__ mask,
                                float* __restrict__ dx, long count, float strength) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float v = 255.0f * (x[i] + strength * noise[i]);
        const float d = 1.0f - __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(v));      // cos(2 pi v)
        dx[i] = mask[i] ? dy[i] * d : 0.f;
    }
}

// y = pow(clip(softq(pow(x, g)), 1/255, 1), 1/g)
__global__ void gamma_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long count, float g) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float p = powf(x[i], g);
        const float q = rintf(255.0f * p) / 255.0f;
        y[i] = powf(fminf(fmaxf(q, 1.0f / 255.0f), 1.f), 1.0f / g);
    }
}
__global__ void gamma_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                 long count, float g) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float xv = x[i];
        const float p = powf(xv, g);
        const float q = rintf(255.0f * p) / 255.0f;
        const float c = fminf(fmaxf(q, 1.0f / 255.0f), 1.f);
        float grad = dy[i] * (1.0f / g) * powf(c, 1.0f / g - 1.0f);                    // d pow(c, 1/g)
        grad = (q >= 1.0f / 255.0f && q <= 1.f) ? grad : 0.f;                          // clip
        grad *= 1.0f - __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(255.0f * p));     // soft quantisation
        dx[i] = grad * g * powf(xv, g - 1.0f);                                         // d pow(x, g)
    }
}

// k x k median (k odd <= 9), REFLECT pad; sel (optional) = index (0..k*k-1, row-major in the window) of the element that
// was selected, so that the backward can route the gradient to it like tf.nn.top_k's gradient does.
__global__ void median_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ sel, int n,
                                  int h, int w, int k) {
    const long total = (long)n * h * w * 3;
    const int area = k * k, r = k / 2, rank = (area + 1) / 2 - 1;          // descending order, index floor-1
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % 3);
        long t = i / 3;
        const int px = (int)(t % w);
        t /= w;
        const int py = (int)(t % h);
        const long im = t / h;
        float v[81];
        for (int a = 0; a < k; ++a) {
            int yy = py + a - r;
            map_coord(yy, h, 2);
            for (int b = 0; b < k; ++b) {
                int xx = px + b - r;
                map_coord(xx, w, 2);
                v[a * k + b] = x[((im * h + yy) * w + xx) * 3 + ch];
            }
        }
        // the element with exactly `rank` strictly-greater-or-earlier-equal elements (stable descending order)
        int pick = 0;
        for (int a = 0; a < area; ++a) {
            int before = 0;
            for (int b = 0; b < area; ++b) before += (v[b] > v[a]) || (v[b] == v[a] && b < a);
            if (before == rank) pick = a;
        }
        y[i] = v[pick];
        if (sel) sel[i] = (uint8_t)pick;
    }
}
__global__ void median_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ sel,
                                  float* __restrict__ dx, int n, int h, int w, int k) {
    const long total = (long)n * h * w * 3;
    const int r = k / 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % 3);
        long t = i / 3;
        const int px = (int)(t % w);
        t /= w;
        const int py = (int)(t % h);
        const long im = t / h;
        const int s = sel[i];
        int yy = py + s / k - r, xx = px + s % k - r;
        map_coord(yy, h, 2);
        map_coord(xx, w, 2);
        atomicAdd(dx + ((im * h + yy) * w + xx) * 3 + ch, dy[i]);
    }
}

}  // namespace

extern "C" {

int nimg_awgn_fwd(const float* x, const float* noise, float* y, uint8_t* mask, long count, float strength,
                  void* stream) {
    if (!x || !noise || !y || count < 0) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    hipLaunchKernelGGL(awgn_fwd_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, x, noise, y, mask,
                       count, strength);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}
int nimg_awgn_bwd(const float* x, const float* noise, const float* dy, const uint8_t* mask, float* dx, long count,
                  float strength, void* stream) {
    if (!x || !noise || !dy || !mask || !dx || count < 0) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    hipLaunchKernelGGL(awgn_bwd_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, x, noise, dy, mask,
                       dx, count, strength);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}
int nimg_gamma_fwd(const float* x, float* y, long count, float gamma, void* stream) {
    if (!x || !y || count < 0 || gamma <= 0.f) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    hipLaunchKernelGGL(gamma_fwd_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, x, y, count, gamma);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}
int nimg_gamma_bwd(const float* x, const float* dy, float* dx, long count, float gamma, void* stream) {
    if (!x || !dy || !dx || count < 0 || gamma <= 0.f) return NIMG_ERR_ARG;
    if (count == 0) return NIMG_OK;
    hipLaunchKernelGGL(gamma_bwd_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, count,
                       gamma);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}
int nimg_median_fwd(const float* x, float* y, uint8_t* sel, int n, int h, int w, int kernel, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!x || !y || n < 0 || kernel < 1 || kernel > 9 || !(kernel & 1) || h <= kernel / 2 || w <= kernel / 2)
        return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    hipLaunchKernelGGL(median_fwd_kernel, dim3(grid_for((long)n * h * w * 3)), dim3(256), 0, (hipStream_t)stream, x, y,
                       sel, n, h, w, kernel);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}
/* dx must be zero-initialised by the caller (gradients are scattered with atomic adds) */
int nimg_median_bwd(const float* dy, const uint8_t* sel, float* dx, int n, int h, int w, int kernel, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!dy || !sel || !dx || n < 0 || kernel < 1 || kernel > 9 || !(kernel & 1)) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    hipLaunchKernelGGL(median_bwd_kernel, dim3(grid_for((long)n * h * w * 3)), dim3(256), 0, (hipStream_t)stream, dy,
                       sel, dx, n, h, w, kernel);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

}  // extern "C"
