// Throughput mode of the convolutions: bf16 operands on the CDNA4 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the f32
// MFMA rate), float32 accumulation, float32 master weights and float32 activations in HBM (converted to bf16 while the
// tiles are staged into LDS).  Same im2col-free implicit GEMM as conv_mfma.hip / conv_wgrad.hip; judged on PSNR /
// accuracy parity, not on the 1e-4 contract (that is the f32 mode).
//
//   forward / input gradient: A tile [pixel][16 ci] bf16 (32 B per pixel, one ds_read_b128 per fragment, 1-bit XOR
//       swizzle of the two 16-byte halves => conflict-free), B tile [tap][co][16 ci] bf16 read the same way from weights
//       that nimg_conv_weights_bf16 lays out once per step as [ci chunk][tap][co][16].
//   weight gradient: K = 16 output pixels per MFMA; the f32 NHWC tiles of conv_wgrad.hip are kept and each lane gathers
//       its 8 pixels with ds_read_b32 (conflict-free) and packs them to bf16 in registers (v_cvt_pk_bf16_f32).
#include <stdlib.h>

#include <cstdlib>
#include "common.h"

// defined in wgrad5.hip: slabs written (0 = not its shape, -1 = launch error)
int nimg_internal_wgrad5_alltaps(const void* in, int cin, const void* g, const unsigned char* idx, int cout, float* partial,
                                 float* db_partial, int n, int h, int wd, int max_slabs, hipStream_t stream);
// defined in wgrad3.hip: same contract for the UNet's 3x3 layers (bf16 input(s) and output gradient)
int nimg_internal_wgrad3_alltaps(const void* in1, int c1, const void* in2, int c2, const void* dz, int cout, float* partial,
                                 float* db_partial, int n, int h, int wd, int max_slabs, hipStream_t stream, float* dw, float* db,
                                 int accumulate, const void* pre);
// defined in conv_small.hip
size_t nimg_internal_wgrad_tiny_bytes(int ks, int cin, int cout);
int nimg_internal_conv_wgrad_tiny(const float* in, const float* dz, float* dw, int cin, int cout, int n, int h, int wd,
                                  int ks, int pad, int pad_mode, int accumulate, void* workspace, hipStream_t s, bool bf16_ok);


namespace {

using namespace nimg;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 pack8(const float (&f)[8]) {
    bf16x8 r;
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = (__bf16)f[k];
    return r;
}

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store4_bf16(float* base_as_bf16, long elem, float4 v) {
    bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(base_as_bf16) + elem) = o;
}
__device__ __forceinline__ float4 load4_bf16(const float* base_as_bf16, long elem) {
    const bf16x4 o = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(base_as_bf16) + elem);
    return make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
}

// wb[chunk][tap'][co][16] (mode 0, forward) = w[tap][ci = 16*chunk + k][co];  mode 1 (input gradient): roles of ci/co
// swap and the taps are flipped.  Chunk-major, so the weight tile a workgroup stages per 16-channel K chunk is one
// contiguous 2 KiB run per tap (fully coalesced 16-byte loads).  Padding channels are zero.
__global__ void weights_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wb, int taps, int cin, int cout,
                                    int mode) {
    const int rows = mode == 0 ? cout : cin, cols = mode == 0 ? cin : cout;
    const int cpad = (cols + 15) / 16 * 16;
    const long total = (long)taps * rows * cpad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % 16) + 16 * (int)(i / (16L * rows * taps)), r = (int)((i / 16) % rows);
        const int t = (int)((i / (16L * rows)) % taps);
        float v = 0.f;
        if (c < cols) v = mode == 0 ? w[((long)t * cin + c) * cout + r] : w[((long)(taps - 1 - t) * cin + r) * cout + c];
        wb[i] = (__bf16)v;
    }
}

// All bf16 weight images of a model in ONE launch (blockIdx.y = table entry): the per-layer launches were 54 kernels of
// ~5 us per training step.  Table entry = 4 x int64: {w pointer, wb pointer, (taps << 32) | mode, (cin << 32) | cout}.
__global__ __launch_bounds__(256) void weights_bf16_batch_kernel(const long long* __restrict__ table) {
    const long long* e = table + 4 * blockIdx.y;
    const float* w = reinterpret_cast<const float*>(e[0]);
    __bf16* wb = reinterpret_cast<__bf16*>(e[1]);
    const int taps = (int)(e[2] >> 32), mode = (int)(e[2] & 0xffffffffll);
    const int cin = (int)(e[3] >> 32), cout = (int)(e[3] & 0xffffffffll);
    const int rows = mode == 0 ? cout : cin, cols = mode == 0 ? cin : cout;
    const int chunks = (cols + 15) / 16, rblocks = (rows + 63) / 64;
    const int ntiles = chunks * taps * rblocks;            // tile = one (chunk, tap) x 64 rows x 16 columns
    __shared__ float tile[16][65];
    const int tid = threadIdx.x;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int rb = tl % rblocks, ct = tl / rblocks, t = ct % taps, chunk = ct / taps;
        const int r0 = rb * 64, c0 = chunk * 16;
        __syncthreads();
        if (mode == 0) {       // w[(t*cin + c)*cout + r]: r is the contiguous axis -> 16 rows of 64 floats, transposed via LDS
            const int cl = tid >> 4, rq = (tid & 15) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + cl, r = r0 + rq + k;
                tile[cl][rq + k] = (c < cols && r < rows) ? w[((long)t * cin + c) * cout + r] : 0.f;
            }
        } else {               // w[((taps-1-t)*cin + r)*cout + c]: c is contiguous
            const int rl = tid >> 2, cq = (tid & 3) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + cq + k, r = r0 + rl;
                tile[cq + k][rl] = (c < cols && r < rows) ? w[((long)(taps - 1 - t) * cin + r) * cout + c] : 0.f;
            }
        }
        __syncthreads();
        const int rl = tid >> 2, kq = (tid & 3) * 4;
        if (r0 + rl < rows) {
            bf16x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (__bf16)tile[kq + k][rl];
            *reinterpret_cast<bf16x4*>(wb + ((long)ct * rows + r0 + rl) * 16 + kq) = o;
        }
    }
}

struct ConvParamsB {
    const float* in1;
    const float* in2;
    const __bf16* wb;     // [CinP/16][KS*KS][Cout][16]
    const float* bias;
    float* out1;
    float* out2;
    const float* act1;
    float* pool_out;            // optional fused activation + 2x2 max-pool output (replaces out1), see common.h
    unsigned char* pool_idx;
    int C1, C2, O1, O2, CinP;
    int N, H, W, Hout, Wout, pad_t, pad_l;
    int tiles_y, tiles_x, act, pad_mode;
    float alpha;
    int convt;            // 1: Conv2DTranspose(2x2, stride 2) as four 1x1 products; workgroup id & 3 = output phase (dy, dx)
    int flags;            // NIMG_BF16_IN: in1 (and in2) hold bf16; _OUT: out1 / out2 / pool_out are bf16; _MASK: act1 is bf16
    const unsigned char* in_idx;   // UNP kernels: in1 is the POOLED tensor (N, H/2, W/2, C1) bf16 and in_idx its arg-max bytes;
                                   // the convolution runs on their 2x2 un-pooling (H x W), built while staging
    const float* res;              // optional float32 tensor of out1's shape added to the result (after bias, activation and mask):
                                   // the skip connection of a residual block, forward and backward (3x3, float32 output)
    float* out1b;                  // optional SECOND copy of out1 as bf16 (float32 out1 only): the running float32 sum of a residual
                                   // stream stays exact while its consumers - convolutions and weight gradients, which round to
                                   // bf16 anyway - read half the bytes through the bf16-input kernels
};

// INB: in1 is stored as bf16 (compile-time: a run-time branch around the prefetch loads makes the backend wait for them at
// the join, which serialises the staging latency the async-stage split hides)
// BUF (with INB, one input tensor, Cin % 16 == 0, Cout % TN == 0, tensors < 2 GB): the prefetch goes through buffer
// descriptors - per-lane byte offsets resolved once per tile, the channel chunk in the scalar offset, padding pixels as
// out-of-range offsets that the hardware answers with zeros.  hipcc wraps every predicated flat load in s_and_saveexec /
// branch / zero-fill / 64-bit address arithmetic (~9 instructions x 17 loads per thread and chunk, issued in front of the
// MFMA loop); a buffer load is one instruction.
// UNP (with BUF): in1 is a POOLED gradient + arg-max bytes; a halo pixel (y, x) reads pooled pixel (y/2, x/2) and keeps channel
// c iff argmax == 2 (y & 1) + (x & 1) - the MaxPool2D routing as a packed byte-mask operation on the staged 16 bytes, so the
// full-resolution gradient (4x the bytes, 3/4 of them zeros) is never written to HBM nor read back.
__device__ __forceinline__ unsigned unp_eq_bytes(unsigned k, unsigned pos) {          // 0xFF in every byte of k equal to pos
    const unsigned x = k ^ (pos * 0x01010101u);
    return (((x | (x >> 1)) & 0x01010101u) ^ 0x01010101u) * 0xFFu;
}
__device__ __forceinline__ uint4 unp_route(uint4 g, unsigned k0, unsigned k1, unsigned pos) {
    const unsigned m0 = unp_eq_bytes(k0, pos), m1 = unp_eq_bytes(k1, pos);
    return make_uint4(g.x & __builtin_amdgcn_perm(m0, m0, 0x01010000u), g.y & __builtin_amdgcn_perm(m0, m0, 0x03030202u),
                      g.z & __builtin_amdgcn_perm(m1, m1, 0x01010000u), g.w & __builtin_amdgcn_perm(m1, m1, 0x03030202u));
}

// Vector epilogue shared by the convolution kernels: the accumulators of a wave (MI x NI fragments of 32 pixels x 32 channels)
// are turned around through the wave's LDS scratch so that each lane stores 16 B along the NHWC channel axis; bias, activation,
// previous-layer LeakyReLU' mask, residual, bf16 copy and the depth_to_space / space_to_depth output layouts are applied on the
// way (ConvParamsB).  Requires O1 % 4 == 0 and O2 % 4 == 0; contains one workgroup barrier (the scratch aliases the tiles).
// The eight biases a lane adds in the 8-wide epilogue below are the same on every call (channels 8 (lane % (4 NI)) .. + 7 of the
// wave's strip): the kernels request them at their START (EpiBias), so the epilogue of a workgroup that has its SIMDs to itself
// does not open with a memory round trip (conv3_rows.hip: that round trip was 20 % of a byte-bound layer).
struct EpiBias { float b[8]; };
template <int NI>
__device__ __forceinline__ EpiBias epi_bias_preload(const ConvParamsB& p, int lane, int wn, int co0, int Cout) {
    EpiBias r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r.b[e] = 0.f;
    const int co = co0 + wn * NI * 32 + (lane % (NI * 4)) * 8;
    if (p.bias && co + 7 < Cout) {
        const int cb = (p.flags & NIMG_D2S_CONVT) ? co % (p.O1 >> 2) : co;
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + cb), b1 = *reinterpret_cast<const float4*>(p.bias + cb + 4);
        r.b[0] = b0.x; r.b[1] = b0.y; r.b[2] = b0.z; r.b[3] = b0.w; r.b[4] = b1.x; r.b[5] = b1.y; r.b[6] = b1.z; r.b[7] = b1.w;
    }
    return r;
}

// Which tile pixel row P of the workgroup's M dimension is (fragment P >> 5, lane P & 31).  PMAP 0: row-major over (image, y, x) -
// a fragment = 32 consecutive tile pixels (two rows of a 16-wide tile).  PMAP 1 (conv3_dma_kernel's plane layout): a 16 x 16
// tile's fragment = 8 columns x 4 rows (fragment = column half + 2 x row group), an 8 x 8 x 4-image tile's fragment = ONE row
// of all four images (lane = 8 image + x) - the maps whose 16-lane ds_read_b128 groups meet 16 different bank quads.
template <int TH, int TW, int NB, int PMAP>
__device__ __forceinline__ void tile_pixel(int P, int& img, int& dy, int& dx) {
    if constexpr (PMAP == 0) {
        img = P / (TH * TW);
        const int rem = P % (TH * TW);
        dy = rem / TW;
        dx = rem % TW;
    } else if constexpr (NB == 1) {
        static_assert(TH == 16 && TW == 16, "plane layout: 16 x 16 tile");
        const int f = P >> 5, l = P & 31;
        img = 0;
        dy = 4 * (f >> 1) + (l >> 3);
        dx = 8 * (f & 1) + (l & 7);
    } else {
        static_assert(TH == 8 && TW == 8 && NB == 4, "plane layout: 8 x 8 x 4 tile");
        img = (P & 31) >> 3;
        dy = P >> 5;
        dx = P & 7;
    }
}

template <int KS, int TH, int TW, int NB, int MI, int NI, int PMAP = 0>
__device__ __forceinline__ void conv_epilogue_vec(const f32x16 (&acc)[MI][NI], const ConvParamsB& p, unsigned char* smem_raw,
                                                  int wave, int lane, int wm, int wn, int co0, int Cout, int ty0, int tx0,
                                                  int grp, int phase, const EpiBias& pre_) {
#ifdef NIMG_NO_EPI_PRELOAD                      // A/B: the biases requested where the epilogue starts, as before round 5
    const EpiBias pre = epi_bias_preload<NI>(p, lane, wn, co0, Cout);
    (void)pre_;
#else
    const EpiBias& pre = pre_;
#endif
    float* elds = reinterpret_cast<float*>(smem_raw) + wave * (32 * (NI * 32 + EPI_PAD));
    __syncthreads();                    // the scratch aliases the tiles: everyone is done reading them; from here on every
                                        // wave works in its own region (wave-level ordering only)
#ifndef NIMG_NO_EPI8
    // bf16-stored outputs (and mask) in the plain layout - the UNet's and the codec's inner layers: eight channels per lane,
    // 16-byte stores / mask loads (the store-issue rate, not the bytes, bounds a row-per-lane epilogue)
    // NIMG_UNPOOL_OUT: the result is the gradient of a 2x2 max-pool's OUTPUT (the UNet's encoder levels, pipelines.py:160-173
    // backward): every value goes to the first maximum of its window of the stored activation act1 (n, 2 hout, 2 wout, o1), the
    // skip gradient `res` (same shape, bf16, optional; may be out1 itself) is added to all four positions, LeakyReLU'(act1)
    // applied (act == 1) - maxpool2_bwd_bf16_kernel's arithmetic on the value this kernel would have stored as bf16
    if (p.flags & NIMG_UNPOOL_OUT) {
        const __bf16* ya = reinterpret_cast<const __bf16*>(p.act1);
        const __bf16* sk = reinterpret_cast<const __bf16*>(p.res);
        __bf16* dzo = reinterpret_cast<__bf16*>(p.out1);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            epilogue_via_lds8<NI>(acc[mi], elds, lane, [&](int row, int c, float4 lo, float4 hi) {
                const int co = co0 + wn * NI * 32 + c;
                if (co >= Cout) return;
                const int P = (wm * MI + mi) * 32 + row;
                int img, dy_, dx_;
                tile_pixel<TH, TW, NB, PMAP>(P, img, dy_, dx_);
                const int oy = ty0 + dy_, ox = tx0 + dx_, n = grp * NB + img;
                if (n >= p.N || oy >= p.Hout || ox >= p.Wout) return;
                const float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                const long W2 = 2L * p.Wout;
                const long base = (((long)n * 2 * p.Hout + 2 * oy) * W2 + 2 * ox) * p.O1 + co;
                const long offs[4] = {0, (long)p.O1, W2 * p.O1, W2 * p.O1 + p.O1};
                bf16x8 v[4], a[4], o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = *reinterpret_cast<const bf16x8*>(ya + base + offs[q]);
                    if (sk) a[q] = *reinterpret_cast<const bf16x8*>(sk + base + offs[q]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float g = (float)(__bf16)f[e];
                    const float v0 = (float)v[0][e], v1 = (float)v[1][e], v2 = (float)v[2][e], v3 = (float)v[3][e];
                    const float m = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
                    const int sel = v0 == m ? 0 : (v1 == m ? 1 : (v2 == m ? 2 : 3));
                    const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float t = (q == sel) ? g : 0.f;
                        if (sk) t += (float)a[q][e];
                        if (p.act == 1) t *= (vv[q] > 0.f ? 1.0f : p.alpha);
                        o[q][e] = (__bf16)t;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<bf16x8*>(dzo + base + offs[q]) = o[q];
            });
        }
        return;
    }
    // NIMG_D2S_CONVT: Conv2DTranspose(2x2, stride 2) as one 1x1 product over 4 x cout columns (nimg_convt2x2_fwd_bf16_ex) - column
    // block b holds output phase 3 - b (the weight image lists the taps flipped), which goes to pixel (2 y + dy, 2 x + dx)
    if (p.flags & NIMG_D2S_CONVT) {
        const int cd = p.O1 >> 2;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            epilogue_via_lds8<NI>(acc[mi], elds, lane, [&](int row, int c, float4 lo, float4 hi) {
                const int co = co0 + wn * NI * 32 + c;
                if (co >= Cout) return;
                const int P = (wm * MI + mi) * 32 + row;
                int img, dy_, dx_;
                tile_pixel<TH, TW, NB, PMAP>(P, img, dy_, dx_);
                const int oy = ty0 + dy_, ox = tx0 + dx_, n = grp * NB + img;
                if (n >= p.N || oy >= p.Hout || ox >= p.Wout) return;
                const int blk = co / cd, cc = co - blk * cd, ph = 3 - blk;
                const long o = (((long)n * 2 * p.Hout + 2 * oy + (ph >> 1)) * (2 * p.Wout) + 2 * ox + (ph & 1)) * cd + cc;
                float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if (p.bias) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += pre.b[e];
                }
                *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.out1) + o) = pack8(f);
            });
        }
        return;
    }
    if ((p.flags & NIMG_BF16_OUT) && !(p.flags & (NIMG_D2S_OUT | NIMG_S2D_OUT)) && !p.res && !p.out1b &&
        (!p.act1 || (p.flags & NIMG_BF16_MASK)) && (p.O1 & 7) == 0 && (p.O2 & 7) == 0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            epilogue_via_lds8<NI>(acc[mi], elds, lane, [&](int row, int c, float4 lo, float4 hi) {
                const int co = co0 + wn * NI * 32 + c;
                if (co >= Cout) return;
                const int P = (wm * MI + mi) * 32 + row;
                int img, dy_, dx_;
                tile_pixel<TH, TW, NB, PMAP>(P, img, dy_, dx_);
                const int oy = ty0 + dy_, ox = tx0 + dx_, n = grp * NB + img;
                if (n >= p.N || oy >= p.Hout || ox >= p.Wout) return;
                const long pixoff = (KS == 1 && p.convt)
                    ? ((long)n * 2 * p.Hout + 2 * oy + (phase >> 1)) * (2 * p.Wout) + 2 * ox + (phase & 1)
                    : ((long)n * p.Hout + oy) * p.Wout + ox;
                float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if (p.bias) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += pre.b[e];
                }
                if (p.act == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = lrelu(f[e], p.alpha);
                }
                if (co < p.O1) {
                    const long o = pixoff * p.O1 + co;
                    if (p.act1) {
                        const bf16x8 m = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(p.act1) + o);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] *= (float)m[e] > 0.f ? 1.0f : p.alpha;
                    }
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.out1) + o) = pack8(f);
                } else {
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.out2) + pixoff * p.O2 + (co - p.O1)) = pack8(f);
                }
            });
            // NIMG_POOL_ALSO: the 2x2 max-pooled tensor next to the full one (the UNet's encoder keeps the full tensor for its skip
            // connection and feeds the pooled one to the next level): a fragment = two rows of 16 pixels = 8 windows per channel
            if constexpr (TW == 16 && NB == 1 && PMAP == 0)
                if (p.pool_out) {
                    const int Hp = p.Hout >> 1, Wp = p.Wout >> 1, py = (ty0 >> 1) + wm * MI + mi;
                    pool_in_regs8<NI>(acc[mi], elds, lane, p.act == 1 ? p.alpha : 1.0f,
                        [&](int c) { const int co = co0 + wn * NI * 32 + c; return (p.bias && co < Cout) ? p.bias[co] : 0.f; },
                        [&](int pc, int c, float4 lo, float4 hi, uint2 k) {
                            const int co = co0 + wn * NI * 32 + c, px = (tx0 >> 1) + pc;
                            if (co >= Cout || grp >= p.N || py >= Hp || px >= Wp) return;
                            const long o = (((long)grp * Hp + py) * Wp + px) * Cout + co;
                            const float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                            *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.pool_out) + o) = pack8(f);
                            if (p.pool_idx) *reinterpret_cast<uint2*>(p.pool_idx + o) = k;
                        });
                }
        }
        return;
    }
#endif
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        epilogue_via_lds<NI, false>(acc[mi], elds, lane, [&](int row, int c, float4 v) {
            const int co = co0 + wn * NI * 32 + c;
            if (co >= Cout) return;
            const int P = (wm * MI + mi) * 32 + row;
            int img, dy_, dx_;
            tile_pixel<TH, TW, NB, PMAP>(P, img, dy_, dx_);
            const int oy = ty0 + dy_, ox = tx0 + dx_, n = grp * NB + img;
            if (n >= p.N || oy >= p.Hout || ox >= p.Wout) return;
            const long pixoff = (KS == 1 && p.convt)
                ? ((long)n * 2 * p.Hout + 2 * oy + (phase >> 1)) * (2 * p.Wout) + 2 * ox + (phase & 1)
                : ((long)n * p.Hout + oy) * p.Wout + ox;
            if (p.bias) {
                const float4 b = *reinterpret_cast<const float4*>(p.bias + co);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            if (p.act == 1) {
                v.x = lrelu(v.x, p.alpha); v.y = lrelu(v.y, p.alpha); v.z = lrelu(v.z, p.alpha); v.w = lrelu(v.w, p.alpha);
            }
            if (co < p.O1) {
                long o = pixoff * p.O1 + co;
                const long om_conv = o;            // NIMG_MASK_CONV: the mask keeps the convolution's own layout
                if (p.flags & NIMG_D2S_OUT) {      // depth_to_space(2): channel block (2 dy + dx) of pixel (oy, ox) is pixel
                    const int cd = p.O1 >> 2, blk = co / cd;                 // (2 oy + dy, 2 ox + dx) of the output
                    o = (((long)n * 2 * p.Hout + 2 * oy + (blk >> 1)) * (2 * p.Wout) + 2 * ox + (blk & 1)) * cd + (co - blk * cd);
                }
                if (p.act1) {
                    const long om = (p.flags & NIMG_MASK_CONV) ? om_conv : o;
                    const float4 m = (p.flags & NIMG_BF16_MASK) ? load4_bf16(p.act1, om)
                                                                : *reinterpret_cast<const float4*>(p.act1 + om);
                    v.x *= m.x > 0.f ? 1.0f : p.alpha; v.y *= m.y > 0.f ? 1.0f : p.alpha;
                    v.z *= m.z > 0.f ? 1.0f : p.alpha; v.w *= m.w > 0.f ? 1.0f : p.alpha;
                }
                if (p.res) {
                    const float4 r = *reinterpret_cast<const float4*>(p.res + o);
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                if (p.flags & NIMG_S2D_OUT)        // space_to_depth(2): pixel (oy, ox) is channel block 2 (oy & 1) + (ox & 1) of
                    o = (((long)n * (p.Hout >> 1) + (oy >> 1)) * (p.Wout >> 1) + (ox >> 1)) * (4 * p.O1) +    // pixel (oy/2, ox/2);
                        (2 * (oy & 1) + (ox & 1)) * p.O1 + co;                  // mask and residual keep the convolution's layout
                if (p.out1b) {
                    float4 c = v;
                    if (p.flags & NIMG_COPY_LRELU) { c.x = lrelu(c.x, p.alpha); c.y = lrelu(c.y, p.alpha); c.z = lrelu(c.z, p.alpha); c.w = lrelu(c.w, p.alpha); }
                    store4_bf16(p.out1b, o, c);
                }
                if (p.flags & NIMG_BF16_OUT) store4_bf16(p.out1, o, v);
                else *reinterpret_cast<float4*>(p.out1 + o) = v;
            } else {
                if (p.flags & NIMG_BF16_OUT) store4_bf16(p.out2, pixoff * p.O2 + (co - p.O1), v);
                else *reinterpret_cast<float4*>(p.out2 + pixoff * p.O2 + (co - p.O1)) = v;
            }
        });
    }
}

template <int KS, int STRIDE, int TH, int TW, int NB, int TN, bool INB, bool BUF = false, bool UNP = false, int CKT = 16>
__global__ __launch_bounds__(256) void conv_fwd_bf16_kernel(const ConvParamsB p) {
    // K chunk: 16 channels (one MFMA k-step per tap).  -DNIMG_CK32 stages 32 channels (two k-steps, half the barriers) for the
    // 3x3 layers with float32 tensors (UNet / TwitterDCN) - measured on the bench step: the 8x8 bottleneck variant gains 12 %
    // (43 -> 38 us) but the 16x16 variants LOSE 4-22 % (47 -> 49 / 58 us: twice the LDS per workgroup and 84 staging registers
    // cost more than the halved barrier count returns), -5 % on the whole step - so it stays an experiment switch.
    // CKT = 64 (1x1 kernels with Cin % 64 == 0: FAN conv5, the UNet's Conv2DTranspose): a 1x1 layer has ONE tap per chunk, i.e.
    // 4 MFMAs per wave between two barriers with 16-channel chunks; 64 channels make it 16.
#ifdef NIMG_CK32
    constexpr int CK = CKT != 16 ? CKT : ((KS == 3 && !INB && !BUF) ? 32 : 16);
#else
    constexpr int CK = CKT;
#endif
    static_assert(CK == 16 || CK == 32 || (CK == 64 && KS == 1 && !BUF), "K chunk");
    constexpr int CKH = CK / 8;                          // 16-byte slots (8 channels) per pixel / weight row
    static_assert(!UNP || (INB && BUF && STRIDE == 1), "un-pooling input: bf16 buffer-load path only");
    constexpr int THH = (TH - 1) * STRIDE + KS, TWH = (TW - 1) * STRIDE + KS;
    constexpr int NPIXH = NB * THH * TWH;
    constexpr int MFRAGS = NB * TH * TW / 32, NFRAGS = TN / 32;
    constexpr int WAVES_M = MFRAGS >= 4 ? 4 : MFRAGS, WAVES_N = 4 / WAVES_M;
    constexpr int MI = MFRAGS / WAVES_M, NI = NFRAGS / WAVES_N;
    constexpr int TAPS = KS * KS;
    static_assert(NI >= 1 && MFRAGS % WAVES_M == 0, "bad tile configuration");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // A tile layouts.  PLANAR (stride 1, 16-wide tiles): [k-half][row][32 columns] - 16 consecutive lanes read 256
    // contiguous bytes, the row stride is a multiple of 256 bytes so the four 16-lane groups of a ds_read_b128 each cover
    // all 64 banks exactly once, and a tap is a compile-time offset (ky*32 + kx) on one per-fragment base register.
    // Otherwise: [pixel][2] 16-byte halves, XOR-swizzled by pixel.
    constexpr bool PLANAR = (STRIDE == 1 && TW == 16 && NB == 1 && KS == 5);   // 3x3: the extra registers cost a wave per SIMD
    constexpr int PLSZ = THH * 32, A_ENTRIES = PLANAR ? 2 * PLSZ : NPIXH * CKH;
    uint4* sA = reinterpret_cast<uint4*>(smem_raw);
    uint4* sB = sA + A_ENTRIES;                                           // [TAPS*TN][CKH]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int Cin = p.C1 + p.C2, Cout = p.O1 + p.O2;
    const int cot = (Cout + TN - 1) / TN;
    int bid = xcd_order(blockIdx.x);
    int phase = 0;                                       // Conv2DTranspose: which of the 2x2 output phases
    if (KS == 1 && p.convt) { phase = bid & 3; bid >>= 2; }
    const int co0 = (bid % cot) * TN;
    bid /= cot;
    const int tiles = p.tiles_y * p.tiles_x;
    const int tile = bid % tiles, grp = bid / tiles;
    const int ty0 = (tile / p.tiles_x) * TH, tx0 = (tile % p.tiles_x) * TW;
    const int iy0 = ty0 * STRIDE - p.pad_t, ix0 = tx0 * STRIDE - p.pad_l;
    const int half = lane >> 5;
    const EpiBias epi_pre = epi_bias_preload<NI>(p, lane, wn, co0, Cout);       // in flight under the whole main loop

    int abase[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int P = (wm * MI + mi) * 32 + (lane & 31);
        const int img = P / (TH * TW), rem = P % (TH * TW);
        abase[mi] = PLANAR ? half * PLSZ + (rem / TW) * 32 + (rem % TW)
                           : img * (THH * TWH) + (rem / TW) * STRIDE * TWH + (rem % TW) * STRIDE;
    }
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.0f;

    // Async-stage split (issue early / write late): the global loads of chunk c+1 are issued into registers before the
    // MFMA loop of chunk c and committed to LDS after it, so HBM/L2 latency hides under the matrix work.
    constexpr int AP = (NPIXH * CKH + 255) / 256, BP = (TAPS * TN * CKH + 255) / 256;
    float4 preA[AP][2];
    uint4 preB[BP];
    // The halo-pixel -> image-pixel map (padding mode, tile clipping, image index) does not depend on the channel chunk:
    // resolve it ONCE per workgroup.  Left inside fetch() it was ~700 instructions of branchy address arithmetic per
    // chunk in front of every MFMA loop (in-order issue: as long as the 100 MFMAs themselves).
    int apix[AP];
    unsigned upos[UNP ? AP : 1];                         // UNP: position of the halo pixel inside its 2x2 pooling window
    typedef unsigned int u32x2k __attribute__((ext_vector_type(2)));
    u32x2k preK[UNP ? AP : 1];                           // UNP: the 8 arg-max bytes of the staged 8 channels
#pragma unroll
    for (int q = 0; q < AP; ++q) {
        const int item = tid + q * 256;
        const int pix = item / CKH;
        const int img = pix / (THH * TWH), rem = pix % (THH * TWH);
        int gy = iy0 + rem / TWH, gx = ix0 + rem % TWH;
        const int n = grp * NB + img;
        const bool ok = item < NPIXH * CKH && n < p.N && map_coord(gy, p.H, p.pad_mode) && map_coord(gx, p.W, p.pad_mode);
        if constexpr (UNP) {
            apix[q] = ok ? (n * (p.H >> 1) + (gy >> 1)) * (p.W >> 1) + (gx >> 1) : -1;
            upos[q] = (unsigned)(((gy & 1) << 1) | (gx & 1));
        } else {
            apix[q] = ok ? (n * p.H + gy) * p.W + gx : -1;
        }
    }
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    unsigned aoff[AP];                                   // BUF: byte offset of this thread's 16 B inside the input tensor
    unsigned boff0 = 0, boff_last = 0;
    if constexpr (BUF) {
        static_assert(INB && 128 % TN == 0, "buffer prefetch: bf16-stored input, TN divides 128");
#pragma unroll
        for (int q = 0; q < AP; ++q)
            aoff[q] = apix[q] >= 0 ? (unsigned)((apix[q] * p.C1 + (tid & 1) * 8) * 2) : 0x80000000u;
        const int row = tid >> 1;
        boff0 = (unsigned)((((row / TN) * Cout + co0 + row % TN) * 16 + (tid & 1) * 8) * 2);
        boff_last = (tid + (BP - 1) * 256 < TAPS * TN * 2) ? boff0 : 0x80000000u;
    }
    auto fetch = [&](int c0) {
        if constexpr (BUF) {
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.in1), 0, (int)(((long)p.N * p.H * p.W * p.C1 * 2) >> (UNP ? 2 : 0)), 0x00020000);
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<__bf16*>(p.wb), 0, (int)((long)(p.CinP >> 4) * TAPS * 16 * Cout * 2), 0x00020000);
#pragma unroll
            for (int q = 0; q < AP; ++q) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra, aoff[q], c0 * 2, 0);
                preA[q][0] = *reinterpret_cast<const float4*>(&v);
            }
            if constexpr (UNP) {
                const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<unsigned char*>(p.in_idx), 0, (int)(((long)p.N * p.H * p.W * p.C1) >> 2), 0x00020000);
#pragma unroll
                for (int q = 0; q < AP; ++q)        // byte offsets are half the bf16 offsets; padding stays out of range
                    preK[q] = __builtin_amdgcn_raw_buffer_load_b64(rk, aoff[q] >= 0x80000000u ? 0x80000000u : aoff[q] >> 1, c0, 0);
            }
            const int chunk_base = (c0 >> 4) * (TAPS * 16 * 2) * Cout, qstride = (128 / TN) * Cout * 32;
#pragma unroll
            for (int q = 0; q < BP; ++q) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rb, q == BP - 1 ? boff_last : boff0,
                                                                      chunk_base + q * qstride, 0);
                preB[q] = *reinterpret_cast<const uint4*>(&v);
            }
            return;
        }
        const int c = c0 + (tid & (CKH - 1)) * 8;       // item = tid + 256 q: the 8-channel slot is fixed per thread
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            preA[q][0] = preA[q][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (apix[q] >= 0 && c < Cin) {
                const long pixoff = apix[q];
                if constexpr (INB) {                 // the tensors already hold bf16: 16 bytes = this item's 8 channels
                    const __bf16* src = c < p.C1 ? reinterpret_cast<const __bf16*>(p.in1) + pixoff * p.C1 + c
                                                 : reinterpret_cast<const __bf16*>(p.in2) + pixoff * p.C2 + (c - p.C1);
                    preA[q][0] = *reinterpret_cast<const float4*>(src);
                } else {
                    const float* src = c < p.C1 ? p.in1 + pixoff * p.C1 + c : p.in2 + pixoff * p.C2 + (c - p.C1);
                    preA[q][0] = *reinterpret_cast<const float4*>(src);
                    if (c + 4 < Cin) preA[q][1] = *reinterpret_cast<const float4*>(src + 4);     // Cin % 8 == 4: a 4-channel tail
                }
            }
        }
        // wave-uniform base of this K chunk (the transposed convolution reads tap slot 3 - phase of a 4-tap image)
        const int h8 = tid & (CKH - 1), c16 = (c0 >> 4) + (h8 >> 1);       // this thread's 16-channel image of the chunk
        const __bf16* wchunk = (KS == 1 && p.convt) ? p.wb + ((long)c16 * 4 + (3 - phase)) * 16 * Cout
                                                    : p.wb + (long)c16 * (TAPS * 16) * Cout;
        const bool wok = c16 * 16 < p.CinP;
#pragma unroll
        for (int q = 0; q < BP; ++q) {
            const int item = tid + q * 256;
            const int row = item / CKH, j = row % TN, tap = row / TN;
            preB[q] = make_uint4(0u, 0u, 0u, 0u);
            if (item < TAPS * TN * CKH && co0 + j < Cout && wok)
                preB[q] = *reinterpret_cast<const uint4*>(wchunk + (unsigned)((tap * Cout + co0 + j) * 16 + (h8 & 1) * 8));
        }
    };
    // Diagnostic builds only (-DNIMG_GEN_ABLATE=<bits>, results are wrong): 1 no prefetch after the first chunk, 2 no commit (LDS
    // writes) after the first chunk, 4 no barriers in the loop, 16 no epilogue stores.
#ifndef NIMG_GEN_ABLATE
#define NIMG_GEN_ABLATE 0
#endif
    constexpr int GABL = NIMG_GEN_ABLATE;
    fetch(0);
    for (int ci0 = 0; ci0 < Cin; ci0 += CK) {
        if (!(GABL & 4) || ci0 == 0) __syncthreads();
        if (!(GABL & 2) || ci0 == 0)
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            const int item = tid + q * 256;
            if (item < NPIXH * CKH) {
                const int pix = item / CKH, h8 = item % CKH;
                uint4 packed;
                if constexpr (INB) {
                    packed = *reinterpret_cast<const uint4*>(&preA[q][0]);
                    if constexpr (UNP) packed = unp_route(packed, preK[q][0], preK[q][1], upos[q]);
                } else {
                    const float f[8] = {preA[q][0].x, preA[q][0].y, preA[q][0].z, preA[q][0].w,
                                        preA[q][1].x, preA[q][1].y, preA[q][1].z, preA[q][1].w};
                    const bf16x8 b = pack8(f);
                    packed = *reinterpret_cast<const uint4*>(&b);
                }
                if constexpr (PLANAR) sA[h8 * PLSZ + (pix / TWH) * 32 + pix % TWH] = packed;
                else if constexpr (CKH == 2) sA[pix * 2 + (h8 ^ ((pix >> 3) & 1))] = packed;
                else if constexpr (CKH == 4) sA[pix * 4 + (h8 ^ ((pix >> 2) & 3))] = packed;
                else sA[pix * 8 + (h8 ^ ((pix >> 1) & 7))] = packed;     // 128-byte rows: a b128 lane group = 8 even + 8 odd
                                                                         // pixels, (pix >> 1) & 7 distinct inside each set
            }
        }
        if (!(GABL & 2) || ci0 == 0)
#pragma unroll
        for (int q = 0; q < BP; ++q) {
            const int item = tid + q * 256;
            if (item < TAPS * TN * CKH) {
                const int h8 = item % CKH, row = item / CKH;
                if constexpr (CKH == 2) sB[row * 2 + (h8 ^ ((row >> 3) & 1))] = preB[q];
                else if constexpr (CKH == 4) sB[row * 4 + (h8 ^ ((row >> 2) & 3))] = preB[q];
                else sB[row * 8 + (h8 ^ ((row >> 1) & 7))] = preB[q];
            }
        }
        if (!(GABL & 4) || ci0 == 0) __syncthreads();
        if (ci0 + CK < Cin && !(GABL & 1)) fetch(ci0 + CK);
        // one kernel row unrolled: the next taps' ds_reads overlap the MFMAs.  The 32-channel tiles of the small kernels (the
        // UNet's deep and narrow layers: one or two workgroups per CU, nobody else to cover an LDS round trip) unroll ALL
        // taps - one exposed read latency per chunk instead of one per kernel row: ec42 / ec52 / dc11 -9 ... -13 %; the 64-channel
        // tiles lose 2 - 3 % with it (profiles/r03_m_conv3_unroll_ab.txt)
#pragma unroll(KS <= 3 && TN == 32 ? KS : 1)
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const int tap = ky * KS + kx;
            const int toff = PLANAR ? ky * 32 + kx : ky * TWH + kx;
#pragma unroll
            for (int ks2 = 0; ks2 < CKH / 2; ++ks2) {        // MFMA k-steps of this chunk (16 channels each)
                bf16x8 a[MI], b[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int pix = abase[mi] + toff;
                    const uint4 v = PLANAR ? sA[pix] : (CKH == 2 ? sA[pix * 2 + (half ^ ((pix >> 3) & 1))]
                                             : (CKH == 4 ? sA[pix * 4 + ((2 * ks2 + half) ^ ((pix >> 2) & 3))]
                                                         : sA[pix * 8 + ((2 * ks2 + half) ^ ((pix >> 1) & 7))]));
                    a[mi] = *reinterpret_cast<const bf16x8*>(&v);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int row = tap * TN + (wn * NI + ni) * 32 + (lane & 31);
                    const uint4 v = CKH == 2 ? sB[row * 2 + (half ^ ((row >> 3) & 1))]
                                             : (CKH == 4 ? sB[row * 4 + ((2 * ks2 + half) ^ ((row >> 2) & 3))]
                                                         : sB[row * 8 + ((2 * ks2 + half) ^ ((row >> 1) & 7))]);
                    b[ni] = *reinterpret_cast<const bf16x8*>(&v);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            }
        }
    }
    if constexpr ((GABL & 16) != 0) {          // keep the accumulators alive without storing them
        float sacc = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int j = 0; j < 16; ++j) sacc += acc[mi][ni][j];
        if (sacc == 123.456f) p.out1[0] = sacc;
        return;
    }
    // ---- epilogue, fused activation + 2x2 max-pool (16x16 tiles; the entry point guarantees even Hout/Wout, Cout % 4 == 0)
    if constexpr (TW == 16 && NB == 1 && STRIDE == 1) {
        if (p.pool_out && !(p.flags & NIMG_POOL_ALSO)) {
            float* elds = reinterpret_cast<float*>(smem_raw) + wave * (32 * (NI * 32 + EPI_PAD));
            const int Hp = p.Hout >> 1, Wp = p.Wout >> 1;
            const float al = p.act == 1 ? p.alpha : 1.0f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int py = (ty0 >> 1) + wm * MI + mi;
                pool_via_lds<NI>(acc[mi], elds, lane, al,
                    [&](int c) {
                        const int co = co0 + wn * NI * 32 + c;
                        return (p.bias && co < Cout) ? *reinterpret_cast<const float4*>(p.bias + co)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
                    },
                    [&](int pc, int c, float4 v, uchar4 k) {
                        const int co = co0 + wn * NI * 32 + c, px = (tx0 >> 1) + pc;
                        if (co >= Cout || grp >= p.N || py >= Hp || px >= Wp) return;
                        const long o = (((long)grp * Hp + py) * Wp + px) * Cout + co;
                        if (p.flags & NIMG_BF16_OUT) store4_bf16(p.pool_out, o, v);
                        else *reinterpret_cast<float4*>(p.pool_out + o) = v;
                        if (p.pool_idx) *reinterpret_cast<uchar4*>(p.pool_idx + o) = k;
                    });
            }
            return;
        }
    }
    // ---- epilogue, vector form: accumulators turned around through LDS so each lane stores 16 B along the channels
    if ((p.O1 & 3) == 0 && (p.O2 & 3) == 0) {
        conv_epilogue_vec<KS, TH, TW, NB, MI, NI>(acc, p, smem_raw, wave, lane, wm, wn, co0, Cout, ty0, tx0, grp, phase, epi_pre);
        return;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int co = co0 + (wn * NI + ni) * 32 + (lane & 31);
        if (co >= Cout) continue;
        const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int row = (j & 3) + 8 * (j >> 2) + 4 * half;
                const int P = (wm * MI + mi) * 32 + row;
                const int img = P / (TH * TW), rem = P % (TH * TW);
                const int oy = ty0 + rem / TW, ox = tx0 + rem % TW, n = grp * NB + img;
                if (n >= p.N || oy >= p.Hout || ox >= p.Wout) continue;
                const long pixoff = (KS == 1 && p.convt)
                    ? ((long)n * 2 * p.Hout + 2 * oy + (phase >> 1)) * (2 * p.Wout) + 2 * ox + (phase & 1)
                    : ((long)n * p.Hout + oy) * p.Wout + ox;
                float v = acc[mi][ni][j] + bv;
                if (p.act == 1) v = lrelu(v, p.alpha);
                if (co < p.O1) {
                    if (p.act1) v *= (p.act1[pixoff * p.O1 + co] > 0.f ? 1.0f : p.alpha);
                    p.out1[pixoff * p.O1 + co] = v;
                } else {
                    p.out2[pixoff * p.O2 + (co - p.O1)] = v;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Ring form of the 5x5 stride-1 convolution over bf16-stored activations with Cout % 64 == 0 (FAN conv2 / conv3 / conv4
// forward, conv4 / conv3 input gradient): 4 waves x 8 accumulator fragments per workgroup -
//     TN = 128: 16 x 16 pixels x 128 output channels, 2 x 4 fragments per wave;  TN = 64: 32 x 16 pixels x 64, 4 x 2;
//     TN = 32 (conv2's input gradient, 64 -> 32): 32 x 16 pixels x 32, 4 x 1 fragments - 64 accumulator registers, three
//     workgroups per CU (47 KB of LDS).
// conv_fwd_bf16_kernel stages the whole 25-tap weight tile of a 16-channel chunk through registers (52 VGPRs, 13
// ds_write_b128 per thread and chunk, 51 KB of LDS for 64 output channels) behind two barriers per chunk.  Here the weights
// arrive one KERNEL ROW at a time (5 taps x TN co x 16 ci = 20 / 10 KB) by LDS-DMA (buffer_load_dwordx4 ... lds: no staging
// registers, no write pass) into a two-slot ring - row r + 1 lands while the MFMAs of row r run, one barrier per row - and
// the halo tile of the next chunk (12.5 / 22.5 KB, through registers: padding, un-pool routing) is committed to the second
// of two A buffers (TN = 128) or between two barriers at the chunk boundary (TN = 64: one buffer, LDS budget).  The freed
// registers hold the 8-fragment block: 6 ds_read_b128 per 8 MFMAs instead of 4 per 4, the halo tile is staged once per
// 128 channels (or per 512 pixels) instead of once per 64 x 256, and half as many workgroups pay the prologue / epilogue.
// LDS-DMA writes base + 16 lane: the ring image is lane-linear per 1 KB piece and the XOR swizzle of the 16-byte halves
// (conflict-free ds_read_b128) is applied on the SOURCE address.  LDS: 80 KB (TN = 128) / 56 KB -> two workgroups per CU.
//
// One LDS-DMA piece: 64 lanes x 16 B from buffer `rsrc` (per-lane byte offset voff + scalar soff) to LDS bytes
// [lds_addr, lds_addr + 1024), lane-linear.  Issued as inline asm on purpose: hipcc orders the builtin form
// (__builtin_amdgcn_raw_ptr_buffer_load_lds) against every later ds_read - it emits s_waitcnt vmcnt(0) right behind the
// issue, which serialises the transfer with the MFMA loop it is meant to run under.  The asm form is invisible to the
// compiler's counters (its own waits only become conservative: loads retire in order); the kernel waits for the DMA itself
// (dma_wait) in front of the barrier that publishes the slot.
typedef unsigned int r_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(r_u32x4 rsrc, unsigned lds_addr, unsigned voff, int soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N>
__device__ __forceinline__ void dma_wait_leave() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }   // loads retire in order

// Diagnostic builds only (tools/build_variant.sh + tools/ring_time.py; results are WRONG with any bit set): what the main loop of
// the ring kernels spends where.  1: no weight DMA after the first kernel row; 2: no input fetch / commit after the first chunk;
// 4: no barrier / DMA wait in the loop; 8: the operand fragments are read once, before the loop; 16: the epilogue stores nothing.
#ifndef NIMG_RING_ABLATE
#define NIMG_RING_ABLATE 0
#endif

// NW = waves per workgroup, stacked along the pixels: 4 (a 256- or 512-pixel tile, two or three workgroups per CU) or 8 (twice
// the pixels against the SAME weight ring, one workgroup per CU = still two waves per SIMD).  The weights are 8x the bytes of
// the input tile per K chunk (25 taps x 16 ci x TN co against one halo tile reused by all 25 taps), every workgroup streams
// ALL of them from L2, and the stream is what the main loop loses most to (profiles/r04_b_ring_ablation.txt: without the
// weight DMA the TN = 128 layers run 20 - 26 % faster, without the input fetch 6 %): doubling the pixels per workgroup halves
// the DMA pieces and the L2 bytes per matrix instruction.
template <int TN, int NW = 4, int KS = 5>
struct RingGeom {
    static constexpr int NI = TN / 32, MI = NI == 1 ? 4 : 8 / NI;   // fragment block of a wave (NW waves stacked along the pixels)
    static constexpr int NT = 64 * NW;
    static constexpr int TH = 2 * NW * MI, TW = 16, THH = TH + KS - 1, TWH = TW + KS - 1;
    static constexpr int NPIXH = THH * TWH, AP = (NPIXH * 2 + NT - 1) / NT;
    static constexpr int PLSZ = THH * 32;                    // uint4 entries of one k-half plane of the halo tile
    static constexpr int ABUF = 2 * PLSZ;                    // one A buffer
    static constexpr bool ADBL = TN == 128;                  // two A buffers
    static constexpr int SLOT = KS * TN * 2;                 // one ring slot: [KS taps x TN co][2 halves]
    static constexpr int PIECES = KS * NI, NPW = (PIECES + NW - 1) / NW;   // 1 KB DMA pieces per kernel row, per wave
    // ring depth.  3 (with NW = 8, where the LDS of the one resident workgroup has the room): the row requested in phase r is
    // needed in phase r + 2, so the wait at the end of a phase leaves the youngest row's transfers in flight (counted vmcnt)
    // instead of draining the queue - a weight row gets two phases to arrive from L2 instead of one.
#ifdef NIMG_RING_SLOTS2
    static constexpr int NSLOT = 2;
#else
    static constexpr int NSLOT = (NW == 8 && KS == 5) ? 3 : 2;
#endif
    static constexpr size_t LDS_TILES = (size_t)(NSLOT * SLOT + (ADBL ? 2 : 1) * ABUF) * sizeof(uint4);
    static constexpr size_t LDS_EPI = (size_t)NW * 32 * (TN + EPI_PAD) * sizeof(float);
    static constexpr size_t LDS = LDS_TILES > LDS_EPI ? LDS_TILES : LDS_EPI;
};

template <int TN, bool UNP, int NW = 4, int KS = 5>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : (TN == 32 ? 3 : 2)) void conv5_ring_kernel(const ConvParamsB p) {
    using G = RingGeom<TN, NW, KS>;
    static_assert(KS == 5 || (KS == 3 && !UNP && NW == 4), "kernel size 3: plain input, four waves");
    constexpr int NT = G::NT;
    constexpr int NI = G::NI, MI = G::MI, TH = G::TH, TW = G::TW, TWH = G::TWH, NPIXH = G::NPIXH, AP = G::AP;
    constexpr int PLSZ = G::PLSZ, ABUF = G::ABUF, SLOT = G::SLOT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* sB = reinterpret_cast<uint4*>(smem_raw);          // ring first: the LDS-DMA base (M0) stays below 64 KB
    uint4* sA = sB + G::NSLOT * SLOT;
    const unsigned sB_addr = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cout = p.O1;
    const int cot = Cout / TN;
    int bid = xcd_order(blockIdx.x);
    const int co0 = (bid % cot) * TN;
    bid /= cot;
    const int tiles = p.tiles_y * p.tiles_x;
    const int tile = bid % tiles, grp = bid / tiles;
    const int ty0 = (tile / p.tiles_x) * TH, tx0 = (tile % p.tiles_x) * TW;
    const int iy0 = ty0 - p.pad_t, ix0 = tx0 - p.pad_l;
    const int half = lane >> 5;

    int abase[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int P = (wave * MI + mi) * 32 + (lane & 31);
        abase[mi] = half * PLSZ + (P / TW) * 32 + (P % TW);
    }
    const int bbase = (lane & 31) * 2 + (half ^ (((lane & 31) >> 3) & 1));
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.0f;

    // halo tile: NPIXH pixels x 2 eight-channel slots, items of 16 B, item = tid + 256 q (the slot is fixed per thread)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2k __attribute__((ext_vector_type(2)));
    unsigned aoff[AP], upos[UNP ? AP : 1];
    int adst[AP];
#pragma unroll
    for (int q = 0; q < AP; ++q) {
        const int item = tid + q * NT, pix = item >> 1;
        int gy = iy0 + pix / TWH, gx = ix0 + pix % TWH;
        const bool ok = (item < NPIXH * 2) & (grp < p.N) & map_coord(gy, p.H, p.pad_mode) & map_coord(gx, p.W, p.pad_mode);
        int apix;
        if constexpr (UNP) {
            apix = (grp * (p.H >> 1) + (gy >> 1)) * (p.W >> 1) + (gx >> 1);
            upos[q] = (unsigned)(((gy & 1) << 1) | (gx & 1));
        } else {
            apix = (grp * p.H + gy) * p.W + gx;
        }
        aoff[q] = ok ? (unsigned)((apix * p.C1 + (tid & 1) * 8) * 2) : 0x80000000u;
        adst[q] = (tid & 1) * PLSZ + (pix / TWH) * 32 + pix % TWH;
    }
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in1), 0, (int)(((long)p.N * p.H * p.W * p.C1 * 2) >> (UNP ? 2 : 0)), 0x00020000);
    const unsigned long wb_addr = (unsigned long)p.wb;
    const r_u32x4 rb = {(unsigned)wb_addr, (unsigned)(wb_addr >> 32) & 0xffffu,
                        (unsigned)((long)(p.CinP >> 4) * KS * KS * 16 * Cout * 2), 0x00020000u};
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(UNP ? p.in_idx : reinterpret_cast<const unsigned char*>(p.in1)), 0,
        (int)(((long)p.N * p.H * p.W * p.C1) >> 2), 0x00020000);
    uint4 preA[AP];
    u32x2k preK[UNP ? AP : 1];
    auto fetchA = [&](int c0) {
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra, aoff[q], c0 * 2, 0);
            preA[q] = *reinterpret_cast<const uint4*>(&v);
            if constexpr (UNP)
                preK[q] = __builtin_amdgcn_raw_buffer_load_b64(rk, aoff[q] >= 0x80000000u ? 0x80000000u : aoff[q] >> 1, c0, 0);
        }
    };
    auto commitA = [&](int buf) {                       // buf: entry offset of the A buffer
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            if (tid + q * NT < NPIXH * 2) {
                uint4 v = preA[q];
                if constexpr (UNP) v = unp_route(v, preK[q][0], preK[q][1], upos[q]);
                sA[buf + adst[q]] = v;
            }
        }
    };
    // weights wb[chunk][tap][co][16]: one kernel row of a chunk = 5 taps x TN co x 32 B = 5 NI pieces of 1 KB (piece k =
    // tap k / NI, 32-channel block k % NI, at ring byte 1024 k); wave w moves the pieces w, w + 4, ...  Lane l of a piece
    // writes 16-byte position l: row l >> 1, and position parity (l & 1) must hold half h = (l & 1) ^ (row >> 3 & 1) - the
    // swizzle the fragment reads undo.
    const unsigned bvoff = (unsigned)(((co0 + (lane >> 1)) * 16 + (((lane & 1) ^ ((lane >> 4) & 1)) * 8)) * 2);
    auto gldsB = [&](int chunk, int ky, int slot) {
#pragma unroll
        for (int j = 0; j < G::NPW; ++j) {
            const int k = wave + NW * j;
            if (G::PIECES % NW == 0 || k < G::PIECES) {
                const int soff = ((chunk * KS * KS + ky * KS + k / NI) * Cout + (k % NI) * 32) * 32;
                glds16(rb, sB_addr + (unsigned)((slot * SLOT + k * 64) * 16), bvoff, soff);
            }
        }
    };
    const int chunks = p.C1 >> 4;
    gldsB(0, 0, 0);
    if constexpr (G::NSLOT == 3) gldsB(0, 1, 1);
    fetchA(0);
    commitA(0);
    dma_wait();
    __syncthreads();
    constexpr int ABL = NIMG_RING_ABLATE;
    bf16x8 a0[MI], b0[NI];
    if constexpr (ABL & 8) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) { const uint4 v = sA[abase[mi]]; a0[mi] = *reinterpret_cast<const bf16x8*>(&v); }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) { const uint4 v = sB[bbase + ni * 64]; b0[ni] = *reinterpret_cast<const bf16x8*>(&v); }
    }
    if constexpr (G::NSLOT == 3) {
        static_assert(G::ADBL, "the three-slot ring is written for the double-buffered input tile");
        // transfers of ONE wave per kernel row: waves below PIECES % NW move one piece more
        constexpr int NLO = G::PIECES / NW, NREM = G::PIECES % NW;
        constexpr int NA = AP * (UNP ? 2 : 1);             // the input prefetch of phase 0: loads queued BEHIND that phase's row
        int s0 = 0;                                        // slot of kernel row 0 of this chunk = (5 c) % 3
        for (int c = 0; c < chunks; ++c) {
            const int ab = (c & 1) * ABUF;
            const bool more = c + 1 < chunks;
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                const int slot = (s0 + ky) % 3, slot2 = (s0 + ky + 2) % 3;
                if (ky == 2 && more) commitA(ab ^ ABUF);   // (the compiler drains the queue for the prefetched registers here)
                const bool issue = ky < 3 || more;         // row r + 2 exists
                if (ky < 3) gldsB(c, ky + 2, slot2);
                else if (more) gldsB(c + 1, ky - 3, slot2);
                if (ky == 0 && more) fetchA((c + 1) * 16);
                const uint4* sBs = sB + slot * SLOT + bbase;
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
                    bf16x8 a[MI], b[NI];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const uint4 v = sA[ab + abase[mi] + ky * 32 + kx];
                        a[mi] = *reinterpret_cast<const bf16x8*>(&v);
                    }
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const uint4 v = sBs[(kx * TN + ni * 32) * 2];
                        b[ni] = *reinterpret_cast<const bf16x8*>(&v);
                    }
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
                }
                // row r + 1 must have landed; what may stay in flight is younger: this phase's row and, in phases 0 / 1 of a
                // chunk, the input prefetch queued behind phase 0's row
                if (!issue) dma_wait();
                else if (ky < 2 && more) {
                    if (NREM && wave < NREM) dma_wait_leave<NLO + 1 + NA>();
                    else dma_wait_leave<NLO + NA>();
                } else {
                    if (NREM && wave < NREM) dma_wait_leave<NLO + 1>();
                    else dma_wait_leave<NLO>();
                }
                // raw barrier: __syncthreads() may drain the memory queue for its fence - the youngest row has to stay in flight.
                // What has to be ordered here is LDS only: this wave's tile writes (lgkmcnt) and its landed transfers (above).
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            s0 = (s0 + 2) % 3;
        }
    } else
    for (int c = 0; c < chunks; ++c) {
        const int ab = G::ADBL ? (c & 1) * ABUF : 0;
        const bool more = c + 1 < chunks;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int slot = (c + ky) & 1;                 // (KS c + ky) & 1, KS odd
            if constexpr (!(ABL & 1)) {
                if (ky < KS - 1) gldsB(c, ky + 1, slot ^ 1);
                else if (more) gldsB(c + 1, 0, slot ^ 1);
            }
            if constexpr (!(ABL & 2)) if (ky == 0 && more) fetchA((c + 1) * 16);
            const uint4* sBs = sB + slot * SLOT + bbase;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                bf16x8 a[MI], b[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    if constexpr (ABL & 8) { a[mi] = a0[mi]; continue; }
                    const uint4 v = sA[ab + abase[mi] + ky * 32 + kx];
                    a[mi] = *reinterpret_cast<const bf16x8*>(&v);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if constexpr (ABL & 8) { b[ni] = b0[ni]; continue; }
                    const uint4 v = sBs[(kx * TN + ni * 32) * 2];
                    b[ni] = *reinterpret_cast<const bf16x8*>(&v);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            }
            if constexpr (!(ABL & 2)) if (G::ADBL && ky == KS - 1 && more) commitA(ab ^ ABUF);
            if constexpr (!(ABL & 4)) {
                dma_wait();                                // the next kernel row has landed ...
                __syncthreads();                           // ... and everyone is done with this one (slot and A buffer free)
            }
            if constexpr (!(ABL & 2)) if (!G::ADBL && ky == KS - 1 && more) {
                commitA(0);
                if constexpr (!(ABL & 4)) __syncthreads();
            }
        }
    }
    if constexpr (ABL & 4) { dma_wait(); __syncthreads(); }
    if constexpr (ABL & 16) {              // keep the accumulators alive without storing them
        float sacc = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int j = 0; j < 16; ++j) sacc += acc[mi][ni][j];
        if (sacc == 123.456f) p.out1[0] = sacc;
        return;
    }
    if constexpr (KS == 3) {        // the 3x3 layers (codec, UNet): every epilogue option of conv_fwd_bf16_kernel, same code
        conv_epilogue_vec<3, TH, TW, 1, MI, NI>(acc, p, smem_raw, wave, lane, wave, 0, co0, Cout, ty0, tx0, grp, 0,
                                                epi_bias_preload<NI>(p, lane, 0, co0, Cout));
        return;
    }
    // epilogue: per-wave private LDS scratch (the loop's last barrier released the tiles) -> wave-level ordering only
    float* elds = reinterpret_cast<float*>(smem_raw) + wave * (32 * (NI * 32 + EPI_PAD));
    if (p.pool_out) {                                      // fused activation + 2x2 max-pool (even Hout / Wout)
        const int Hp = p.Hout >> 1, Wp = p.Wout >> 1;
        const float al = p.act == 1 ? p.alpha : 1.0f;
        if (p.flags & NIMG_BF16_OUT) {                     // bf16-stored: 16-byte stores of 8 channels (+ 8 arg-max bytes)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int py = (ty0 >> 1) + wave * MI + mi;
                pool_in_regs8<NI>(acc[mi], elds, lane, al,
                    [&](int c) { return p.bias ? p.bias[co0 + c] : 0.f; },
                    [&](int pc, int c, float4 lo, float4 hi, uint2 k) {
                        const int px = (tx0 >> 1) + pc;
                        if (grp >= p.N || py >= Hp || px >= Wp) return;
                        const long o = (((long)grp * Hp + py) * Wp + px) * Cout + co0 + c;
                        const float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                        *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.pool_out) + o) = pack8(f);
                        if (p.pool_idx) *reinterpret_cast<uint2*>(p.pool_idx + o) = k;
                    });
            }
            return;
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int py = (ty0 >> 1) + wave * MI + mi;
            pool_in_regs<NI>(acc[mi], elds, lane, al,
                [&](int c) { return p.bias ? p.bias[co0 + c] : 0.f; },
                [&](int pc, int c, float4 v, uchar4 k) {
                    const int co = co0 + c, px = (tx0 >> 1) + pc;
                    if (grp >= p.N || py >= Hp || px >= Wp) return;
                    const long o = (((long)grp * Hp + py) * Wp + px) * Cout + co;
                    if (p.flags & NIMG_BF16_OUT) store4_bf16(p.pool_out, o, v);
                    else *reinterpret_cast<float4*>(p.pool_out + o) = v;
                    if (p.pool_idx) *reinterpret_cast<uchar4*>(p.pool_idx + o) = k;
                });
        }
        return;
    }
    if ((p.flags & NIMG_BF16_OUT) && (!p.act1 || (p.flags & NIMG_BF16_MASK))) {     // bf16-stored output (and mask): 16-byte rows
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            epilogue_via_lds8<NI>(acc[mi], elds, lane, [&](int row, int c, float4 lo, float4 hi) {
                const int P = (wave * MI + mi) * 32 + row;
                const int oy = ty0 + P / TW, ox = tx0 + P % TW;
                if (grp >= p.N || oy >= p.Hout || ox >= p.Wout) return;
                const long o = (((long)grp * p.Hout + oy) * p.Wout + ox) * Cout + co0 + c;
                float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if (p.bias) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += p.bias[co0 + c + e];
                }
                if (p.act == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = lrelu(f[e], p.alpha);
                }
                if (p.act1) {
                    const bf16x8 m = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(p.act1) + o);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] *= (float)m[e] > 0.f ? 1.0f : p.alpha;
                }
                *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.out1) + o) = pack8(f);
            });
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        epilogue_via_lds<NI, false>(acc[mi], elds, lane, [&](int row, int c, float4 v) {
            const int co = co0 + c;
            const int P = (wave * MI + mi) * 32 + row;
            const int oy = ty0 + P / TW, ox = tx0 + P % TW;
            if (grp >= p.N || oy >= p.Hout || ox >= p.Wout) return;
            const long o = (((long)grp * p.Hout + oy) * p.Wout + ox) * Cout + co;
            if (p.bias) {
                const float4 b = *reinterpret_cast<const float4*>(p.bias + co);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            if (p.act == 1) {
                v.x = lrelu(v.x, p.alpha); v.y = lrelu(v.y, p.alpha); v.z = lrelu(v.z, p.alpha); v.w = lrelu(v.w, p.alpha);
            }
            if (p.act1) {
                const float4 m = (p.flags & NIMG_BF16_MASK) ? load4_bf16(p.act1, o) : *reinterpret_cast<const float4*>(p.act1 + o);
                v.x *= m.x > 0.f ? 1.0f : p.alpha; v.y *= m.y > 0.f ? 1.0f : p.alpha;
                v.z *= m.z > 0.f ? 1.0f : p.alpha; v.w *= m.w > 0.f ? 1.0f : p.alpha;
            }
            if (p.flags & NIMG_BF16_OUT) store4_bf16(p.out1, o, v);
            else *reinterpret_cast<float4*>(p.out1 + o) = v;
        });
    }
}

// ------------------------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution over bf16-STORED activations (the UNet's and the codec's layers in throughput mode, forward and input
// gradient) with BOTH operand tiles staged by LDS-DMA.  conv_fwd_bf16_kernel prefetches the next 16-channel chunk into registers
// (3 + 5 ... 9 x 16 B per thread), writes it to LDS between two barriers per chunk and keeps ~40 VGPRs for it; on these layers
// that staging is a quarter of the kernel's time (profiles/r04_j_conv3_ablation.txt: -24 % without prefetch / commit / barriers,
// up to -42 % on the deep layers whose K loop is long and whose tiles are few).  Here the halo tile (16-byte items = one pixel's
// 8 channels; padding = out-of-range offsets the hardware answers with zeros; the two tensors of a concatenated input are two
// buffer descriptors chosen per chunk) and the weight tile ([tap][co] rows of 32 B, wb[chunk][tap][co][16]) of chunk c + 1 are
// requested with buffer_load ... lds into the SECOND of two tile buffers while the matrix instructions of chunk c run: no staging
// registers, no write pass, one barrier per chunk.  LDS images are lane-linear per 1 KB piece; the XOR swizzle of the 16-byte
// halves (conflict-free ds_read_b128, as in the generic kernel) is applied on the source address.  Same tiles, fragment reads
// and epilogue as conv_fwd_bf16_kernel<3, 1, TH, TW, NB, TN, true, ...>: the results are bit-identical.
// LAY 1: the halo tile as two PLANES (k-half 0 / 1) of 16 B per pixel - row pitch 24 entries for the 16 x 16 tile (18 used),
// 10 for the 8 x 8 x 4 tile with an image stride of 104 - and the fragment -> pixel maps of tile_pixel<..., 1>: every 16-lane
// group of a ds_read_b128 meets 16 different 16-byte bank quads at every tap shift (the pixel-major tile with its one XOR bit
// cannot: profiles/r06_conv3_stages_pipe.txt, SQ_LDS_BANK_CONFLICT = 0.57 of the LDS cycles on the 8 x 8 x 4 tile).
template <int TH, int TW, int NB, int TN, int NS = 2, int LAY = 0>
struct Dma3Geom {
    static constexpr int THH = TH + 2, TWH = TW + 2, NPIXH = NB * THH * TWH;
    static constexpr int PITCH = NB == 1 ? 24 : TWH, IMGS = NB == 1 ? THH * PITCH : 104;        // LAY 1: entries
    static constexpr int PLANE = ((NB * IMGS + 63) / 64) * 64;                                  // LAY 1: whole 1 KB pieces
    static constexpr int A_PIECES = LAY ? 2 * PLANE / 64 : (NPIXH * 2 + 63) / 64, B_PIECES = 9 * TN * 2 / 64;
    static constexpr int A_ENT = A_PIECES * 64, B_ENT = 9 * TN * 2;                     // uint4 entries of one buffer
    static constexpr int APW = (A_PIECES + 3) / 4, BPW = (B_PIECES + 3) / 4;            // pieces per wave
    static constexpr size_t LDS_TILES = (size_t)NS * (A_ENT + B_ENT) * sizeof(uint4);
    static constexpr size_t LDS_EPI = (size_t)4 * 32 * (TN + EPI_PAD) * sizeof(float);
    static constexpr size_t LDS = LDS_TILES > LDS_EPI ? LDS_TILES : LDS_EPI;
    static_assert(!LAY || (NB == 1 && TH == 16 && TW == 16) || (NB == 4 && TH == 8 && TW == 8), "plane layout: two tile shapes");
    static_assert(!LAY || THH * TWH <= IMGS, "image stride");
};

#ifdef NIMG_CONV3_TIMING
// Diagnostic build (tools/build_variant.sh timing "-DNIMG_CONV3_TIMING" conv_bf16; tools/conv3_timing.py): per-wave s_memtime sums
// of the K loop's segments - [0] transfer issue, [1] operand reads + matrix instructions, [2] wait for the next chunk's
// transfers, [3] barrier, [4] prologue (kernel start -> first chunk ready), [5] epilogue, [6] chunks - of workgroups 0 .. 63.
__device__ unsigned long long g_conv3_timing[64 * 4 * 8];
#define T3_NOW() ({ unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t_; })
#endif
// LDR: a FIFTH wave does nothing but request the tiles (all pieces of chunk c + 1 while the four others multiply chunk c) and waits
// for them in front of the chunk's barrier.  tools/conv3_timing.py: a wave that requests its share of a chunk in front of the
// taps spends 440 - 490 cycles per chunk there (the 22 - 32 pieces of the workgroup queue up in the CU's one address unit at
// 64 B / clock), with its SIMD's matrix pipe idle when it is the only wave on it; behind the taps' matrix instructions the
// pieces cost the same (ILV).  The loader takes that time off the multiplying waves' loop.
template <int TH, int TW, int NB, int TN, int NS = 2, bool PIPE = false, int LAY = 0, bool ILV = false, bool LDR = false>
__global__ __launch_bounds__(LDR ? 320 : 256) void conv3_dma_kernel(const ConvParamsB p) {
    using G = Dma3Geom<TH, TW, NB, TN, NS, LAY>;
    constexpr int THH = G::THH, TWH = G::TWH, NPIXH = G::NPIXH;
    constexpr int MFRAGS = NB * TH * TW / 32, NFRAGS = TN / 32;
    constexpr int WAVES_M = MFRAGS >= 4 ? 4 : MFRAGS, WAVES_N = 4 / WAVES_M;
    constexpr int MI = MFRAGS / WAVES_M, NI = NFRAGS / WAVES_N;
    static_assert(NI >= 1 && MFRAGS % WAVES_M == 0, "bad tile configuration");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
#ifdef NIMG_CONV3_TIMING
    const unsigned long long tk0 = T3_NOW();
    unsigned long long tsum[4] = {0, 0, 0, 0};
#endif
    uint4* sA = reinterpret_cast<uint4*>(smem_raw);                  // [NS][A_ENT] then [NS][B_ENT]
    uint4* sB = sA + NS * G::A_ENT;
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M, wn = (wave / WAVES_M) % WAVES_N;
    static_assert(!LDR || (NS == 2 && !PIPE && !ILV), "loader wave: two tile buffers, plain tap loop");
    constexpr int NLW = LDR ? 1 : 4;                                   // waves that request tiles, lw = this wave's index among them
    const bool is_loader = LDR ? wave == 4 : true;
    const int lw = LDR ? 0 : wave;
    constexpr int APW_L = (G::A_PIECES + NLW - 1) / NLW, BPW_L = (G::B_PIECES + NLW - 1) / NLW;
    const int Cin = p.C1 + p.C2, Cout = p.O1 + p.O2;
    const int cot = (Cout + TN - 1) / TN;
    int bid = xcd_order(blockIdx.x);
    const int co0 = (bid % cot) * TN;
    bid /= cot;
    const int tiles = p.tiles_y * p.tiles_x;
    const int tile = bid % tiles, grp = bid / tiles;
    const int ty0 = (tile / p.tiles_x) * TH, tx0 = (tile % p.tiles_x) * TW;
    const int iy0 = ty0 - p.pad_t, ix0 = tx0 - p.pad_l;
    const int half = lane >> 5;
    const EpiBias epi_pre = epi_bias_preload<NI>(p, lane, wn, co0, Cout);       // older than every DMA request: retires first
    // DMA pieces THIS wave requests per chunk (the wave's vmcnt sees only its own): the partial waits of the deeper rings count them
    constexpr int AREM = G::A_PIECES % 4, BREM = G::B_PIECES % 4;
    const int short_by = ((AREM && wave >= AREM) ? 1 : 0) + ((BREM && wave >= BREM) ? 1 : 0);

    int abase[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int P = (wm * MI + mi) * 32 + (lane & 31);
        if constexpr (LAY) {
            int img, dy, dx;
            tile_pixel<TH, TW, NB, 1>(P, img, dy, dx);
            abase[mi] = (lane >> 5) * G::PLANE + img * G::IMGS + dy * G::PITCH + dx;      // the lane's k-half plane included
        } else {
            const int img = P / (TH * TW), rem = P % (TH * TW);
            abase[mi] = img * (THH * TWH) + (rem / TW) * TWH + (rem % TW);
        }
    }
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.0f;

    // halo tile: piece k = wave + 4 j, lane l -> item 64 k + l = (pixel, 16-byte slot); the slot holds the channel half
    // slot ^ (pixel >> 3 & 1).  The pixel's element index in the tensors is resolved once; per chunk only the scalar offset moves.
    unsigned aoff1[APW_L], aoff2[APW_L];
    if (is_loader)
#pragma unroll
    for (int j = 0; j < APW_L; ++j) {
        const int item = (lw + NLW * j) * 64 + lane;
        int h, img, ry, rx;
        bool in_tile;
        if constexpr (LAY) {               // entry = (plane, image, row, column); pad entries read nothing (zeros)
            h = item / G::PLANE;
            const int r = item % G::PLANE;
            img = r / G::IMGS;
            const int q = r % G::IMGS;
            ry = q / G::PITCH;
            rx = q % G::PITCH;
            in_tile = (item < 2 * G::PLANE) & (img < NB) & (ry < THH) & (rx < TWH);
        } else {
            const int pix = item >> 1;
            h = (item & 1) ^ ((pix >> 3) & 1);
            img = pix / (THH * TWH);
            const int rem = pix % (THH * TWH);
            ry = rem / TWH;
            rx = rem % TWH;
            in_tile = item < NPIXH * 2;
        }
        int gy = iy0 + ry, gx = ix0 + rx;
        const int n = grp * NB + img;
        const bool ok = in_tile & (n < p.N) & map_coord(gy, p.H, p.pad_mode) & map_coord(gx, p.W, p.pad_mode);
        const unsigned apix = (unsigned)((n * p.H + gy) * p.W + gx);
        aoff1[j] = ok ? (apix * (unsigned)p.C1 + 8u * h) * 2u : 0x80000000u;
        aoff2[j] = ok ? (apix * (unsigned)p.C2 + 8u * h) * 2u : 0x80000000u;
    }
    const unsigned long a1 = (unsigned long)p.in1, a2 = (unsigned long)p.in2, wb_addr = (unsigned long)p.wb;
    const long npx = (long)p.N * p.H * p.W;
    const r_u32x4 ra1 = {(unsigned)a1, (unsigned)(a1 >> 32) & 0xffffu, (unsigned)(npx * p.C1 * 2), 0x00020000u};
    const r_u32x4 ra2 = {(unsigned)a2, (unsigned)(a2 >> 32) & 0xffffu, (unsigned)(npx * p.C2 * 2), 0x00020000u};
    const r_u32x4 rb = {(unsigned)wb_addr, (unsigned)(wb_addr >> 32) & 0xffffu,
                        (unsigned)((long)(p.CinP >> 4) * 9 * 16 * Cout * 2), 0x00020000u};
    // weights: piece k = (tap, 32-channel block) in tap-major order = rows 32 k .. 32 k + 31 of the [9 TN] x 32 B tile; lane l
    // writes 16-byte position l: row l >> 1, which must hold half (l & 1) ^ (row >> 3 & 1)
    const int bj = lane >> 1;
    const unsigned bvoff = (unsigned)(((co0 + bj) * 16 + (((lane & 1) ^ ((lane >> 4) & 1)) * 8)) * 2);
    constexpr int NB32 = TN / 32;
    auto issue = [&](int c0, int buf) {
        const bool first = c0 < p.C1;
        const r_u32x4 ra = first ? ra1 : ra2;
        const int soff = (first ? c0 : c0 - p.C1) * 2;
#pragma unroll
        for (int j = 0; j < APW_L; ++j) {
            const int k = lw + NLW * j;
            if (G::A_PIECES % NLW == 0 || k < G::A_PIECES)
                glds16(ra, lds0 + (unsigned)((buf * G::A_ENT + k * 64) * 16), first ? aoff1[j] : aoff2[j], soff);
        }
        const int chunk = c0 >> 4;
#pragma unroll
        for (int j = 0; j < BPW_L; ++j) {
            const int k = lw + NLW * j;
            if (G::B_PIECES % NLW == 0 || k < G::B_PIECES) {
                const int tap = k / NB32, nb = k % NB32;
                // rows beyond Cout (a partial last channel tile) read the next tap's rows or run out of range (zeros): their
                // products land in accumulator columns the epilogue never stores
                glds16(rb, lds0 + (unsigned)((NS * G::A_ENT + buf * G::B_ENT + k * 64) * 16), bvoff,
                       ((chunk * 9 + tap) * Cout + nb * 32) * 32);
            }
        }
    };
    // ILV: the same pieces one at a time, each behind the matrix instructions of one tap (conv3_timing: a wave spends ~100 cycles
    // per piece in the issue; in front of the taps that is 490 cycles per chunk during which its SIMD's matrix pipe idles)
    auto issue_one = [&](int c0, int buf, int idx) {
        const bool first = c0 < p.C1;
        if (idx < G::APW) {
            const int k = wave + 4 * idx;
            if (G::A_PIECES % 4 == 0 || k < G::A_PIECES)
                glds16(first ? ra1 : ra2, lds0 + (unsigned)((buf * G::A_ENT + k * 64) * 16), first ? aoff1[idx] : aoff2[idx],
                       (first ? c0 : c0 - p.C1) * 2);
        } else if (idx < G::APW + G::BPW) {
            const int k = wave + 4 * (idx - G::APW);
            if (G::B_PIECES % 4 == 0 || k < G::B_PIECES)
                glds16(rb, lds0 + (unsigned)((NS * G::A_ENT + buf * G::B_ENT + k * 64) * 16), bvoff,
                       (((c0 >> 4) * 9 + k / NB32) * Cout + (k % NB32) * 32) * 32);
        }
    };
    static_assert(!ILV || (G::APW + G::BPW <= 9 && NS == 2 && !PIPE), "one piece per tap");
    auto ldA = [&](const uint4* tA, int mi, int ky, int kx) -> uint4 {
        if constexpr (LAY) return tA[abase[mi] + ky * G::PITCH + kx];          // tap shift = an immediate offset
        else {
            const int pix = abase[mi] + ky * TWH + kx;
            return tA[pix * 2 + (half ^ ((pix >> 3) & 1))];
        }
    };
    // NS-slot ring, chunk c + NS - 1 requested while chunk c multiplies: a request has NS - 1 chunks' worth of matrix
    // instructions (18 ... 36 per wave and chunk) to come back from L2 / HBM instead of one.  NS = 2 is the original double buffer.
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s * 16 < Cin && is_loader) issue(s * 16, s);
    dma_wait();
    __syncthreads();
#ifdef NIMG_CONV3_TIMING
    const unsigned long long tk1 = T3_NOW();
#endif
    for (int c0 = 0, buf = 0, nbuf = NS - 1; c0 < Cin; c0 += 16) {
        const bool more = c0 + (NS - 1) * 16 < Cin;
#ifdef NIMG_CONV3_TIMING
        const unsigned long long t0 = T3_NOW();
#endif
        if constexpr (!ILV) { if (more && is_loader) issue(c0 + (NS - 1) * 16, nbuf); }
#ifdef NIMG_CONV3_TIMING
        const unsigned long long t1 = T3_NOW();
#endif
        const uint4* tA = sA + buf * G::A_ENT;
        const uint4* tB = sB + buf * G::B_ENT;
        if (LDR && wave == 4) {
            // the loader multiplies nothing
        } else
        if constexpr (PIPE) {
            // every operand fragment of the chunk is requested from LDS BEFORE the first matrix instruction (9 (MI + NI) x 4
            // registers); the waits in front of the matrix instructions then count down one queue instead of each tap paying
            // an LDS round trip: with one or two waves per SIMD nothing else covers that latency (profiles/r06_conv3_pipe.txt)
            bf16x8 a[9][MI], b[9][NI];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int row = tap * TN + (wn * NI + ni) * 32 + (lane & 31);
                    const uint4 v = tB[row * 2 + (half ^ ((row >> 3) & 1))];
                    b[tap][ni] = *reinterpret_cast<const bf16x8*>(&v);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const uint4 v = ldA(tA, mi, tap / 3, tap % 3);
                    a[tap][mi] = *reinterpret_cast<const bf16x8*>(&v);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tap][mi], b[tap][ni], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else
#pragma unroll((TN == 32 || ILV) ? 3 : 1)
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tap = ky * 3 + kx;
                bf16x8 a[MI], b[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const uint4 v = ldA(tA, mi, ky, kx);
                    a[mi] = *reinterpret_cast<const bf16x8*>(&v);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int row = tap * TN + (wn * NI + ni) * 32 + (lane & 31);
                    const uint4 v = tB[row * 2 + (half ^ ((row >> 3) & 1))];
                    b[ni] = *reinterpret_cast<const bf16x8*>(&v);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
                if constexpr (ILV) { if (more) issue_one(c0 + 16, nbuf, tap); }
            }
#ifdef NIMG_CONV3_TIMING
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t2 = T3_NOW();
#endif
        // chunk c + 1 has landed (the NS - 2 younger ones may still be in flight: loads retire in order) ...
        if constexpr (NS == 2) dma_wait();
        else {
            constexpr int FULL = (NS - 2) * (G::APW + G::BPW);
            if (!more) dma_wait();
            else if (short_by == 0) dma_wait_leave<FULL>();
            else if (short_by == 1) dma_wait_leave<FULL - (NS - 2)>();
            else dma_wait_leave<FULL - 2 * (NS - 2)>();
        }
#ifdef NIMG_CONV3_TIMING
        const unsigned long long t3 = T3_NOW();
#endif
        __syncthreads();                               // ... and everyone is done with this one
#ifdef NIMG_CONV3_TIMING
        const unsigned long long t4 = T3_NOW();
        tsum[0] += t1 - t0; tsum[1] += t2 - t1; tsum[2] += t3 - t2; tsum[3] += t4 - t3;
#endif
        buf = buf + 1 == NS ? 0 : buf + 1;
        nbuf = nbuf + 1 == NS ? 0 : nbuf + 1;
    }
#ifdef NIMG_CONV3_TIMING
    const unsigned long long tk2 = T3_NOW();
#endif
    if (LDR && wave == 4) {
        __syncthreads();                   // the epilogue's one barrier (the scratch aliases the tiles)
        return;
    }
    conv_epilogue_vec<3, TH, TW, NB, MI, NI, LAY>(acc, p, smem_raw, wave, lane, wm, wn, co0, Cout, ty0, tx0, grp, 0, epi_pre);
#ifdef NIMG_CONV3_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tk3 = T3_NOW();
    if (blockIdx.x < 64 && lane == 0 && wave < 4) {
        unsigned long long* d = g_conv3_timing + (blockIdx.x * 4 + wave) * 8;
        d[0] = tsum[0]; d[1] = tsum[1]; d[2] = tsum[2]; d[3] = tsum[3]; d[4] = tk1 - tk0; d[5] = tk3 - tk2; d[6] = Cin / 16; d[7] = tk3 - tk0;
    }
#endif
}

template <int TH, int TW, int NB, int TN, int NS = 2, bool PIPE = false, int LAY = 0, bool ILV = false, bool LDR = false>
int launch_conv3_dma(const ConvParamsB& p, hipStream_t stream) {
    using G = Dma3Geom<TH, TW, NB, TN, NS, LAY>;
    if constexpr (NS == 2 && !PIPE && LAY == 0 && !ILV && !LDR) {
        // The conflict-free plane layout of the halo tile (Dma3Geom LAY 1) on the 8 x 8 x 4 tile - the UNet's 8 x 8 level, -6 %
        // (profiles/r06_conv3_planes.txt); NIMG_CONV3_PLANES=0 switches it off.  Its fragment -> pixel map has no fused-pooling
        // epilogue: those layers keep the pixel-major tile.
        static const int planes = getenv("NIMG_CONV3_PLANES") ? atoi(getenv("NIMG_CONV3_PLANES")) : 1;
#ifdef NIMG_CONV3_VARIANTS
        // Round-6 experiments on what the K loop waits for, all measured NEGATIVE (profiles/r06_conv3_stages_pipe.txt,
        // r06_conv3_planes.txt, r06_conv3_loop_anatomy.txt); instantiated in A/B builds only
        // (tools/build_variant.sh v "-DNIMG_CONV3_VARIANTS" conv_bf16):
        //   NIMG_CONV3_STAGES=3   three-slot ring (a chunk's transfers get two chunks of matrix work to land)
        //   NIMG_CONV3_PIPE=1     all operand fragments of a chunk requested from LDS before its first matrix instruction
        //   NIMG_CONV3_PLANES=2   the plane layout on the 16 x 16 tile too
        //   NIMG_CONV3_ILV=1      the next chunk's transfers issued one piece per tap behind that tap's matrix instructions
        //   NIMG_CONV3_LOADER=1   a fifth wave requests the tiles
        static const int stages = getenv("NIMG_CONV3_STAGES") ? atoi(getenv("NIMG_CONV3_STAGES")) : 2;
        static const int pipe = getenv("NIMG_CONV3_PIPE") ? atoi(getenv("NIMG_CONV3_PIPE")) : 0;
        static const int ilv = getenv("NIMG_CONV3_ILV") ? atoi(getenv("NIMG_CONV3_ILV")) : 0;
        static const int loader = getenv("NIMG_CONV3_LOADER") ? atoi(getenv("NIMG_CONV3_LOADER")) : 0;
        const bool plain = stages != 3 && !pipe && !ilv;
        if ((planes == 2 || (planes == 1 && NB == 4)) && !p.pool_out && stages != 3) {
            if (pipe) return launch_conv3_dma<TH, TW, NB, TN, 2, true, 1>(p, stream);
            if (ilv) return launch_conv3_dma<TH, TW, NB, TN, 2, false, 1, true>(p, stream);
            if (loader) return launch_conv3_dma<TH, TW, NB, TN, 2, false, 1, false, true>(p, stream);
            return launch_conv3_dma<TH, TW, NB, TN, 2, false, 1>(p, stream);
        }
        if (ilv && stages != 3 && !pipe) return launch_conv3_dma<TH, TW, NB, TN, 2, false, 0, true>(p, stream);
        if (loader && plain) return launch_conv3_dma<TH, TW, NB, TN, 2, false, 0, false, true>(p, stream);
        if (stages == 3) return pipe ? launch_conv3_dma<TH, TW, NB, TN, 3, true>(p, stream) : launch_conv3_dma<TH, TW, NB, TN, 3, false>(p, stream);
        if (pipe) return launch_conv3_dma<TH, TW, NB, TN, 2, true>(p, stream);
#else
        if constexpr (NB == 4) {
            if (planes && !p.pool_out) return launch_conv3_dma<TH, TW, NB, TN, 2, false, 1>(p, stream);
        }
#endif
    }
    ConvParamsB q = p;
    q.tiles_y = cdiv(p.Hout, TH);
    q.tiles_x = cdiv(p.Wout, TW);
    const long blocks = (long)cdiv(p.O1 + p.O2, TN) * q.tiles_y * q.tiles_x * cdiv(p.N, NB);
    auto kern = conv3_dma_kernel<TH, TW, NB, TN, NS, PIPE, LAY, ILV, LDR>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(LDR ? 320 : 256), G::LDS, stream, q);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

template <int TN, int NW = 4, int KS = 5>
int launch_conv5_ring(const ConvParamsB& p, hipStream_t stream) {
    using G = RingGeom<TN, NW, KS>;
    if constexpr (NW == 4 && TN == 128 && KS == 5) {
        // NIMG_RING_NW8=1 (A/B switch, not the product path): eight waves on a 32 x 16 tile against one three-slot weight ring.
        // Measured (profiles/r04_ring_*.txt): half the weight DMA per matrix instruction raises the clock the chip sustains
        // (1.81 -> 1.94 GHz on conv3) but the single resident workgroup loses more to its lock-step phases (MFMA pipe busy
        // 0.665 -> 0.568): 436 -> 459 us.  The four-wave form with two independent workgroups per CU stays.
        static const bool nw8 = getenv("NIMG_RING_NW8") != nullptr;
        if (nw8 && p.Hout % 32 == 0) return launch_conv5_ring<TN, 8>(p, stream);
    }
    ConvParamsB q = p;
    q.tiles_y = cdiv(p.Hout, G::TH);
    q.tiles_x = cdiv(p.Wout, G::TW);
    const long blocks = (long)(p.O1 / TN) * q.tiles_y * q.tiles_x * p.N;
    auto kern = conv5_ring_kernel<TN, false, NW, KS>;
    if constexpr (KS == 5) { if (p.in_idx) kern = conv5_ring_kernel<TN, true, NW, KS>; }
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(G::NT), G::LDS, stream, q);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

template <int KS, int STRIDE, int TH, int TW, int NB, int TN, bool INB = false, bool BUF = false, int CKT = 16>
int launch_conv_b(const ConvParamsB& p, hipStream_t stream) {
    if constexpr (KS == 1 && !BUF && CKT == 16) {          // 1x1: 64-channel K chunks when the channels allow
        static const bool no_ck64 = getenv("NIMG_NO_CK64") != nullptr;
        if (!no_ck64 && (p.C1 + p.C2) % 64 == 0 && (p.C2 == 0 || p.C1 % 64 == 0))
            return launch_conv_b<KS, STRIDE, TH, TW, NB, TN, INB, false, 64>(p, stream);
    }
    if constexpr (INB && !BUF && KS == 3 && STRIDE == 1 && CKT == 16 && (TN == 32 || TN == 64)) {
        // both tiles by LDS-DMA (conv3_dma_kernel): bf16-stored input(s) whose channel counts are whole 16-channel chunks, the
        // vector epilogue's output shapes; descriptors cover < 2 GB tensors
        static const bool no_dma = getenv("NIMG_NO_CONV3_DMA") != nullptr || getenv("NIMG_NO_BUFFER_LOADS") != nullptr;
        const long px = (long)p.N * p.H * p.W;
        const long w_bytes = (long)(p.CinP >> 4) * KS * KS * 16 * (p.O1 + p.O2) * 2;
        // where it pays (profiles/r04_k_conv3_dma_ab.txt): the two-tensor inputs of the UNet's decoder (the register prefetch has
        // no buffer-descriptor form for them) and the layers with a long K loop over small images (levels 3 - 5: -6 ... -13 %);
        // the large-image layers with 2 - 4 chunks lose 10 - 20 % to the doubled LDS footprint (two instead of four
        // workgroups per CU cover each other's prologue / epilogue)
        static const long max_hw = getenv("NIMG_CONV3_DMA_MAXHW") ? atol(getenv("NIMG_CONV3_DMA_MAXHW")) : 1024;
        const bool pays = p.C2 > 0 || (long)p.Hout * p.Wout <= max_hw;
        if (!no_dma && pays && !p.convt && !p.in_idx && (!p.pool_out || (p.flags & NIMG_POOL_ALSO)) && p.C1 % 16 == 0 && p.C2 % 16 == 0 && (p.O1 & 3) == 0 &&
            (p.O2 & 3) == 0 && px * p.C1 * 2 < (1l << 31) - 65536 && px * p.C2 * 2 < (1l << 31) - 65536 &&
            w_bytes < (1l << 31) - 65536)
            return launch_conv3_dma<TH, TW, NB, TN>(p, stream);
    }
    if constexpr (INB && !BUF && (KS == 5 || KS == 3) && STRIDE == 1 && 128 % TN == 0) {
        static const bool no_buf = getenv("NIMG_NO_BUFFER_LOADS") != nullptr;
        const int Cin = p.C1 + p.C2, Cout = p.O1 + p.O2;
        const long in_bytes = ((long)p.N * p.H * p.W * p.C1 * 2) >> (p.in_idx ? 2 : 0);
        const long w_bytes = (long)(p.CinP >> 4) * KS * KS * 16 * Cout * 2;
        if (!no_buf && p.C2 == 0 && Cin % 16 == 0 && Cout % TN == 0 && !p.convt && in_bytes < (1l << 31) - 65536 &&
            w_bytes < (1l << 31) - 65536) {
            if constexpr (KS == 3 && TH == 16 && TW == 16 && NB == 1 && TN == 64) {
                // ring form at kernel size 3 (conv5_ring_kernel<128, false, 4, 3>: 16 x 16 pixels x 128 channels, 2 x 4 fragments
                // per wave, weights by LDS-DMA one kernel row at a time): the big 3x3 layers - the codec's 128 -> 128 residual
                // blocks at 64 x 64 - where the generic kernel's 2 x 2 fragment block reads as many operands as it multiplies.
                // OPT-IN (NIMG_CONV3_RING_MIN=<workgroups>, default off): stand-alone the 128 -> 128 layer gains 4 - 7 % at B = 48 ... 80
                // (profiles/r04_p_conv3_ring_codec.txt), but the config-3 step LOSES 5 % (5.02 -> 5.27 ms): two 67 KB workgroups per
                // CU leave the side streams' weight-gradient kernels no LDS to run beside them.
                static const long ring3_min = getenv("NIMG_CONV3_RING_MIN") ? atol(getenv("NIMG_CONV3_RING_MIN")) : -1;
                const long rblocks = (long)(Cout / 128) * cdiv(p.Hout, 16) * cdiv(p.Wout, 16) * p.N;
                if (ring3_min >= 0 && Cout % 128 == 0 && p.O2 == 0 && (p.O1 & 7) == 0 && p.pad_t == 1 && p.pad_l == 1 && p.Hout == p.H &&
                    p.Wout == p.W && !p.pool_out && !p.in_idx && rblocks >= ring3_min)
                    return launch_conv5_ring<128, 4, 3>(p, stream);
            }
            if constexpr (KS == 5 && TH == 16 && TW == 16 && NB == 1 && TN == 64) {
                static const bool no_ring = getenv("NIMG_NO_CONV5_RING") != nullptr;
                static const bool no_ring64 = getenv("NIMG_NO_CONV5_RING64") != nullptr;
                // the ring kernels' epilogue knows bias / activation / mask / pooling only: a second bf16 copy (out1b), a
                // residual or a layout flag stays with the generic kernel, whose epilogue writes them
                const bool plain_epi = !p.res && !p.out1b && !(p.flags & (NIMG_D2S_OUT | NIMG_S2D_OUT | NIMG_COPY_LRELU));
                if (!no_ring && plain_epi && p.O2 == 0 && p.pad_t == 2 && p.pad_l == 2 && p.Hout == p.H && p.Wout == p.W) {
                    static const bool tn64 = getenv("NIMG_RING_TN64") != nullptr;      // A/B: 512 pixels x 64 channels per workgroup
                    if (Cout % 128 == 0 && !(tn64 && p.Hout >= 32)) return launch_conv5_ring<128>(p, stream);
                    if (!no_ring64 && p.Hout >= 32) return launch_conv5_ring<64>(p, stream);
                }
            }
            return launch_conv_b<KS, STRIDE, TH, TW, NB, TN, true, true>(p, stream);
        }
    }
    constexpr int THH = (TH - 1) * STRIDE + KS, TWH = (TW - 1) * STRIDE + KS;
    constexpr bool PLANAR = (STRIDE == 1 && TW == 16 && NB == 1 && KS == 5);   // 3x3: the extra registers cost a wave per SIMD
#ifdef NIMG_CK32
    constexpr int CKH = CKT != 16 ? CKT / 8 : ((KS == 3 && !INB && !BUF) ? 4 : 2);     // the kernel's K chunk in 8-channel slots
#else
    constexpr int CKH = CKT / 8;
#endif
    constexpr size_t a_entries = PLANAR ? (size_t)2 * THH * 32 : (size_t)NB * THH * TWH * CKH;
    constexpr size_t lds_tiles = (a_entries + (size_t)KS * KS * TN * CKH) * sizeof(uint4);
    constexpr size_t lds_epi = (size_t)4 * 32 * (TN + EPI_PAD) * sizeof(float);
    constexpr size_t lds = lds_tiles > lds_epi ? lds_tiles : lds_epi;
    ConvParamsB q = p;
    q.tiles_y = cdiv(p.Hout, TH);
    q.tiles_x = cdiv(p.Wout, TW);
    const long blocks = (long)cdiv(p.O1 + p.O2, TN) * q.tiles_y * q.tiles_x * cdiv(p.N, NB) * (p.convt ? 4 : 1);
    auto kern = conv_fwd_bf16_kernel<KS, STRIDE, TH, TW, NB, TN, INB, BUF, false, CKT>;
    if (p.in_idx) {                     // the input is a pooled tensor + arg-max bytes: only the buffer-load variants un-pool
        if constexpr (INB && BUF && STRIDE == 1 && CKT == 16) kern = conv_fwd_bf16_kernel<KS, STRIDE, TH, TW, NB, TN, true, true, true>;
        else return NIMG_ERR_ARG;
    }
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, q);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

template <int KS, int STRIDE, bool INB>
int dispatch_b_t(const ConvParamsB& p, hipStream_t s) {
    const int Cout = p.O1 + p.O2;
    const bool small = (p.Hout <= 8 && p.Wout <= 8);
    const long blocks64 = (long)cdiv(Cout, 64) * cdiv(p.Hout, small ? 8 : 16) * cdiv(p.Wout, small ? 8 : 16) *
                          cdiv(p.N, small ? 4 : 1);
    // few workgroups = few co-resident ones per CU to cover each other's staging latency: below the threshold the 32-channel
    // tile (twice the workgroups) wins (NIMG_TN32_BELOW overrides it for A/B runs)
    static const long tn32_below = getenv("NIMG_TN32_BELOW") ? atol(getenv("NIMG_TN32_BELOW")) : 384;
    const bool tn32 = (Cout <= 32) || (blocks64 < 512 && Cout % 64 != 0) || (blocks64 < tn32_below);
    if (small)
        return tn32 ? launch_conv_b<KS, STRIDE, 8, 8, 4, 32, INB>(p, s) : launch_conv_b<KS, STRIDE, 8, 8, 4, 64, INB>(p, s);
    // narrow outputs (Cout <= 32) of big images: a 32x16-pixel tile keeps 64 accumulator registers per wave (4 x 1
    // fragments) and stages a third fewer bytes per pixel than 16x16
    if constexpr (STRIDE == 1 && KS == 5) {
        if constexpr (INB) {                // ring form for exactly 32 output channels (FAN conv2 input gradient)
            static const bool no_ring32 = getenv("NIMG_NO_CONV5_RING32") != nullptr || getenv("NIMG_NO_CONV5_RING") != nullptr ||
                                          getenv("NIMG_NO_BUFFER_LOADS") != nullptr;
            const long in_bytes = ((long)p.N * p.H * p.W * p.C1 * 2) >> (p.in_idx ? 2 : 0);
            if (!no_ring32 && !p.res && !p.out1b && !(p.flags & NIMG_COPY_LRELU) && Cout == 32 && p.O2 == 0 && p.C2 == 0 && p.C1 % 16 == 0 && !p.convt && p.pad_t == 2 && p.pad_l == 2 &&
                p.Hout == p.H && p.Wout == p.W && p.Hout >= 32 && in_bytes < (1l << 31) - 65536)
                return launch_conv5_ring<32>(p, s);
        }
        if (Cout <= 32 && !p.pool_out && p.Hout % 32 == 0 && (long)cdiv(p.Hout, 32) * cdiv(p.Wout, 16) * p.N >= 2048)
            return launch_conv_b<KS, STRIDE, 32, 16, 1, 32, INB>(p, s);
    }
    return tn32 ? launch_conv_b<KS, STRIDE, 16, 16, 1, 32, INB>(p, s) : launch_conv_b<KS, STRIDE, 16, 16, 1, 64, INB>(p, s);
}

template <int KS, int STRIDE>
int dispatch_b(const ConvParamsB& p, hipStream_t s) {
    if constexpr (STRIDE == 1) {          // bf16-stored inputs: the stride-1 layers (FAN, UNet) ...
        if (p.flags & NIMG_BF16_IN) return dispatch_b_t<KS, STRIDE, true>(p, s);
    } else if constexpr (KS == 2) {       // ... and for the 2x2 / stride-2 form (input gradient of the UNet's Conv2DTranspose)
        if (p.flags & NIMG_BF16_IN) return dispatch_b_t<KS, STRIDE, true>(p, s);
    } else {
        if (p.flags & NIMG_BF16_IN) return NIMG_ERR_ARG;
    }
    return dispatch_b_t<KS, STRIDE, false>(p, s);
}

// ------------------------------------------------------------------------------------------------------------------
struct WgradParamsB {
    const float* in1;
    const float* in2;
    const float* dz;
    const unsigned char* dz_idx;   // optional (packed kernel): dz is the POOLED gradient (Hout/2 x Wout/2) of a fused
                                   // conv+pool layer and dz_idx its arg-max bytes - the 2x2 un-pooling happens while staging
    float* partial;
    float* db_partial;
    int C1, C2, Cout;
    int N, H, W, Hout, Wout, pad_t, pad_l;
    int tiles_y, tiles_x, splits, work_per_split, pad_mode;
    int flags;                     // NIMG_BF16_IN: in1 (and in2) hold bf16; NIMG_BF16_DZ: dz holds bf16
    // in-kernel finish of the split-K sums by the last-arriving workgroup of a dw tile (common.h ticket_finish); null: slabs only
    unsigned* tickets;
    float* dw;
    float* db;
    int group, accumulate;
    nimg::ReduceEntry pre;         // the reduction the PREVIOUS weight gradient of this stream owes (chained mode), or empty
};

constexpr int B_TH = 8, B_TW = 16, B_CI = 32, B_CO = 64;

template <int KS, int CINP, int NI, int ZMODE>
__global__ void conv_wgrad_packed_bf16_kernel(const WgradParamsB p);      // defined with the FAN front-end kernels below

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// ds_read_b64_tr_b16 (gfx950 LDS transpose read).  Measured semantics (tools/probe/tr_probe.hip): inside each 16-lane
// group lane g supplies the 8-byte-aligned address of 4 contiguous bf16; the 16 addresses are read as a 4 x 16 block
// (row = g >> 2, 4-column group = g & 3) and lane g receives COLUMN g of that block: element j = block[j][g].
// With a pixel-major [pixel][channel] tile this hands every lane 4 consecutive PIXELS of its own channel - the K-major
// fragment the weight-gradient GEMM needs - without any transposed copy in LDS.
__device__ __forceinline__ bf16x8 tr_read8(const unsigned char* p0, const unsigned char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p1);
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return *reinterpret_cast<bf16x8*>(&r);
}

constexpr int B_ZS = 192;      // dz tile row stride in bytes (64 co bf16 = 128 B, padded so 4 rows hit 4 bank quarters)

// NW waves share the taps: 4 for small kernels; 8 for 5x5, where 4 waves would each pin 7 taps x 32 = 224 accumulator
// registers (one wave per SIMD, nothing to hide LDS / barrier latency behind) - with 8 it is 4 taps = 128, two per SIMD.
// TH = output rows per staged tile (8, or 16 for 5x5: the per-tile staging overhead - ~500 instructions of address
// arithmetic, converts and LDS writes - is then amortised over twice as many MFMAs)
// UNP (with DZB): dz is the POOLED gradient (N, Hout/2, Wout/2, Cout) bf16 and p.dz_idx its arg-max bytes; the 2x2 un-pooling
// happens while the dz tile is staged (unp_route: packed byte masks), the full-resolution gradient never exists in HBM.
// NCO = 32-channel output fragments per workgroup: 2 (a 64-wide dz tile) or 1 for layers with Cout <= 32 (UNet level 1), where
// half of the 64-wide tile would be zeros: half the MFMAs, 4 instead of 5 operand reads per pixel row, and the unpadded 64-byte
// tile rows already spread four consecutive pixels over the four bank quarters.
// PAIR (3x3, stride 1, 8 x 8 images - the UNet's bottleneck level): the 8 x 16 tile would be half outside the image (every second
// matrix instruction multiplying zeros).  A tile is then TWO images side by side: columns 0 - 7 = image 2 u, 8 - 15 = image 2 u + 1,
// each with its own zero halo in the input tile ([10][2 x 10] pixels) - the lanes of the second K half read 2 pixels further on.
template <int KS, int STRIDE, int NW, bool INB, bool DZB, int TH, bool UNP = false, int NCO = 2, bool PAIR = false>
__global__ __launch_bounds__(NW * 64, 2) void conv_wgrad_bf16_kernel(const WgradParamsB p) {
    nimg::reduce_entry_inline(p.pre);
    constexpr int TCO = 32 * NCO, ZS = NCO == 2 ? B_ZS : 64, ZI = 4 * NCO;      // dz tile: channels, row stride, 16-byte items per pixel
    static_assert(NCO == 1 || NCO == 2, "one or two output fragments");
    static_assert(!UNP || NCO == 2, "un-pooling dz: 64-wide tile");
    static_assert(!UNP || (DZB && STRIDE == 1), "un-pooling dz: bf16-stored pooled gradient, stride 1");
    constexpr int TAPS = KS * KS, NT = (TAPS + NW - 1) / NW, NTHR = NW * 64;
    static_assert(!PAIR || (KS == 3 && STRIDE == 1 && TH == 8 && INB && DZB && !UNP), "image pairs: the 3x3 layers over 8 x 8 bf16 images");
    constexpr int THH = (TH - 1) * STRIDE + KS, TWH = PAIR ? 20 : (B_TW - 1) * STRIDE + KS;
    constexpr int NPIXH = THH * TWH, NPIX = TH * B_TW;
    static_assert(STRIDE == 1 || STRIDE == 2, "stride");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* sI = smem_raw;                   // [NPIXH][32 ci] bf16, 64 B per pixel
    unsigned char* sZ = smem_raw + NPIXH * 64;      // [NPIX][TCO co] bf16, ZS bytes per pixel
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = p.C1 + p.C2;
    const int cib = (Cin + B_CI - 1) / B_CI, cob = (p.Cout + TCO - 1) / TCO;
    int bid = xcd_order(blockIdx.x);
    const int ci0 = (bid % cib) * B_CI;
    bid /= cib;
    const int co0 = (bid % cob) * TCO;
    const int split = bid / cob;
    const int half = lane >> 5, g = lane & 15, sub = (lane >> 4) & 1;

    f32x16 acc[NT][NCO];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ni = 0; ni < NCO; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[t][ni][j] = 0.0f;
    const int tiles = p.tiles_y * p.tiles_x;
    const int work_total = PAIR ? (p.N + 1) / 2 : p.N * tiles;       // < 2^31 (checked by the entry point); PAIR: image pairs
    const int w_begin = split * p.work_per_split;
    const int w_end = min(work_total, w_begin + p.work_per_split);
    const bool do_bias = p.db_partial && ci0 == 0;
    float bacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // per-lane constant parts of the transpose-read addresses
    const int a_lane = ((half * (PAIR ? 10 : 8) + (g >> 2)) * STRIDE) * 64 + (sub * 16 + (g & 3) * 4) * 2;   // + pixel terms
    const int z_lane = (half * 8 + (g >> 2)) * ZS + (sub * 16 + (g & 3) * 4) * 2;
    // async-stage split: tile t+1 travels HBM -> registers while tile t is multiplied
    constexpr int IP = (NPIXH * 4 + NTHR - 1) / NTHR, ZP = (NPIX * ZI) / NTHR;
    static_assert((NPIX * ZI) % NTHR == 0 && NTHR % ZI == 0, "dz tile must divide over the threads");
    float4 preI[IP][2], preZ[ZP][2];
    uint2 preZK[UNP ? ZP : 1];
    auto fetch = [&](int wk_) {
        const int n_ = (int)(wk_ / tiles), tile_ = (int)(wk_ % tiles);
        const int ty_ = (tile_ / p.tiles_x) * TH, tx_ = (tile_ % p.tiles_x) * B_TW;
        const int iy_ = ty_ * STRIDE - p.pad_t, ix_ = tx_ * STRIDE - p.pad_l;
#pragma unroll
        for (int q = 0; q < IP; ++q) {
            const int item = tid + q * NTHR;
            const int pix = item >> 2, c = ci0 + (item & 3) * 8;
            int gy = iy_ + pix / TWH, gx = ix_ + pix % TWH;
            int ni_ = n_;
            if constexpr (PAIR) {                    // wk_ = image pair: halo columns 0 - 9 image 2 wk_, 10 - 19 image 2 wk_ + 1
                const int hx = pix % TWH;
                ni_ = 2 * wk_ + (hx >= 10 ? 1 : 0);
                gx = (hx >= 10 ? hx - 10 : hx) - 1;
                gy = pix / TWH - 1;
            }
            preI[q][0] = preI[q][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (item < NPIXH * 4 && c < Cin && (!PAIR || ni_ < p.N) && map_coord(gy, p.H, p.pad_mode) && map_coord(gx, p.W, p.pad_mode)) {
                const long pixoff = ((long)ni_ * p.H + gy) * p.W + gx;
                if constexpr (INB) {                 // C1 % 8 == 0, C2 % 8 == 0 (entry point): the 8 channels are one 16-byte load
                    const __bf16* src = c < p.C1 ? reinterpret_cast<const __bf16*>(p.in1) + pixoff * p.C1 + c
                                                 : reinterpret_cast<const __bf16*>(p.in2) + pixoff * p.C2 + (c - p.C1);
                    preI[q][0] = *reinterpret_cast<const float4*>(src);
                } else {
                    const float* src = c < p.C1 ? p.in1 + pixoff * p.C1 + c : p.in2 + pixoff * p.C2 + (c - p.C1);
                    preI[q][0] = *reinterpret_cast<const float4*>(src);
                    if (c + 4 < Cin) preI[q][1] = *reinterpret_cast<const float4*>(src + 4);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < ZP; ++q) {
            const int item = tid + q * NTHR;
            const int pix = item / ZI, c = co0 + (item % ZI) * 8;
            int oy = ty_ + pix / B_TW, ox = tx_ + pix % B_TW, nz_ = n_;
            if constexpr (PAIR) {
                nz_ = 2 * wk_ + ((pix % B_TW) >> 3);
                ox = pix & 7;
                oy = pix / B_TW;
            }
            preZ[q][0] = preZ[q][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (UNP) preZK[q] = make_uint2(0xffffffffu, 0xffffffffu);
            if (oy < p.Hout && ox < p.Wout && c < p.Cout && (!PAIR || nz_ < p.N)) {
                const long zo = UNP ? (((long)n_ * (p.Hout >> 1) + (oy >> 1)) * (p.Wout >> 1) + (ox >> 1)) * p.Cout + c
                                    : (((long)nz_ * p.Hout + oy) * p.Wout + ox) * p.Cout + c;
                if constexpr (UNP) preZK[q] = *reinterpret_cast<const uint2*>(p.dz_idx + zo);
                if constexpr (DZB) {                 // Cout % 8 == 0 (entry point)
                    preZ[q][0] = *reinterpret_cast<const float4*>(reinterpret_cast<const __bf16*>(p.dz) + zo);
                } else {
                    preZ[q][0] = *reinterpret_cast<const float4*>(p.dz + zo);
                    if (c + 4 < p.Cout) preZ[q][1] = *reinterpret_cast<const float4*>(p.dz + zo + 4);
                }
            }
        }
    };
    if (w_begin < w_end) fetch(w_begin);
    for (int wk = w_begin; wk < w_end; ++wk) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < IP; ++q) {
            const int item = tid + q * NTHR;
            if (item < NPIXH * 4) {
                uint4 packed;
                if constexpr (INB) {
                    packed = *reinterpret_cast<const uint4*>(&preI[q][0]);
                } else {
                    const float f[8] = {preI[q][0].x, preI[q][0].y, preI[q][0].z, preI[q][0].w,
                                        preI[q][1].x, preI[q][1].y, preI[q][1].z, preI[q][1].w};
                    const bf16x8 b = pack8(f);
                    packed = *reinterpret_cast<const uint4*>(&b);
                }
                *reinterpret_cast<uint4*>(sI + (item >> 2) * 64 + (item & 3) * 16) = packed;
            }
        }
#pragma unroll
        for (int q = 0; q < ZP; ++q) {
            const int item = tid + q * NTHR;
            float f[8];
            bf16x8 b;
            if constexpr (DZB) {
                if constexpr (UNP) {                 // route: keep a channel iff this pixel was its window's arg-max
                    const int pix_ = item / ZI;      // tile origin (ty, tx) is even: the window position is the pixel's parity
                    const unsigned pos = (unsigned)((((pix_ / B_TW) & 1) << 1) | ((pix_ % B_TW) & 1));
                    const uint4 routed = unp_route(*reinterpret_cast<const uint4*>(&preZ[q][0]), preZK[q].x, preZK[q].y, pos);
                    b = *reinterpret_cast<const bf16x8*>(&routed);
                } else {
                    b = *reinterpret_cast<const bf16x8*>(&preZ[q][0]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (float)b[e];
            } else {
                f[0] = preZ[q][0].x; f[1] = preZ[q][0].y; f[2] = preZ[q][0].z; f[3] = preZ[q][0].w;
                f[4] = preZ[q][1].x; f[5] = preZ[q][1].y; f[6] = preZ[q][1].z; f[7] = preZ[q][1].w;
                b = pack8(f);
            }
            if (do_bias) {                      // fused bias gradient in float32: this thread always owns channels q*8..
#pragma unroll
                for (int e = 0; e < 8; ++e) bacc[e] += f[e];
            }
            *reinterpret_cast<uint4*>(sZ + (item / ZI) * ZS + (item % ZI) * 16) = *reinterpret_cast<const uint4*>(&b);
        }
        __syncthreads();
        if (wk + 1 < w_end) fetch(wk + 1);
        // KS == 1 has one tap: the waves share the tile's pixel rows instead (each keeps a partial of the same 32 x 64 block,
        // folded through LDS behind the loop) - with the tap split three of the four waves had nothing to multiply
#pragma unroll 1
        for (int r = (KS == 1 ? wave : 0); r < TH; r += (KS == 1 ? NW : 1)) {
            const unsigned char* zr = sZ + (r * B_TW) * ZS + z_lane;
            bf16x8 bfr[NCO];
#pragma unroll
            for (int ni = 0; ni < NCO; ++ni) bfr[ni] = tr_read8(zr + 64 * ni, zr + 4 * ZS + 64 * ni);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int tap = KS == 1 ? 0 : wave + NW * t;
                if (tap < TAPS) {
                    const unsigned char* ir = sI + ((r * STRIDE + tap / KS) * TWH + (tap % KS)) * 64 + a_lane;
                    const bf16x8 a = tr_read8(ir, ir + 4 * STRIDE * 64);
#pragma unroll
                    for (int ni = 0; ni < NCO; ++ni)
                        acc[t][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfr[ni], acc[t][ni], 0, 0, 0);
                }
            }
        }
    }
    if (do_bias) {                              // thread t holds channels (t % ZI) * 8 .. + 7: reduce the NTHR / ZI owners
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_raw);
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = bacc[e];
        __syncthreads();
        if (tid < TCO && co0 + tid < p.Cout) {
            float sum = 0.f;
            for (int o = 0; o < NTHR / ZI; ++o) sum += red[(o * ZI + (tid >> 3)) * 8 + (tid & 7)];
            p.db_partial[(long)split * p.Cout + co0 + tid] = sum;
        }
    }
    if constexpr (KS == 1) {                    // fold the waves' row partials: waves 1.. park theirs in LDS, wave 0 adds in order
        static_assert(NCO == 2 && (NW - 1) * 2 * 16 * 64 * 4 <= NPIXH * 64 + NPIX * B_ZS, "fold scratch fits the tiles");
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_raw);
        if (wave > 0) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int j = 0; j < 16; ++j) red[(((wave - 1) * 2 + ni) * 16 + j) * 64 + lane] = acc[0][ni][j];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int w = 1; w < NW; ++w)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[0][ni][j] += red[(((w - 1) * 2 + ni) * 16 + j) * 64 + lane];
        }
    }
    float* slab = p.partial + (long)split * TAPS * Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tap = KS == 1 ? 0 : wave + NW * t;
        if (tap >= TAPS || (KS == 1 && wave > 0)) continue;
#pragma unroll
        for (int ni = 0; ni < NCO; ++ni) {
            const int co = co0 + ni * 32 + (lane & 31);
            if (co >= p.Cout) continue;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int ci = ci0 + (j & 3) + 8 * (j >> 2) + 4 * half;
                if (ci < Cin) slab[((long)tap * Cin + ci) * p.Cout + co] = acc[t][ni][j];
            }
        }
    }
    if (p.tickets == nullptr) return;
    // ---- the last workgroup of this (ci block, co block) tile to get here sums the tile over the splits, in a fixed order
    TicketJob job;
    job.cnt = p.tickets + (long)(xcd_order(blockIdx.x) % (cib * cob)) * ticket_words_per_tile_dev(p.splits, p.group);
    job.slab[0] = p.partial; job.stride[0] = (long)TAPS * Cin * p.Cout; job.dst[0] = p.dw;
    job.slab[1] = p.db_partial; job.stride[1] = p.Cout; job.dst[1] = p.db;
    job.splits = p.splits; job.group = p.group; job.accumulate = p.accumulate;
    const int rows = min(B_CI, Cin - ci0), c4n = min(TCO, p.Cout - co0) >> 2, Cout = p.Cout;
    const int witems = TAPS * rows * c4n;
    const int items = witems + ((p.db_partial && ci0 == 0) ? c4n : 0);
    ticket_finish<NTHR>(job, split, items, [=](int it) {
        TicketItem m;
        if (it >= witems) { m.which = 1; m.off = co0 + (it - witems) * 4; return m; }
        const int c4 = it % c4n, row = it / c4n;                 // row = tap * rows + r
        m.which = 0;
        m.off = ((long)(row / rows) * Cin + ci0 + row % rows) * Cout + co0 + c4 * 4;
        return m;
    }, reinterpret_cast<unsigned*>(smem_raw));
}

int splits_for(int cin, int cout, int n, int hout, int wout, int th = B_TH, int target_blocks = 512) {
    const long blocks_io = (long)cdiv(cin, B_CI) * cdiv(cout, B_CO);
    const long work = (long)n * cdiv(hout, th) * cdiv(wout, B_TW);
    long splits = (target_blocks + blocks_io - 1) / blocks_io;
    if (splits > work) splits = work;
    if (splits < 1) splits = 1;
    const long wps = (work + splits - 1) / splits;
    return (int)((work + wps - 1) / wps);
}

}  // namespace

extern "C" {

size_t nimg_conv_weights_bf16_bytes(int ks_h, int ks_w, int cin, int cout, int mode) {
    const long rows = mode == 0 ? cout : cin, cols = mode == 0 ? cin : cout;
    return (size_t)ks_h * ks_w * rows * ((cols + 15) / 16 * 16) * 2;
}

int nimg_conv_weights_bf16(const float* w, void* wb, int ks_h, int ks_w, int cin, int cout, int mode, void* stream) {
    if (!w || !wb || ks_h <= 0 || ks_w <= 0 || cin <= 0 || cout <= 0 || mode < 0 || mode > 1) return NIMG_ERR_ARG;
    const long total = (long)nimg_conv_weights_bf16_bytes(ks_h, ks_w, cin, cout, mode) / 2;
    const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(weights_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, (__bf16*)wb,
                       ks_h * ks_w, cin, cout, mode);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_conv_weights_bf16_batch(const void* table, int n_entries, void* stream) {
    if (n_entries == 0) return NIMG_OK;
    if (!table || n_entries < 0) return NIMG_ERR_ARG;
    // 384 workgroups per entry: the launch lasts as long as its largest entry (512 x 512 x 9: 2304 tiles -> 6 per workgroup;
    // with 96 it was 24 serial tiles = 36 of the launch's 40 us at the head of every step)
    hipLaunchKernelGGL(weights_bf16_batch_kernel, dim3(384, (unsigned)n_entries), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)table);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

static int conv2d_fwd_bf16_impl(const float* in1, int c1, const float* in2, int c2, const void* wb, const float* bias,
                         float* out1, int o1, float* out2, int o2, const float* act_mask, int n, int h, int wd,
                         int ks, int stride, int pad_t, int pad_l, int pad_mode, int hout, int wout, int act,
                         float alpha, int flags, void* stream, const unsigned char* in_idx = nullptr,
                         const float* res = nullptr, void* out1b = nullptr, float* pool_out = nullptr,
                         unsigned char* pool_idx = nullptr) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!in1 || !wb || !out1 || c1 <= 0 || c2 < 0 || o1 <= 0 || o2 < 0 || n < 0 || h <= 0 || wd <= 0) return NIMG_ERR_ARG;
    if ((c2 > 0 && !in2) || (o2 > 0 && !out2) || hout <= 0 || wout <= 0 || pad_t < 0 || pad_l < 0) return NIMG_ERR_ARG;
    if (act < 0 || act > 1 || pad_mode < 0 || pad_mode > 2) return NIMG_ERR_ARG;
    if (c2 == 0 ? ((c1 % 4) || ((flags & NIMG_BF16_IN) && (c1 % 8))) : ((c1 % 8) || (c2 % 8)))
        return NIMG_ERR_ARG;                           // 8-channel staging granules (a single float32 input may end on 4)
    if (n == 0) return NIMG_OK;
    ConvParamsB p;
    p.in1 = in1; p.in2 = in2; p.wb = (const __bf16*)wb; p.bias = bias; p.out1 = out1; p.out2 = out2; p.act1 = act_mask;
    p.pool_out = pool_out; p.pool_idx = pool_idx; p.convt = 0; p.flags = flags; p.in_idx = in_idx; p.res = res;
    p.out1b = (float*)out1b;
    if (((flags & NIMG_POOL_ALSO) != 0) != (pool_out != nullptr)) return NIMG_ERR_ARG;
    if (res && (ks != 3 || stride != 1)) return NIMG_ERR_ARG;
    if ((res || out1b) && (o2 != 0 || (o1 & 3))) return NIMG_ERR_ARG;
    if (out1b && (flags & NIMG_BF16_OUT)) return NIMG_ERR_ARG;
    if ((flags & NIMG_COPY_LRELU) && (!out1b || act != 0)) return NIMG_ERR_ARG;
    if (in_idx && (!(flags & NIMG_BF16_IN) || stride != 1 || ks != 5 || (h & 1) || (wd & 1) || pad_mode != 0)) return NIMG_ERR_ARG;
    if ((flags & (NIMG_BF16_OUT | NIMG_BF16_MASK)) && ((o1 & 3) || (o2 & 3))) return NIMG_ERR_ARG;   // vector epilogue only
    if ((flags & NIMG_BF16_MASK) && o2 != 0) return NIMG_ERR_ARG;
    if ((flags & NIMG_D2S_OUT) && (ks != 3 || stride != 1 || o2 != 0 || (o1 & 15) || in_idx)) return NIMG_ERR_ARG;
    if ((flags & NIMG_S2D_OUT) && (ks != 3 || stride != 1 || o2 != 0 || (o1 & 3) || in_idx || (hout & 1) || (wout & 1) ||
                                   (flags & NIMG_D2S_OUT)))
        return NIMG_ERR_ARG;
    p.C1 = c1; p.C2 = c2; p.O1 = o1; p.O2 = o2; p.CinP = (c1 + c2 + 15) / 16 * 16;
    p.N = n; p.H = h; p.W = wd; p.Hout = hout; p.Wout = wout; p.pad_t = pad_t; p.pad_l = pad_l;
    p.tiles_y = p.tiles_x = 0; p.act = act; p.pad_mode = pad_mode; p.alpha = alpha;
    hipStream_t s = (hipStream_t)stream;
    if (stride == 1 && ks == 1) return dispatch_b<1, 1>(p, s);
    if (stride == 1 && ks == 3) return dispatch_b<3, 1>(p, s);
    if (stride == 1 && ks == 5) return dispatch_b<5, 1>(p, s);
    if (stride == 2 && ks == 2) return dispatch_b<2, 2>(p, s);
    if (stride == 2 && ks == 5) return dispatch_b<5, 2>(p, s);
    return NIMG_ERR_ARG;
}

int nimg_conv2d_fwd_bf16(const float* in1, int c1, const float* in2, int c2, const void* wb, const float* bias,
                         float* out1, int o1, float* out2, int o2, const float* act_mask, int n, int h, int wd,
                         int ks, int stride, int pad_t, int pad_l, int pad_mode, int hout, int wout, int act,
                         float alpha, void* stream) {
    return conv2d_fwd_bf16_impl(in1, c1, in2, c2, wb, bias, out1, o1, out2, o2, act_mask, n, h, wd, ks, stride, pad_t, pad_l,
                                pad_mode, hout, wout, act, alpha, 0, stream);
}

int nimg_conv2d_fwd_bf16_ex(const float* in1, int c1, const float* in2, int c2, const void* wb, const float* bias,
                            float* out1, int o1, float* out2, int o2, const float* act_mask, int n, int h, int wd,
                            int ks, int stride, int pad_t, int pad_l, int pad_mode, int hout, int wout, int act,
                            float alpha, int flags, void* stream) {
    return conv2d_fwd_bf16_impl(in1, c1, in2, c2, wb, bias, out1, o1, out2, o2, act_mask, n, h, wd, ks, stride, pad_t, pad_l,
                                pad_mode, hout, wout, act, alpha, flags, stream);
}

/* A SAME stride-1 3x3 convolution (+ bias + activation) that stores BOTH its output and the 2x2 max-pooled output, bf16 in
 * and out: the UNet's second encoder convolutions (models/pipelines.py:160-173 - the full tensor is the skip connection, the
 * pooled one the next level's input).  cin % 8 == 0, cout % 8 == 0, even h / wd > 8 (the 16x16-pixel tiles); pool_idx optional. */
int nimg_conv2d_fwd_pool_also_bf16(const float* in, int cin, const void* wb, const float* bias, float* out, float* pool_out,
                                   unsigned char* pool_idx, int cout, int n, int h, int wd, int act, float alpha, void* stream) {
#ifdef NIMG_NO_EPI8
    return NIMG_ERR_ARG;           /* A/B build without the 8-wide epilogue: the NIMG_POOL_ALSO branch lives there (ops probes with n = 0) */
#endif
    if (n == 0) return NIMG_OK;
    if (!pool_out || (h & 1) || (wd & 1) || h <= 8 || wd <= 8 || (cout & 7) || (cin & 7)) return NIMG_ERR_ARG;
    return conv2d_fwd_bf16_impl(in, cin, nullptr, 0, wb, bias, out, cout, nullptr, 0, nullptr, n, h, wd, 3, 1, 1, 1, 0, h, wd, act,
                                alpha, NIMG_BF16_IN | NIMG_BF16_OUT | NIMG_POOL_ALSO, stream, nullptr, nullptr, nullptr, pool_out,
                                pool_idx);
}

/* The input gradient of the first convolution of a UNet encoder level, written THROUGH the 2x2 max-pool in front of it
 * (models/pipelines.py:160-173 backward): dz (bf16, n x cin... see include/nimg.h). */
int nimg_conv2d_dgrad_unpool_out_bf16(const float* dz, int c1, const void* wb, const float* act, const float* skip, float* out,
                                      int cout, int n, int h, int wd, int apply_mask, float alpha, void* stream) {
#ifdef NIMG_NO_EPI8
    return NIMG_ERR_ARG;           /* A/B build without the 8-wide epilogue: the NIMG_UNPOOL_OUT branch lives there (ops probes with n = 0) */
#endif
    if (n == 0) return NIMG_OK;
    if (!dz || !wb || !act || !out || (c1 & 7) || (cout & 7) || h <= 0 || wd <= 0) return NIMG_ERR_ARG;
    if ((long)n * 4 * h * wd * cout * 2 >= (1l << 40)) return NIMG_ERR_ARG;
    ConvParamsB p;
    p.in1 = dz; p.in2 = nullptr; p.wb = (const __bf16*)wb; p.bias = nullptr; p.out1 = out; p.out2 = nullptr; p.act1 = act;
    p.pool_out = nullptr; p.pool_idx = nullptr; p.convt = 0; p.flags = NIMG_BF16_IN | NIMG_BF16_OUT | NIMG_BF16_MASK | NIMG_UNPOOL_OUT;
    p.in_idx = nullptr; p.res = skip; p.out1b = nullptr;
    p.C1 = c1; p.C2 = 0; p.O1 = cout; p.O2 = 0; p.CinP = (c1 + 15) / 16 * 16;
    p.N = n; p.H = h; p.W = wd; p.Hout = h; p.Wout = wd; p.pad_t = p.pad_l = 1;
    p.tiles_y = p.tiles_x = 0; p.act = apply_mask ? 1 : 0; p.pad_mode = 0; p.alpha = alpha;
    return dispatch_b<3, 1>(p, (hipStream_t)stream);
}

/* nimg_conv2d_fwd_bf16_ex for the layers of a residual block (models/compression.py:224-227, 240-243): `residual` (float32, the
 * shape of out1, optional) is added to the result after bias, activation and mask - net + conv(a) forward, d_net + mask * dgrad
 * backward, one pass - and `out_bf16_copy` (optional) receives the same result rounded to bf16 next to the float32 out1: the
 * residual stream keeps its exact float32 sum, its consumers read the bf16 copy (flag NIMG_COPY_LRELU: the copy holds
 * LeakyReLU(alpha) of the result - the codec feeds its first block the activation of the tensor it skips around,
 * models/compression.py:224).  At least one of the two; one float32 output with o1 % 4 == 0, the residual with 3x3 / stride 1
 * layers only (else NIMG_ERR_ARG). */
int nimg_conv2d_fwd_bf16_res(const float* in1, int c1, const void* wb, const float* bias, float* out1, int o1,
                             const float* act_mask, const float* residual, void* out_bf16_copy, int n, int h, int wd, int ks,
                             int stride, int pad_t, int pad_l, int pad_mode, int hout, int wout, int act, float alpha, int flags,
                             void* stream) {
    if (!residual && !out_bf16_copy) return NIMG_ERR_ARG;
    return conv2d_fwd_bf16_impl(in1, c1, nullptr, 0, wb, bias, out1, o1, nullptr, 0, act_mask, n, h, wd, ks, stride, pad_t, pad_l,
                                pad_mode, hout, wout, act, alpha, flags, stream, nullptr, residual, out_bf16_copy);
}

/* The same convolution on the 2x2 UN-POOLING of a pooled bf16 tensor: in_pooled (n, h/2, wd/2, c1) bf16 + in_idx arg-max bytes
 * stand for the (n, h, wd, c1) tensor that holds in_pooled[y/2][x/2][c] where in_idx[y/2][x/2][c] == 2 (y & 1) + (x & 1) and zero
 * elsewhere (the gradient MaxPool2D hands back).  Used as the input-gradient pass of the FAN's fused conv + pool layers: the
 * full-resolution gradient never exists in HBM.  5x5, stride 1, zero padding, c1 % 16 == 0, cout % 32 == 0 (else NIMG_ERR_ARG:
 * un-pool explicitly with nimg_maxpool2_unpool_ex and call nimg_conv2d_fwd_bf16_ex). */
int nimg_conv2d_fwd_bf16_unpool(const void* in_pooled, const unsigned char* in_idx, int c1, const void* wb, const float* bias,
                                float* out1, int o1, const float* act_mask, int n, int h, int wd, int ks, int pad_t, int pad_l,
                                int hout, int wout, int act, float alpha, int flags, void* stream) {
    if (!in_idx) return NIMG_ERR_ARG;
    return conv2d_fwd_bf16_impl((const float*)in_pooled, c1, nullptr, 0, wb, bias, out1, o1, nullptr, 0, act_mask, n, h, wd, ks, 1,
                                pad_t, pad_l, 0, hout, wout, act, alpha, flags | NIMG_BF16_IN, stream, in_idx);
}

/* Conv2DTranspose(cout, 2x2, stride 2) forward (pipelines.py:205) on the matrix core: four 1x1 products, one per output
 * phase (dy, dx), in a single launch.  wb = nimg_conv_weights_bf16(w, 2, 2, cin'=cout, cout'=cin, mode 1) of the Keras
 * kernel (2,2,Cout,Cin).  x (n,h,wd,cin) -> y (n,2h,2wd,cout). */
int nimg_convt2x2_fwd_bf16(const float* x, const void* wb, const float* bias, float* y, int n, int h, int wd, int cin,
                           int cout, void* stream) {
    return nimg_convt2x2_fwd_bf16_ex(x, wb, bias, y, n, h, wd, cin, cout, 0, stream);
}

/* flags: NIMG_BF16_IN = x is stored as bf16, NIMG_BF16_OUT = y is stored as bf16 (cout % 4 == 0) */
int nimg_convt2x2_fwd_bf16_ex(const float* x, const void* wb, const float* bias, float* y, int n, int h, int wd, int cin,
                              int cout, int flags, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!x || !wb || !y || n < 0 || h <= 0 || wd <= 0 || cin <= 0 || cout <= 0 || (cin % 8)) return NIMG_ERR_ARG;
    if ((flags & ~(NIMG_BF16_IN | NIMG_BF16_OUT)) || ((flags & NIMG_BF16_OUT) && (cout & 3))) return NIMG_ERR_ARG;
    ConvParamsB p;
    p.in1 = x; p.in2 = nullptr; p.wb = (const __bf16*)wb; p.bias = bias; p.out1 = y; p.out2 = nullptr; p.act1 = nullptr;
    p.pool_out = nullptr; p.pool_idx = nullptr; p.convt = 1; p.flags = flags; p.in_idx = nullptr; p.res = nullptr; p.out1b = nullptr;
    p.C1 = cin; p.C2 = 0; p.O1 = cout; p.O2 = 0; p.CinP = (cin + 15) / 16 * 16;
    p.N = n; p.H = h; p.W = wd; p.Hout = h; p.Wout = wd; p.pad_t = 0; p.pad_l = 0;
    p.tiles_y = p.tiles_x = 0; p.act = 0; p.pad_mode = 0; p.alpha = 0.f;
#ifndef NIMG_NO_EPI8
    // bf16-stored output, cout % 8 == 0 (the UNet's four layers): ONE 1x1 product with N = 4 cout columns - the input tile is
    // staged once for the four output phases instead of once per phase, a quarter of the workgroups pay prologue and epilogue -
    // and the phase is a pixel offset of the 16-byte store (NIMG_D2S_CONVT).  Same products in the same order: bit-identical.
    static const bool no_fat = getenv("NIMG_NO_CONVT_FAT") != nullptr;
    if (!no_fat && (flags & NIMG_BF16_OUT) && (cout & 7) == 0) {
        p.convt = 0; p.O1 = 4 * cout; p.flags = flags | NIMG_D2S_CONVT;
    }
#endif
    return dispatch_b<1, 1>(p, (hipStream_t)stream);
}

static int packed_splits_b(int cout, int n, int hout, int wout) {
    const long blocks_io = cdiv(cout, cout <= 32 ? 32 : 64);
    const long work = (long)n * cdiv(hout, B_TH) * cdiv(wout, B_TW);
    long splits = (1024 + blocks_io - 1) / blocks_io;
    if (splits > work) splits = work;
    if (splits < 1) splits = 1;
    const long wps = (work + splits - 1) / splits;
    return (int)((work + wps - 1) / wps);
}

size_t nimg_conv2d_wgrad_bf16_workspace_bytes(int cin, int cout, int ks_h, int ks_w, int n, int hout, int wout) {
    if (cin <= 0 || cout <= 0 || n <= 0) return 0;
    const size_t slab = (size_t)ks_h * ks_w * cin * cout * sizeof(float);
    const size_t generic = (slab + cout * sizeof(float)) * splits_for(cin, cout, n, hout, wout);
    const size_t packed = cin <= 4 ? (4 * slab + cout * sizeof(float)) * packed_splits_b(cout, n, hout, wout) : 0;
    const size_t tiny = (cin <= 4 && cout <= 4) ? nimg_internal_wgrad_tiny_bytes(ks_h, cin, cout) : 0;
    const size_t m = generic > packed ? generic : packed;
    return m > tiny ? m : tiny;
}

// deferred mode (nimg_conv2d_wgrad_bf16_deferred): the partial-sum kernel is launched, the reduction it owes is described in
// *g_defer instead of being launched (common.h ReduceEntry); one thread-local pointer, set around the call
static thread_local nimg::ReduceEntry* g_defer = nullptr;
// chained mode (nimg_conv2d_wgrad_bf16_chained): the reduction the PREVIOUS deferred weight gradient of the stream owes; the kernel
// launched by this call runs it in its prologue (conv3_wgrad_alltaps_kernel, conv_wgrad_bf16_kernel), any other path launches it
// as a separate reduction first
static thread_local const nimg::ReduceEntry* g_pre = nullptr;
static inline void finish_reduce2(const float* p1, float* d1, long n1, int splits1, const float* p2, float* d2, long n2,
                                  int splits2, int accumulate, hipStream_t s) {
    if (g_defer) nimg::fill_reduce_entry(g_defer, p1, d1, n1, splits1, p2, d2, n2, splits2, accumulate);
    else launch_reduce2(p1, d1, n1, splits1, p2, d2, n2, splits2, accumulate, s);
}

static int wgrad_bf16_impl(const float* in1, int c1, const float* in2, int c2, const float* dz,
                           const unsigned char* dz_idx, int cout, float* dw, float* db, int n, int h, int wd, int ks,
                           int stride, int pad_t, int pad_l, int pad_mode, int hout, int wout, int accumulate,
                           void* workspace, size_t workspace_bytes, int flags, void* stream) {
    if ((flags & NIMG_BF16_IN) && ((c1 & 7) || (c2 & 7))) return NIMG_ERR_ARG;        /* in1 and in2 are both bf16 then */
    if ((flags & NIMG_BF16_DZ) && (cout & 7)) return NIMG_ERR_ARG;
    if (flags && !dz_idx && c2 == 0 && c1 <= 4) return NIMG_ERR_ARG;        /* the packed / tiny kernels stage float32 */
    if (dz_idx && c1 <= 4 && (flags & ~NIMG_BF16_DZ)) return NIMG_ERR_ARG;
    if (dz_idx && c1 > 4 && (flags != (NIMG_BF16_IN | NIMG_BF16_DZ) || stride != 1 || ks != 5 || (hout & 1) || (wout & 1)))
        return NIMG_ERR_ARG;         /* un-pooling dz in the generic kernel: bf16-stored operands of the FAN's 5x5 layers */
    if (!in1 || !dz || !dw || c1 <= 0 || c2 < 0 || cout <= 0 || n <= 0 || h <= 0 || wd <= 0) return NIMG_ERR_ARG;
    if ((c2 > 0 && !in2) || hout <= 0 || wout <= 0 || !workspace || pad_mode < 0 || pad_mode > 2) return NIMG_ERR_ARG;
    const int cin = c1 + c2;
    if (workspace_bytes < nimg_conv2d_wgrad_bf16_workspace_bytes(cin, cout, ks, ks, n, hout, wout)) return NIMG_ERR_WORKSPACE;
    const nimg::ReduceEntry* pre = g_pre;          // consumed by exactly one of the paths below
    g_pre = nullptr;
    auto pre_alone = [&]() {                       // a path whose kernel cannot run it: as its own launch, now
        if (pre && pre->n1 > 0 && pre->p1 && pre->d1)
            launch_reduce2(pre->p1, pre->d1, pre->n1, pre->splits1, pre->p2, pre->d2, pre->n2, pre->splits2, pre->accumulate,
                           (hipStream_t)stream);
        pre = nullptr;
    };
    if (c2 == 0 && c1 == 3 && cout == 3 && stride == 1 && (ks == 3 || ks == 5) && hout == h && wout == wd &&
        pad_t == (ks - 1) / 2 && pad_l == pad_t && !db) {            // tiny filter: its own kernels (conv_small.hip)
        pre_alone();
        return nimg_internal_conv_wgrad_tiny(in1, dz, dw, c1, cout, n, h, wd, ks, pad_t, pad_mode, accumulate, workspace,
                                             (hipStream_t)stream, true);          // throughput mode: bf16 matrix operands
    }
    if (c2 == 0 && (c1 == 3 || c1 == 4) && stride == 1 && (ks == 3 || ks == 5)) {      // (tap, ci)-packed M dimension
        pre_alone();
        WgradParamsB q;
        q.pre = nimg::empty_reduce_entry();
        q.in1 = in1; q.in2 = nullptr; q.dz = dz; q.dz_idx = dz_idx; q.partial = (float*)workspace; q.db_partial = nullptr;
        q.flags = flags;
        q.tickets = nullptr; q.dw = nullptr; q.db = nullptr; q.group = 1; q.accumulate = accumulate;
        q.C1 = c1; q.C2 = 0; q.Cout = cout; q.N = n; q.H = h; q.W = wd; q.Hout = hout; q.Wout = wout;
        q.pad_t = pad_t; q.pad_l = pad_l; q.pad_mode = pad_mode;
        q.tiles_y = cdiv(hout, B_TH); q.tiles_x = cdiv(wout, B_TW);
        q.splits = packed_splits_b(cout, n, hout, wout);
        const long work_ = (long)n * q.tiles_y * q.tiles_x;
        q.work_per_split = (int)((work_ + q.splits - 1) / q.splits);
        const long cnt = (long)ks * ks * cin * cout;
        if (db) q.db_partial = q.partial + (size_t)4 * q.splits * cnt;
        const int ni = cout <= 32 ? 1 : 2;
        const long pblocks = (long)cdiv(cout, 32 * ni) * q.splits;
        hipStream_t s_ = (hipStream_t)stream;
        int slabs_per_wg = 4;
#define NIMG_WGPB(KS_, C_, NI_)                                                                                 \
        do {                                                                                                  \
            constexpr size_t lds_t = (size_t)((B_TH + KS_ - 1) * (B_TW + KS_ - 1) * C_ + B_TH * B_TW * 32 * NI_) * \
                                     sizeof(float);                                                           \
            constexpr int MF_ = (KS_ * KS_ + 32 / C_ - 1) / (32 / C_);                                        \
            constexpr bool FOLD_ = MF_ * NI_ <= 2;                  /* as in the kernel */                        \
            constexpr size_t lds_f = FOLD_ ? (size_t)3 * MF_ * NI_ * 16 * 64 * sizeof(float) : 0;             \
            constexpr size_t lds = lds_t > lds_f ? lds_t : lds_f;                                             \
            slabs_per_wg = FOLD_ ? 1 : 4;                                                                     \
            if (!q.dz_idx)                                                                                    \
                hipLaunchKernelGGL((conv_wgrad_packed_bf16_kernel<KS_, C_, NI_, 0>), dim3((unsigned)pblocks),     \
                                   dim3(256), lds, s_, q);                                                    \
            else if (q.flags & NIMG_BF16_DZ)                                                                  \
                hipLaunchKernelGGL((conv_wgrad_packed_bf16_kernel<KS_, C_, NI_, 2>), dim3((unsigned)pblocks),     \
                                   dim3(256), lds, s_, q);                                                    \
            else                                                                                              \
                hipLaunchKernelGGL((conv_wgrad_packed_bf16_kernel<KS_, C_, NI_, 1>), dim3((unsigned)pblocks),     \
                                   dim3(256), lds, s_, q);                                                    \
        } while (0)
        if (ks == 5 && c1 == 3) { if (ni == 1) NIMG_WGPB(5, 3, 1); else NIMG_WGPB(5, 3, 2); }
        else if (ks == 5) { if (ni == 1) NIMG_WGPB(5, 4, 1); else NIMG_WGPB(5, 4, 2); }
        else if (c1 == 3) { if (ni == 1) NIMG_WGPB(3, 3, 1); else NIMG_WGPB(3, 3, 2); }
        else { if (ni == 1) NIMG_WGPB(3, 4, 1); else NIMG_WGPB(3, 4, 2); }
#undef NIMG_WGPB
        NIMG_CHECK_LAUNCH();
        finish_reduce2((const float*)workspace, dw, cnt, slabs_per_wg * q.splits, db ? (const float*)q.db_partial : nullptr, db,
                       (long)cout, q.splits, accumulate, s_);
        NIMG_CHECK_LAUNCH();
        return NIMG_OK;
    }
    if ((c1 % 4) || (c2 % 4) || (cout % 4) || (c2 > 0 && (c1 % 8))) { pre_alone(); return NIMG_ERR_ARG; }
    WgradParamsB p;
    p.pre = nimg::empty_reduce_entry();
    p.in1 = in1; p.in2 = in2; p.dz = dz; p.dz_idx = dz_idx; p.partial = (float*)workspace; p.db_partial = nullptr;
    p.flags = flags;
    p.tickets = nullptr; p.dw = dw; p.db = db; p.group = 1; p.accumulate = accumulate;
    p.C1 = c1; p.C2 = c2; p.Cout = cout; p.N = n; p.H = h; p.W = wd; p.Hout = hout; p.Wout = wout;
    p.pad_t = pad_t; p.pad_l = pad_l; p.pad_mode = pad_mode;
    // arrival counters for the in-kernel finish of the generic kernel's split-K sums (set once p.splits is final)
    auto want_tickets = [&](WgradParamsB& w) {
        if (g_defer || (((uintptr_t)dw | (uintptr_t)db) & 15)) return;
        w.group = ticket_group(w.splits);
        w.tickets = nimg_internal_tickets((hipStream_t)stream, (size_t)cdiv(cin, B_CI) * cdiv(cout, B_CO) * ticket_words_per_tile(w.splits));
    };
    const int th = (stride == 1 && ks == 5) ? 16 : B_TH;
    p.tiles_y = cdiv(hout, th); p.tiles_x = cdiv(wout, B_TW);
    // the 8-wave 5x5 kernel runs ONE workgroup per CU: 256 workgroups are one full round, and half the slabs to write and reduce
    static const int wg5_env = getenv("NIMG_WGRAD5_BLOCKS") ? atoi(getenv("NIMG_WGRAD5_BLOCKS")) : 256;
    const int wg5 = wg5_env < 32 ? 32 : (wg5_env > 512 ? 512 : wg5_env);        // 512 = what the workspace bound assumes
    p.splits = splits_for(cin, cout, n, hout, wout, th, (stride == 1 && ks == 5) ? wg5 : 512);   // <= splits_for(.., B_TH): the workspace bound holds
    const long work = (long)n * p.tiles_y * p.tiles_x;
    p.work_per_split = (int)((work + p.splits - 1) / p.splits);
    const long count = (long)ks * ks * cin * cout;
    hipStream_t s = (hipStream_t)stream;
    if (dz_idx && ks == 5 && stride == 1 && c2 == 0 && pad_t == 2 && pad_l == 2 && hout == h && wout == wd && pad_mode == 0) {
        // the FAN's conv2..4: all 25 taps in one wave (wgrad5.hip); slabs laid out inside the same workspace bound
        pre_alone();
        const int max_slabs = splits_for(cin, cout, n, hout, wout);
        float* dbp = db ? (float*)workspace + (size_t)max_slabs * count : nullptr;
        const int slabs = nimg_internal_wgrad5_alltaps(in1, cin, dz, dz_idx, cout, (float*)workspace, dbp, n, h, wd, max_slabs, s);
        if (slabs < 0) return NIMG_ERR_LAUNCH;
        if (slabs > 0) {
            finish_reduce2((const float*)workspace, dw, count, slabs, dbp, db, (long)cout, slabs, accumulate, s);
            NIMG_CHECK_LAUNCH();
            return NIMG_OK;
        }
    }
    if (!dz_idx && ks == 3 && stride == 1 && pad_t == 1 && pad_l == 1 && hout == h && wout == wd && pad_mode == 0 &&
        flags == (NIMG_BF16_IN | NIMG_BF16_DZ)) {
        // the UNet's 3x3 layers with bf16-stored tensors: all 9 taps in one wave, double-buffered tiles (wgrad3.hip)
        const int max_slabs = splits_for(cin, cout, n, hout, wout);
        float* dbp = db ? (float*)workspace + (size_t)max_slabs * count : nullptr;
        const int slabs = nimg_internal_wgrad3_alltaps(in1, c1, in2, c2, dz, cout, (float*)workspace, dbp, n, h, wd, max_slabs, s,
                                                       g_defer ? nullptr : dw, db, accumulate, pre);
        if (slabs != 0) pre = nullptr;                        // launched: its prologue runs the chained reduction
        if (slabs == -1) return NIMG_ERR_LAUNCH;
        if (slabs < -1) return NIMG_OK;                       // finished in the kernel by the last-arriving workgroups
        if (slabs > 0) {
            finish_reduce2((const float*)workspace, dw, count, slabs, dbp, db, (long)cout, slabs, accumulate, s);
            NIMG_CHECK_LAUNCH();
            return NIMG_OK;
        }
    }
    // 8 x 8 images (the UNet's bottleneck level): tiles of two images side by side instead of 8 x 16 tiles that are half empty
    static const bool no_pair = getenv("NIMG_NO_WGRAD_PAIR8") != nullptr;
    if (!no_pair && !dz_idx && ks == 3 && stride == 1 && h == 8 && wd == 8 && hout == 8 && wout == 8 && pad_t == 1 && pad_l == 1 &&
        pad_mode == 0 && flags == (NIMG_BF16_IN | NIMG_BF16_DZ) && n >= 2) {
        const long pairs = (n + 1) / 2;
        long sp = p.splits < pairs ? p.splits : pairs;
        const long wps = (pairs + sp - 1) / sp;
        sp = (pairs + wps - 1) / wps;
        p.splits = (int)sp; p.work_per_split = (int)wps; p.tiles_y = p.tiles_x = 1;
        if (db) p.db_partial = p.partial + (size_t)p.splits * count;
        const long pblocks = (long)cdiv(cin, B_CI) * cdiv(cout, B_CO) * p.splits;
        constexpr size_t lds = (size_t)10 * 20 * 64 + (size_t)B_TH * B_TW * B_ZS;
        auto k = conv_wgrad_bf16_kernel<3, 1, 4, true, true, B_TH, false, 2, true>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        want_tickets(p);
        if (pre) { p.pre = *pre; pre = nullptr; }
        hipLaunchKernelGGL(k, dim3((unsigned)pblocks), dim3(256), lds, s, p);
        NIMG_CHECK_LAUNCH();
        if (p.tickets) return NIMG_OK;
        finish_reduce2((const float*)workspace, dw, count, p.splits, db ? (const float*)p.db_partial : nullptr, db, (long)cout,
                       p.splits, accumulate, s);
        NIMG_CHECK_LAUNCH();
        return NIMG_OK;
    }
    if (db) p.db_partial = p.partial + (size_t)p.splits * count;
    const long blocks = (long)cdiv(cin, B_CI) * cdiv(cout, B_CO) * p.splits;
    want_tickets(p);
    if (pre) { p.pre = *pre; pre = nullptr; }
#define NIMG_WGB1(KS_, ST_, NW_, INB_, DZB_, TH_)                                                               \
    do {                                                                                                      \
        constexpr int THH = (TH_ - 1) * ST_ + KS_, TWH = (B_TW - 1) * ST_ + KS_;                              \
        constexpr size_t lds_t = (size_t)THH * TWH * 64 + (size_t)TH_ * B_TW * B_ZS;                          \
        constexpr size_t lds = lds_t > (size_t)NW_ * 64 * 8 * 4 ? lds_t : (size_t)NW_ * 64 * 8 * 4;          \
        auto k = conv_wgrad_bf16_kernel<KS_, ST_, NW_, INB_, DZB_, TH_>;                                      \
        if constexpr (KS_ == 3 && ST_ == 1) {                  /* narrow outputs: a 32-wide dz tile */            \
            static const bool no_narrow = getenv("NIMG_NO_NARROW_WGRAD") != nullptr;                          \
            if (!no_narrow && p.Cout <= 32 && !p.dz_idx) k = conv_wgrad_bf16_kernel<KS_, ST_, NW_, INB_, DZB_, TH_, false, 1>;  \
        }                                                                                                     \
        if (p.dz_idx) {                                                                                       \
            if constexpr (DZB_ && ST_ == 1 && KS_ == 5) k = conv_wgrad_bf16_kernel<KS_, ST_, NW_, INB_, true, TH_, true>;    \
            else return NIMG_ERR_ARG;                                                                         \
        }                                                                                                     \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(NW_ * 64), lds, s, p);                             \
    } while (0)
#define NIMG_WGB(KS_, ST_)                                                                                     \
    do {                                                                                                      \
        constexpr int NW = KS_ == 5 ? 8 : 4;                                                                  \
        constexpr int TH_ = (KS_ == 5 && ST_ == 1) ? 16 : B_TH;                                               \
        /* bf16-stored operands: the stride-1 layers and the 2x2 / stride-2 form (UNet Conv2DTranspose weight gradient) */ \
        constexpr bool BFOK = ST_ == 1 || KS_ == 2;                                                           \
        constexpr int S1 = BFOK ? ST_ : 1;                                                                    \
        constexpr int T1 = BFOK ? TH_ : B_TH;                                                                 \
        if (BFOK && (p.flags & NIMG_BF16_IN) && (p.flags & NIMG_BF16_DZ)) NIMG_WGB1(KS_, S1, NW, true, true, T1);    \
        else if (BFOK && (p.flags & NIMG_BF16_IN)) NIMG_WGB1(KS_, S1, NW, true, false, T1);                   \
        else if (BFOK && (p.flags & NIMG_BF16_DZ)) NIMG_WGB1(KS_, S1, NW, false, true, T1);                   \
        else if (p.flags) return NIMG_ERR_ARG;                                                                \
        else NIMG_WGB1(KS_, ST_, NW, false, false, TH_);                                                      \
    } while (0)
    if (stride == 1 && ks == 1) NIMG_WGB(1, 1);
    else if (stride == 1 && ks == 3) NIMG_WGB(3, 1);
    else if (stride == 1 && ks == 5) NIMG_WGB(5, 1);
    else if (stride == 2 && ks == 2) NIMG_WGB(2, 2);
    else if (stride == 2 && ks == 5) NIMG_WGB(5, 2);
    else return NIMG_ERR_ARG;
#undef NIMG_WGB
#undef NIMG_WGB1
    NIMG_CHECK_LAUNCH();
    if (p.tickets) return NIMG_OK;
    finish_reduce2((const float*)workspace, dw, count, p.splits, db ? (const float*)p.db_partial : nullptr, db, (long)cout,
                   p.splits, accumulate, s);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_conv2d_wgrad_bf16(const float* in1, int c1, const float* in2, int c2, const float* dz, int cout, float* dw,
                           float* db, int n, int h, int wd, int ks, int stride, int pad_t, int pad_l, int pad_mode,
                           int hout, int wout, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    return wgrad_bf16_impl(in1, c1, in2, c2, dz, nullptr, cout, dw, db, n, h, wd, ks, stride, pad_t, pad_l, pad_mode, hout,
                           wout, accumulate, workspace, workspace_bytes, 0, stream);
}

int nimg_conv2d_wgrad_bf16_ex(const float* in1, int c1, const float* in2, int c2, const float* dz, int cout, float* dw,
                              float* db, int n, int h, int wd, int ks, int stride, int pad_t, int pad_l, int pad_mode,
                              int hout, int wout, int accumulate, void* workspace, size_t workspace_bytes, int flags,
                              void* stream) {
    return wgrad_bf16_impl(in1, c1, in2, c2, dz, nullptr, cout, dw, db, n, h, wd, ks, stride, pad_t, pad_l, pad_mode, hout,
                           wout, accumulate, workspace, workspace_bytes, flags, stream);
}

/* Weight (+bias) gradient of a fused conv + pool layer with MANY input channels (the FAN's conv2..4, 5x5, stride 1, SAME) from
 * the POOLED gradient: in (n,h,wd,cin) bf16, g (n,h/2,wd/2,cout) bf16 already multiplied by LeakyReLU', idx its arg-max bytes.
 * The 2x2 un-pooling happens while the gradient tile is staged.  cin % 8 == 0, cout % 8 == 0, h, wd even. */
int nimg_conv2d_wgrad_bf16_unpool(const void* in, int cin, const void* g, const unsigned char* idx, int cout, float* dw, float* db,
                                  int n, int h, int wd, int ks, int accumulate, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    if (!idx || ks != 5 || (h & 1) || (wd & 1)) return NIMG_ERR_ARG;
    return wgrad_bf16_impl((const float*)in, cin, nullptr, 0, (const float*)g, idx, cout, dw, db, n, h, wd, ks, 1, 2, 2, 0, h, wd,
                           accumulate, workspace, workspace_bytes, NIMG_BF16_IN | NIMG_BF16_DZ, stream);
}

/* DEFERRED forms of nimg_conv2d_wgrad_bf16_ex / _unpool (idx != null): the split-K partial sums are written to `workspace`, the
 * slab reduction is NOT launched - it is described in *entry (nimg_reduce_entry_bytes() bytes of host memory) for a later
 * nimg_reduce_slabs_batch() on the same stream.  The workspace must stay untouched until then.  accumulate must be 0. */
int nimg_conv2d_wgrad_bf16_deferred(const void* in1, int c1, const void* in2, int c2, const void* dz, const unsigned char* idx,
                                    int cout, float* dw, float* db, int n, int h, int wd, int ks, int stride, int pad_t, int pad_l,
                                    int pad_mode, int hout, int wout, void* workspace, size_t workspace_bytes, int flags,
                                    void* entry, void* stream) {
    if (!entry) return NIMG_ERR_ARG;
    nimg::ReduceEntry* e = reinterpret_cast<nimg::ReduceEntry*>(entry);
    nimg::fill_reduce_entry(e, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, 0, 0);          // n1 == 0: nothing owed (paths that reduce themselves)
    e->blocks1 = 0;
    g_defer = e;
    const int rc = wgrad_bf16_impl((const float*)in1, c1, (const float*)in2, c2, (const float*)dz, idx, cout, dw, db, n, h, wd, ks,
                                   stride, pad_t, pad_l, pad_mode, hout, wout, 0, workspace, workspace_bytes, flags, stream);
    g_defer = nullptr;
    return rc;
}

/* nimg_conv2d_wgrad_bf16_deferred that also runs the reduction a PREVIOUS deferred / chained call on the same stream owes
 * (pre_entry, may be NULL): in the prologue of this call's kernel where that kernel can (the UNet's 3x3 all-taps kernel, the generic
 * bf16 kernel), as a separate launch in front of it otherwise.  Bit-identical sums (common.h reduce_seq). */
int nimg_conv2d_wgrad_bf16_chained(const void* in1, int c1, const void* in2, int c2, const void* dz, const unsigned char* idx,
                                   int cout, float* dw, float* db, int n, int h, int wd, int ks, int stride, int pad_t, int pad_l,
                                   int pad_mode, int hout, int wout, void* workspace, size_t workspace_bytes, int flags,
                                   const void* pre_entry, void* entry, void* stream) {
    if (!entry) return NIMG_ERR_ARG;
    nimg::ReduceEntry pre_copy;
    if (pre_entry) { pre_copy = *reinterpret_cast<const nimg::ReduceEntry*>(pre_entry); g_pre = &pre_copy; }   // (entry may alias pre_entry)
    nimg::ReduceEntry* e = reinterpret_cast<nimg::ReduceEntry*>(entry);
    *e = nimg::empty_reduce_entry();
    g_defer = e;
    const int rc = wgrad_bf16_impl((const float*)in1, c1, (const float*)in2, c2, (const float*)dz, idx, cout, dw, db, n, h, wd, ks,
                                   stride, pad_t, pad_l, pad_mode, hout, wout, 0, workspace, workspace_bytes, flags, stream);
    g_defer = nullptr;
    if (g_pre) {                   // an argument check returned before any path took it: do not lose the reduction
        const nimg::ReduceEntry* p = g_pre;
        g_pre = nullptr;
        if (p->n1 > 0 && p->p1 && p->d1)
            nimg::launch_reduce2(p->p1, p->d1, p->n1, p->splits1, p->p2, p->d2, p->n2, p->splits2, p->accumulate, (hipStream_t)stream);
    }
    return rc;
}

size_t nimg_reduce_entry_bytes(void) { return sizeof(nimg::ReduceEntry); }
int nimg_reduce_batch_max(void) { return nimg::REDUCE_BATCH_MAX; }

/* The reductions owed by up to nimg_reduce_batch_max() deferred weight gradients, one launch; entries = n x
 * nimg_reduce_entry_bytes() bytes of HOST memory as the deferred calls filled them (entries that owe nothing are skipped). */
int nimg_reduce_slabs_batch(const void* entries, int n, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!entries || n < 0 || n > nimg::REDUCE_BATCH_MAX) return NIMG_ERR_ARG;
    const nimg::ReduceEntry* src = reinterpret_cast<const nimg::ReduceEntry*>(entries);
    nimg::ReduceBatch b;
    b.n = 0;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (src[i].n1 <= 0 || !src[i].p1 || !src[i].d1) continue;
        b.e[b.n] = src[i];
        b.first_block[b.n] = blocks;
        blocks += src[i].blocks1 + ((src[i].p2 && src[i].d2) ? nimg::reduce_grid(src[i].n2) : 0);
        ++b.n;
    }
    b.first_block[b.n] = blocks;
    if (b.n == 0) return NIMG_OK;
    hipLaunchKernelGGL(nimg::reduce_slabs_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, b);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

/* Weight (+bias) gradient of a fused conv+pool layer (nimg_conv2d_pool_fwd_bf16) with few input channels (cin 3|4):
 * the output gradient arrives POOLED - g (n,h/2,wd/2,cout), already multiplied by LeakyReLU'(pooled) - with the
 * arg-max bytes of the forward pass; the sparse full-resolution gradient is never materialised. */
int nimg_conv2d_wgrad_pooled_bf16(const float* in, int cin, const float* g, const unsigned char* idx, int cout,
                                  float* dw, float* db, int n, int h, int wd, int ks, int accumulate, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    return nimg_conv2d_wgrad_pooled_bf16_ex(in, cin, g, idx, cout, dw, db, n, h, wd, ks, accumulate, workspace,
                                            workspace_bytes, 0, stream);
}

/* flags: NIMG_BF16_DZ = the pooled gradient g is stored as bf16 */
int nimg_conv2d_wgrad_pooled_bf16_ex(const float* in, int cin, const float* g, const unsigned char* idx, int cout,
                                     float* dw, float* db, int n, int h, int wd, int ks, int accumulate, void* workspace,
                                     size_t workspace_bytes, int flags, void* stream) {
    if (!idx || (cin != 3 && cin != 4) || (ks != 3 && ks != 5) || (h & 1) || (wd & 1) || (cout & 3)) return NIMG_ERR_ARG;
    return wgrad_bf16_impl(in, cin, nullptr, 0, g, idx, cout, dw, db, n, h, wd, ks, 1, (ks - 1) / 2, (ks - 1) / 2, 0, h, wd,
                           accumulate, workspace, workspace_bytes, flags & NIMG_BF16_DZ, stream);
}

}  // extern "C"

// ==================================================================================================================
// FAN front end in throughput mode (the 3-channel side of the first convolution, models/forensics.py:69): the three
// passes that touch the 256x256x32 tensor are HBM-bound (2.7 GB per 320-image batch each), so they must not waste the
// matrix core on channel padding nor the LDS on re-reads.
namespace {

// ---- forward, Cin <= 4: K = (tap, ci) packed (75 -> 80), A gathered from f32 channel planes, B = [co][k] bf16 ---------
template <int KS, int CINP, int TN>
__global__ __launch_bounds__(256) void conv_fwd_packed_bf16_kernel(const float* __restrict__ in,
                                                                   const float* __restrict__ w,
                                                                   const float* __restrict__ bias,
                                                                   float* __restrict__ out,
                                                                   float* __restrict__ pool_out,
                                                                   unsigned char* __restrict__ pool_idx, int N, int H,
                                                                   int W, int Cout, int pad_mode, int act, float alpha,
                                                                   int tiles_y, int tiles_x, int tiles_per_wg,
                                                                   int out_bf16) {
    constexpr int TH = 16, TW = 16, THH = TH + KS - 1, TWH = TW + KS - 1, P = (KS - 1) / 2;
    constexpr int NPIXH = THH * TWH, PS = ((NPIXH + 31) / 32) * 32 + 2;
    constexpr int KTOT = KS * KS * CINP, KSTEPS = (KTOT + 15) / 16, KP = KSTEPS * 16;
    constexpr int NI = TN / 32, MI = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* sA = reinterpret_cast<float*>(smem_raw);                           // [CINP][PS] f32
    constexpr int A_BYTES = (CINP * PS * 4 + 15) / 16 * 16;
    __bf16* sB = reinterpret_cast<__bf16*>(smem_raw + A_BYTES);               // [TN][KP] bf16, 16-byte aligned
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cot = (Cout + TN - 1) / TN;
    const int xbid = xcd_order(blockIdx.x);
    const int co0 = (xbid % cot) * TN, wg = xbid / cot;
    const int tiles = tiles_y * tiles_x;
    const long total_tiles = (long)tiles * N;
    for (int item = tid; item < TN * KP; item += 256) {
        const int k = item % KP, j = item / KP;
        sB[item] = (__bf16)((k < KTOT && co0 + j < Cout) ? w[(long)k * Cout + co0 + j] : 0.f);
    }
    int koff[KSTEPS][8];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int k = s * 16 + half * 8 + j;
            k = k < KTOT ? k : 0;                                  // padded slots meet zero weights
            const int tap = k / CINP, ci = k - tap * CINP;
            koff[s][j] = ci * PS + (tap / KS) * TWH + (tap % KS);
        }
    int abase[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int Pp = (wave * MI + mi) * 32 + (lane & 31);
        abase[mi] = (Pp / TW) * TWH + (Pp % TW);
    }
    // the halo of tile t+1 is fetched into registers while tile t is computed and stored (the tiles are tiny, so the
    // loop is otherwise a chain of exposed HBM latencies)
    constexpr int PPT = (NPIXH + 255) / 256;
    float pre[PPT][CINP];
    auto fetch = [&](long gt) {
        const int n_ = (int)(gt / tiles), tile_ = (int)(gt % tiles);
        const int ty_ = (tile_ / tiles_x) * TH, tx_ = (tile_ % tiles_x) * TW;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int pix = tid + q * 256;
            int gy = ty_ - P + pix / TWH, gx = tx_ - P + pix % TWH;
            const bool ok = pix < NPIXH && map_coord(gy, H, pad_mode) && map_coord(gx, W, pad_mode);
            const float* src = in + (((long)n_ * H + gy) * W + gx) * CINP;
#pragma unroll
            for (int c = 0; c < CINP; ++c) pre[q][c] = ok ? src[c] : 0.f;
        }
    };
    const long gt0 = (long)wg * tiles_per_wg;
    if (gt0 < total_tiles) fetch(gt0);
    for (int tt = 0; tt < tiles_per_wg; ++tt) {
        const long gt = gt0 + tt;
        if (gt >= total_tiles) break;
        const int n = (int)(gt / tiles), tile = (int)(gt % tiles);
        const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int pix = tid + q * 256;
            if (pix < NPIXH) {
#pragma unroll
                for (int c = 0; c < CINP; ++c) sA[c * PS + pix] = pre[q][c];
            }
        }
        __syncthreads();
        if (tt + 1 < tiles_per_wg && gt + 1 < total_tiles) fetch(gt + 1);
        f32x16 acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.0f;
        // an opaque per-tile copy of the pixel bases: otherwise all KSTEPS*8*MI gather addresses (loop invariant) are
        // hoisted out of the tile loop and pinned in ~80 VGPRs, which drops the kernel to one wave per SIMD
        int ab[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            ab[mi] = abase[mi];
            asm volatile("" : "+v"(ab[mi]));
        }
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            bf16x8 b[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                b[ni] = *reinterpret_cast<const bf16x8*>(sB + (ni * 32 + (lane & 31)) * KP + s * 16 + half * 8);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                float f[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = sA[koff[s][j] + ab[mi]];
                const bf16x8 a = pack8(f);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[ni], acc[mi][ni], 0, 0, 0);
            }
            // keep the gathers of later k-steps from being hoisted up here: that costs ~250 VGPRs (one wave per SIMD);
            // with the fence the kernel fits 3-4 waves per SIMD, which is what hides the LDS gather latency
            __builtin_amdgcn_sched_barrier(0);
        }
        if (pool_out) {                 // fused activation + 2x2 max-pool (common.h); private per-wave scratch
            float* elds = reinterpret_cast<float*>(smem_raw + A_BYTES + TN * KP * 2) + wave * (32 * (NI * 32 + EPI_PAD));
            const int Hp = H >> 1, Wp = W >> 1;
            const float al = act == 1 ? alpha : 1.0f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int py = (ty0 >> 1) + wave * MI + mi;
                pool_via_lds<NI, false>(acc[mi], elds, lane, al,
                    [&](int c) {
                        return (bias && co0 + c < Cout) ? *reinterpret_cast<const float4*>(bias + co0 + c)
                                                        : make_float4(0.f, 0.f, 0.f, 0.f);
                    },
                    [&](int pc, int c, float4 v, uchar4 k) {
                        const int co = co0 + c, px = (tx0 >> 1) + pc;
                        if (co >= Cout || py >= Hp || px >= Wp) return;
                        const long o = (((long)n * Hp + py) * Wp + px) * Cout + co;
                        if (out_bf16) store4_bf16(pool_out, o, v);
                        else *reinterpret_cast<float4*>(pool_out + o) = v;
                        if (pool_idx) *reinterpret_cast<uchar4*>(pool_idx + o) = k;
                    });
            }
            continue;
        }
        if ((Cout & 3) == 0) {          // vector epilogue: 16 B per lane along the channels (common.h)
            float* elds = reinterpret_cast<float*>(smem_raw + A_BYTES + TN * KP * 2) + wave * (32 * (NI * 32 + EPI_PAD));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                epilogue_via_lds<NI, false>(acc[mi], elds, lane, [&](int row, int c, float4 v) {
                    const int co = co0 + c;
                    if (co >= Cout) return;
                    const int Pp = (wave * MI + mi) * 32 + row;
                    const int oy = ty0 + Pp / TW, ox = tx0 + Pp % TW;
                    if (oy >= H || ox >= W) return;
                    if (bias) {
                        const float4 b4 = *reinterpret_cast<const float4*>(bias + co);
                        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
                    }
                    if (act == 1) {
                        v.x = lrelu(v.x, alpha); v.y = lrelu(v.y, alpha); v.z = lrelu(v.z, alpha); v.w = lrelu(v.w, alpha);
                    }
                    const long o = (((long)n * H + oy) * W + ox) * Cout + co;
                    if (out_bf16) store4_bf16(out, o, v);
                    else *reinterpret_cast<float4*>(out + o) = v;
                });
            }
            continue;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int co = co0 + ni * 32 + (lane & 31);
            if (co >= Cout) continue;
            const float bv = bias ? bias[co] : 0.f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int Pp = (wave * MI + mi) * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
                    const int oy = ty0 + Pp / TW, ox = tx0 + Pp % TW;
                    if (oy >= H || ox >= W) continue;
                    float v = acc[mi][ni][j] + bv;
                    if (act == 1) v = lrelu(v, alpha);
                    out[(((long)n * H + oy) * W + ox) * Cout + co] = v;
                }
        }
    }
}

// ---- weight gradient, Cin <= 4: M = (tap, ci) packed, K = 16 pixels per MFMA, operands gathered from f32 tiles --------
// ZMODE 0: dz at full resolution (float32); 1: POOLED gradient (float32) + arg-max bytes, un-pooled while staging;
// 2: the same with the pooled gradient stored as bf16.  Compile-time, so the prefetch loads sit in straight-line code.
template <int KS, int CINP, int NI, int ZMODE>
__global__ __launch_bounds__(256) void conv_wgrad_packed_bf16_kernel(const WgradParamsB p) {
    constexpr int TAPS = KS * KS, TPF = 32 / CINP, MF = (TAPS + TPF - 1) / TPF;
    constexpr int THH = B_TH + KS - 1, TWH = B_TW + KS - 1, NPIXH = THH * TWH, NPIX = B_TH * B_TW, COT = 32 * NI;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sI = smem;                    // [NPIXH][CINP]
    float* sZ = smem + NPIXH * CINP;     // [NPIX][COT]
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cob = (p.Cout + COT - 1) / COT;
    const int xbid = xcd_order(blockIdx.x);
    const int co0 = (xbid % cob) * COT, split = xbid / cob;
    int aoff[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int i = lane & 31, tl = i / CINP, ci = i % CINP, tap = f * TPF + tl;
        aoff[f] = (tl < TPF && tap < TAPS) ? ((tap / KS) * TWH + (tap % KS)) * CINP + ci : -1;
    }
    f32x16 acc[MF][NI];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[f][ni][j] = 0.0f;
    const int tiles = p.tiles_y * p.tiles_x;
    const int work_total = p.N * tiles;                 // < 2^31 (checked by the entry point)
    const int w_begin = split * p.work_per_split, w_end = min(work_total, w_begin + p.work_per_split);
    const bool do_bias = p.db_partial != nullptr;
    float bsum = 0.f;
    // register prefetch of the next tile (both operands) while the current one is multiplied
    constexpr int IPT = (NPIXH + 255) / 256, ZPT = NPIX * (COT / 4) / 256;
    float prei[IPT][CINP];
    float4 prez[ZPT];
    unsigned int prek[ZPT];
    const bool vec_z = (p.Cout % 4 == 0);
    auto fetch = [&](int wk_) {
        const int n_ = (int)(wk_ / tiles), tile_ = (int)(wk_ % tiles);
        const int ty_ = (tile_ / p.tiles_x) * B_TH, tx_ = (tile_ % p.tiles_x) * B_TW;
#pragma unroll
        for (int q = 0; q < IPT; ++q) {
            const int pix = tid + q * 256;
            int gy = ty_ - p.pad_t + pix / TWH, gx = tx_ - p.pad_l + pix % TWH;
            const bool ok = pix < NPIXH && map_coord(gy, p.H, p.pad_mode) && map_coord(gx, p.W, p.pad_mode);
            const float* src = p.in1 + (((long)n_ * p.H + gy) * p.W + gx) * CINP;
#pragma unroll
            for (int c = 0; c < CINP; ++c) prei[q][c] = ok ? src[c] : 0.f;
        }
        if (vec_z) {
#pragma unroll
            for (int q = 0; q < ZPT; ++q) {
                const int item = tid + q * 256;
                const int pix = item / (COT / 4), c = co0 + (item % (COT / 4)) * 4;
                const int oy = ty_ + pix / B_TW, ox = tx_ + pix % B_TW;
                prez[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (ZMODE >= 1) {  // pooled gradient + arg-max: this pixel receives it iff it was the window maximum
                    prek[q] = 0xffffffffu;
                    if (oy < p.Hout && ox < p.Wout && c < p.Cout) {
                        const long po = (((long)n_ * (p.Hout >> 1) + (oy >> 1)) * (p.Wout >> 1) + (ox >> 1)) * p.Cout + c;
                        if constexpr (ZMODE == 2) {
                            const uint2 raw = *reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(p.dz) + po);
                            prez[q].x = __uint_as_float(raw.x);          // 4 x bf16, expanded when the tile is committed
                            prez[q].y = __uint_as_float(raw.y);
                        } else {
                            prez[q] = *reinterpret_cast<const float4*>(p.dz + po);
                        }
                        prek[q] = *reinterpret_cast<const unsigned int*>(p.dz_idx + po);
                    }
                } else if (oy < p.Hout && ox < p.Wout && c < p.Cout) {
                    prez[q] = *reinterpret_cast<const float4*>(p.dz + (((long)n_ * p.Hout + oy) * p.Wout + ox) * p.Cout + c);
                }
            }
        }
    };
    if (w_begin < w_end) fetch(w_begin);
    for (int wk = w_begin; wk < w_end; ++wk) {
        const int n = (int)(wk / tiles), tile = (int)(wk % tiles);
        const int ty0 = (tile / p.tiles_x) * B_TH, tx0 = (tile % p.tiles_x) * B_TW;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < IPT; ++q) {
            const int pix = tid + q * 256;
            if (pix < NPIXH) {
#pragma unroll
                for (int c = 0; c < CINP; ++c) sI[pix * CINP + c] = prei[q][c];
            }
        }
        if (vec_z) {
#pragma unroll
            for (int q = 0; q < ZPT; ++q) {
                const int item = tid + q * 256;
                float4 v = prez[q];
                if constexpr (ZMODE == 2) {
                    const unsigned lo = __float_as_uint(prez[q].x), hi = __float_as_uint(prez[q].y);
                    v = make_float4(__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u),
                                    __uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u));
                }
                if constexpr (ZMODE >= 1) {
                    const int pix = item / (COT / 4);
                    const unsigned pos = (unsigned)((((ty0 + pix / B_TW) & 1) << 1) | ((tx0 + pix % B_TW) & 1));
                    const unsigned k = prek[q];
                    v.x = (k & 0xffu) == pos ? v.x : 0.f;
                    v.y = ((k >> 8) & 0xffu) == pos ? v.y : 0.f;
                    v.z = ((k >> 16) & 0xffu) == pos ? v.z : 0.f;
                    v.w = (k >> 24) == pos ? v.w : 0.f;
                }
                *reinterpret_cast<float4*>(sZ + (item / (COT / 4)) * COT + (item % (COT / 4)) * 4) = v;
            }
        } else {
            for (int item = tid; item < NPIX * COT; item += 256) {
                const int pix = item / COT, c = co0 + item % COT;
                const int oy = ty0 + pix / B_TW, ox = tx0 + pix % B_TW;
                sZ[item] = (oy < p.Hout && ox < p.Wout && c < p.Cout)
                               ? p.dz[(((long)n * p.Hout + oy) * p.Wout + ox) * p.Cout + c] : 0.f;
            }
        }
        __syncthreads();
        if (wk + 1 < w_end) fetch(wk + 1);
        if (do_bias && tid < COT) {
#pragma unroll 8
            for (int px = 0; px < NPIX; ++px) bsum += sZ[px * COT + tid];
        }
        for (int r = wave; r < B_TH; r += 4) {
            float f8[8];
            bf16x8 b[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                for (int k = 0; k < 8; ++k) f8[k] = sZ[(r * B_TW + half * 8 + k) * COT + ni * 32 + (lane & 31)];
                b[ni] = pack8(f8);
            }
#pragma unroll
            for (int f = 0; f < MF; ++f) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    f8[k] = aoff[f] >= 0 ? sI[(r * TWH + half * 8 + k) * CINP + aoff[f]] : 0.f;
                const bf16x8 a = pack8(f8);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[f][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[ni], acc[f][ni], 0, 0, 0);
            }
        }
    }
    if (do_bias && tid < COT && co0 + tid < p.Cout) p.db_partial[(long)split * p.Cout + co0 + tid] = bsum;
    // the four waves hold row partials of the same (tap, ci) x co block: fold them through LDS (waves 1..3 park theirs, wave 0
    // adds in order) - one slab per workgroup instead of four (4096 slabs of the UNet's first layer took a 42 us reduction)
    constexpr bool FOLD = MF * NI <= 2;              // 3 x MF x NI x 4 KB of scratch: the small (3x3) layers only
    if constexpr (FOLD) {
        __syncthreads();
        float* red = smem;
        if (wave > 0) {
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int j = 0; j < 16; ++j) red[((((wave - 1) * MF + f) * NI + ni) * 16 + j) * 64 + lane] = acc[f][ni][j];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 1; w < 4; ++w)
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[f][ni][j] += red[((((w - 1) * MF + f) * NI + ni) * 16 + j) * 64 + lane];
    }
    float* slab = p.partial + ((long)split * (FOLD ? 1 : 4) + (FOLD ? 0 : wave)) * TAPS * CINP * p.Cout;
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int co = co0 + ni * 32 + (lane & 31);
            if (co >= p.Cout) continue;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int i = (j & 3) + 8 * (j >> 2) + 4 * half;
                const int tl = i / CINP, ci = i % CINP, tap = f * TPF + tl;
                if (tl < TPF && tap < TAPS) slab[((long)tap * CINP + ci) * p.Cout + co] = acc[f][ni][j];
            }
        }
}

// ---- input gradient towards FEW channels (CI <= 6 with KS*CI <= 32), from CZ = 32 gradient channels -----------------
//   out[u][v][ci] = sum_{ky,kx,co} dz[u+P-ky][v+P-kx][co] * w[ky][kx][ci][co]
// The kx loop is folded into the MFMA N dimension: T[u][x'][(kx,ci)] = sum_{ky,co} dz[u+P-ky][x'][co] * w[ky][kx][ci][co]
// is one 32 x 32 x (KS*CZ) GEMM per output row (32 positions x', KS*CI <= 32 columns, no padded channels), and
// out[u][v][ci] = sum_kx T[u][v+P-kx][(kx,ci)] is a shift-add through a 2 KB LDS tile.  w is the forward kernel
// (kh,kw,CI,CZ) as stored - its [ky][(kx,ci)][co] order is exactly the B operand.
// ZMODE 0: dz at full resolution (float32); 1: POOLED gradient (float32) + arg-max bytes; 2: pooled gradient as bf16.
template <int KS, int CI, int ZMODE>
__global__ __launch_bounds__(256) void conv_dgrad_fewin_bf16_kernel(const float* __restrict__ dz,
                                                                    const unsigned char* __restrict__ dz_idx,
                                                                    const float* __restrict__ w,
                                                                    float* __restrict__ out, int N, int H, int W,
                                                                    int tiles_y, int tiles_x) {
    constexpr int CZ = 32, P = (KS - 1) / 2, TH = 8, TWO = 32 - (KS - 1);      // TWO output columns per tile
    constexpr int ROWS = TH + KS - 1, NJ = KS * CI;
    static_assert(NJ <= 32, "KS * CI must fit one MFMA N tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4* sD = reinterpret_cast<uint4*>(smem_raw);                           // [ROWS*32 px][4 chunks of 8 co]
    uint4* sW = sD + ROWS * 32 * 4;                                           // [KS*32 rows][4]
    float* sT = reinterpret_cast<float*>(sW + KS * 32 * 4);                   // [4 waves][32 x'][16]
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = tiles_y * tiles_x;
    const int total_tiles = N * tiles;
    // weights: row (ky, j) = w[(ky*NJ + j)*CZ + co], j < NJ; zero rows above.  Staged ONCE: the workgroup is persistent
    // over tiles (one workgroup per tile re-staged these 10 KB - 160 scalar loads + converts per thread - 102 400 times)
    for (int item = tid; item < KS * 32 * 4; item += 256) {
        const int q = item & 3, row = item >> 2, j = row & 31, ky = row >> 5;
        float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (j < NJ) {
            const float* src = w + ((long)(ky * NJ + j)) * CZ + q * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = src[e];
        }
        const bf16x8 b = pack8(f);
        sW[row * 4 + (q ^ ((row >> 2) & 3))] = *reinterpret_cast<const uint4*>(&b);
    }
    // the next tile travels HBM/L2 -> registers while the current one is multiplied
    constexpr int NPC = (ROWS * 32 * 4 + 255) / 256;
    uint4 pd0[NPC], pd1[NPC];
    uint2 pk[NPC];
    auto fetch = [&](int gt_) {
        const int n_ = gt_ / tiles, tile_ = gt_ % tiles;
        const int u_ = (tile_ / tiles_x) * TH, v_ = (tile_ % tiles_x) * TWO;
#pragma unroll
        for (int qq = 0; qq < NPC; ++qq) {
            const int item = tid + qq * 256;
            const int q = item & 3, pix = item >> 2, xx = pix & 31, rr = pix >> 5;
            const int gy = u_ + rr - (KS - 1 - P), gx = v_ - (KS - 1 - P) + xx;
            pd0[qq] = pd1[qq] = make_uint4(0u, 0u, 0u, 0u);
            pk[qq] = make_uint2(0xffffffffu, 0xffffffffu);
            if (item < ROWS * 32 * 4 && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                const long off = ZMODE >= 1 ? (((long)n_ * (H >> 1) + (gy >> 1)) * (W >> 1) + (gx >> 1)) * CZ + q * 8
                                            : (((long)n_ * H + gy) * W + gx) * CZ + q * 8;
                if constexpr (ZMODE == 2) {
                    pd0[qq] = *reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(dz) + off);
                } else {
                    pd0[qq] = *reinterpret_cast<const uint4*>(dz + off);
                    pd1[qq] = *reinterpret_cast<const uint4*>(dz + off + 4);
                }
                if constexpr (ZMODE >= 1) pk[qq] = *reinterpret_cast<const uint2*>(dz_idx + off);
            }
        }
    };
    const int first_tile = xcd_order(blockIdx.x);      // every round hands each XCD one contiguous range of tiles
    if (first_tile < total_tiles) fetch(first_tile);
    for (int gt = first_tile; gt < total_tiles; gt += gridDim.x) {
    const int n = gt / tiles, tile = gt % tiles;
    const int u0 = (tile / tiles_x) * TH, v0 = (tile % tiles_x) * TWO;
    __syncthreads();                                 // previous tile fully consumed (and the weights staged)
    // dz tile: rows u0-P .. (row index rr <-> image row u0 + rr - (KS-1-P)), 32 column positions from v0-(KS-1-P)
#pragma unroll
    for (int qq = 0; qq < NPC; ++qq) {
        const int item = tid + qq * 256;
        if (item < ROWS * 32 * 4) {
            const int q = item & 3, pix = item >> 2, xx = pix & 31, rr = pix >> 5;
            float f[8];
            if constexpr (ZMODE == 2) {
                const bf16x8 gb = *reinterpret_cast<const bf16x8*>(&pd0[qq]);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (float)gb[e];
            } else {
                f[0] = __uint_as_float(pd0[qq].x); f[1] = __uint_as_float(pd0[qq].y);
                f[2] = __uint_as_float(pd0[qq].z); f[3] = __uint_as_float(pd0[qq].w);
                f[4] = __uint_as_float(pd1[qq].x); f[5] = __uint_as_float(pd1[qq].y);
                f[6] = __uint_as_float(pd1[qq].z); f[7] = __uint_as_float(pd1[qq].w);
            }
            if constexpr (ZMODE >= 1) {              // un-pool: this pixel receives the gradient iff it was the window maximum
                const int gy = u0 + rr - (KS - 1 - P), gx = v0 - (KS - 1 - P) + xx;
                const unsigned pos = (unsigned)(((gy & 1) << 1) | (gx & 1));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f[e] = ((pk[qq].x >> (8 * e)) & 0xffu) == pos ? f[e] : 0.f;
                    f[4 + e] = ((pk[qq].y >> (8 * e)) & 0xffu) == pos ? f[4 + e] : 0.f;
                }
            }
            const bf16x8 b = pack8(f);
            sD[pix * 4 + (q ^ ((pix >> 2) & 3))] = *reinterpret_cast<const uint4*>(&b);
        }
    }
    __syncthreads();
    if (gt + (int)gridDim.x < total_tiles) fetch(gt + gridDim.x);
    float* myT = sT + wave * 32 * 16;
    for (int ur = wave; ur < TH; ur += 4) {
        f32x16 acc;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
        // dz row needed for (output row u0+ur, tap ky): image row u0+ur+P-ky  ->  tile row rr = ur + (KS-1) - ky
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int pix = (ur + (KS - 1) - ky) * 32 + (lane & 31);
            const int row = ky * 32 + (lane & 31);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int c = h2 * 2 + half;
                const uint4 av = sD[pix * 4 + (c ^ ((pix >> 2) & 3))];
                const uint4 bv = sW[row * 4 + (c ^ ((row >> 2) & 3))];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&av),
                                                              *reinterpret_cast<const bf16x8*>(&bv), acc, 0, 0, 0);
            }
        }
        // T[x'][j] -> LDS (only the NJ real columns), then the kx shift-add
        if ((lane & 31) < 16) {
#pragma unroll
            for (int j = 0; j < 16; ++j) myT[((j & 3) + 8 * (j >> 2) + 4 * half) * 16 + (lane & 31)] = acc[j];
        }
        __builtin_amdgcn_wave_barrier();             // myT is private to this wave; its LDS operations complete in order
        const int u = u0 + ur;
        for (int o = lane; o < TWO * CI; o += 64) {
            const int vi = o / CI, ci = o % CI, v = v0 + vi;
            float s = 0.f;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) s += myT[(vi + (KS - 1) - kx) * 16 + kx * CI + ci];
            if (u < H && v < W) out[(((long)n * H + u) * W + v) * CI + ci] = s;
        }
        __builtin_amdgcn_wave_barrier();
    }
    }
}

}  // namespace

extern "C" {

/* FAN front end in throughput mode.  Few INPUT channels (cin 3|4, float32 HWIO weights, converted in-kernel). */
static int launch_packed_bf16(const float* in, int cin, const float* w, const float* bias, float* out, float* pool_out,
                              unsigned char* pool_idx, int cout, int n, int h, int wd, int ks, int pad_mode, int act,
                              float alpha, hipStream_t s, int out_bf16 = 0) {
    const int ty = cdiv(h, 16), tx = cdiv(wd, 16);
    const long total_tiles = (long)ty * tx * n;
    const int tpw = total_tiles >= 8192 ? 8 : (total_tiles >= 2048 ? 2 : 1);
#define NIMG_FP(KS_, C_, TN_)                                                                                     \
    do {                                                                                                          \
        constexpr int THH = 16 + KS_ - 1, NPIXH = THH * THH, PS = ((NPIXH + 31) / 32) * 32 + 2;                    \
        constexpr int KP = (KS_ * KS_ * C_ + 15) / 16 * 16;                                                       \
        constexpr size_t lds = (size_t)((C_ * PS * 4 + 15) / 16 * 16) + (size_t)TN_ * KP * 2 +                    \
                               (size_t)4 * 32 * (TN_ + EPI_PAD) * sizeof(float);                                  \
        const long blocks = cdiv(total_tiles, tpw) * (long)cdiv(cout, TN_);                                       \
        hipLaunchKernelGGL((conv_fwd_packed_bf16_kernel<KS_, C_, TN_>), dim3((unsigned)blocks), dim3(256), lds, s, \
                           in, w, bias, out, pool_out, pool_idx, n, h, wd, cout, pad_mode, act, alpha, ty, tx, tpw,    \
                           out_bf16);                                                                             \
    } while (0)
    if (ks == 5 && cin == 3) { if (cout > 32) NIMG_FP(5, 3, 64); else NIMG_FP(5, 3, 32); }
    else if (ks == 5 && cin == 4) { if (cout > 32) NIMG_FP(5, 4, 64); else NIMG_FP(5, 4, 32); }
    else if (ks == 3 && cin == 3) { if (cout > 32) NIMG_FP(3, 3, 64); else NIMG_FP(3, 3, 32); }
    else if (ks == 3 && cin == 4) { if (cout > 32) NIMG_FP(3, 4, 64); else NIMG_FP(3, 4, 32); }
    else return NIMG_ERR_ARG;
#undef NIMG_FP
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_conv2d_fwd_smallc_bf16(const float* in, int cin, const float* w, const float* bias, float* out, int cout,
                                int n, int h, int wd, int ks, int pad_mode, int act, float alpha, void* stream) {
    return nimg_conv2d_fwd_smallc_bf16_ex(in, cin, w, bias, out, cout, n, h, wd, ks, pad_mode, act, alpha, 0, stream);
}

/* flags: NIMG_BF16_OUT = out is stored as bf16 (cout % 4 == 0) */
int nimg_conv2d_fwd_smallc_bf16_ex(const float* in, int cin, const float* w, const float* bias, float* out, int cout,
                                   int n, int h, int wd, int ks, int pad_mode, int act, float alpha, int flags, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!in || !w || !out || n < 0 || h <= 0 || wd <= 0 || cout <= 0 || pad_mode < 0 || pad_mode > 2) return NIMG_ERR_ARG;
    if ((flags & ~NIMG_BF16_OUT) || ((flags & NIMG_BF16_OUT) && (cout & 3))) return NIMG_ERR_ARG;
    return launch_packed_bf16(in, cin, w, bias, out, nullptr, nullptr, cout, n, h, wd, ks, pad_mode, act, alpha,
                              (hipStream_t)stream, (flags & NIMG_BF16_OUT) ? 1 : 0);
}

/* conv (SAME, stride 1) + optional LeakyReLU + 2x2/2 max-pool in one pass, bf16 operands: w = f32 kernel (used when
 * cin <= 4), wb = nimg_conv_weights_bf16(mode 0) image of it (used otherwise) */
int nimg_conv2d_pool_fwd_bf16(const float* in, int cin, const float* w, const void* wb, const float* bias,
                              float* pool_out, unsigned char* pool_idx, int cout, int n, int h, int wd, int ks, int act,
                              float alpha, void* stream) {
    return nimg_conv2d_pool_fwd_bf16_ex(in, cin, w, wb, bias, pool_out, pool_idx, cout, n, h, wd, ks, act, alpha, 0, stream);
}

int nimg_conv2d_pool_fwd_bf16_ex(const float* in, int cin, const float* w, const void* wb, const float* bias,
                                 float* pool_out, unsigned char* pool_idx, int cout, int n, int h, int wd, int ks,
                                 int act, float alpha, int flags, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!in || !pool_out || cin <= 0 || cout <= 0 || (cout & 3) || n < 0 || h <= 0 || wd <= 0) return NIMG_ERR_ARG;
    if ((h & 1) || (wd & 1) || (ks != 3 && ks != 5) || act < 0 || act > 1) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    hipStream_t s = (hipStream_t)stream;
    if (cin == 3 || cin == 4) {
        if (!w) return NIMG_ERR_ARG;
        if (flags & NIMG_BF16_IN) return NIMG_ERR_ARG;
        return launch_packed_bf16(in, cin, w, bias, nullptr, pool_out, pool_idx, cout, n, h, wd, ks, 0, act, alpha, s,
                                  (flags & NIMG_BF16_OUT) ? 1 : 0);
    }
    if (!wb || (cin % 8)) return NIMG_ERR_ARG;
    ConvParamsB p;
    p.in1 = in; p.in2 = nullptr; p.wb = (const __bf16*)wb; p.bias = bias; p.out1 = nullptr; p.out2 = nullptr;
    p.act1 = nullptr; p.pool_out = pool_out; p.pool_idx = pool_idx; p.convt = 0; p.flags = flags; p.in_idx = nullptr; p.res = nullptr;
    p.out1b = nullptr;
    p.C1 = cin; p.C2 = 0; p.O1 = cout; p.O2 = 0; p.CinP = (cin + 15) / 16 * 16;
    p.N = n; p.H = h; p.W = wd; p.Hout = h; p.Wout = wd; p.pad_t = p.pad_l = (ks - 1) / 2;
    p.tiles_y = p.tiles_x = 0; p.act = act; p.pad_mode = 0; p.alpha = alpha;
    const bool tn32 = cout <= 32 || (long)cdiv(cout, 64) * cdiv(h, 16) * cdiv(wd, 16) * n < 384;
    if (flags & NIMG_BF16_IN) {
        if (ks == 3) return tn32 ? launch_conv_b<3, 1, 16, 16, 1, 32, true>(p, s) : launch_conv_b<3, 1, 16, 16, 1, 64, true>(p, s);
        return tn32 ? launch_conv_b<5, 1, 16, 16, 1, 32, true>(p, s) : launch_conv_b<5, 1, 16, 16, 1, 64, true>(p, s);
    }
    if (ks == 3) return tn32 ? launch_conv_b<3, 1, 16, 16, 1, 32>(p, s) : launch_conv_b<3, 1, 16, 16, 1, 64>(p, s);
    return tn32 ? launch_conv_b<5, 1, 16, 16, 1, 32>(p, s) : launch_conv_b<5, 1, 16, 16, 1, 64>(p, s);
}

/* input gradient of a (ks,ks,ci,32) SAME stride-1 convolution towards its ci (= 3) input channels; w = the FORWARD
 * kernel as stored (not flipped) */
static int dgrad_fewin_impl(const float* dz, const unsigned char* dz_idx, int dz_bf16, const float* w, float* out,
                            int ci, int cz, int n, int h, int wd, int ks, void* stream) {
    if (!dz || !w || !out || n < 0 || h <= 0 || wd <= 0) return NIMG_ERR_ARG;
    if (cz != 32 || ci != 3 || ks != 5) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    constexpr int KS = 5, TH = 8, TWO = 32 - (KS - 1), ROWS = TH + KS - 1;
    const int ty = cdiv(h, TH), tx = cdiv(wd, TWO);
    constexpr size_t lds = (size_t)(ROWS * 32 * 4 + KS * 32 * 4) * sizeof(uint4) + 4 * 32 * 16 * sizeof(float);
    const long total = (long)n * ty * tx;
    if (total >= (1L << 31)) return NIMG_ERR_ARG;
    const unsigned grid = (unsigned)(total < 4096 ? total : 4096);       // persistent: ~16 workgroups per CU
    hipStream_t s = (hipStream_t)stream;
    if (!dz_idx)
        hipLaunchKernelGGL((conv_dgrad_fewin_bf16_kernel<5, 3, 0>), dim3(grid), dim3(256), lds, s, dz, dz_idx, w, out, n, h, wd,
                           ty, tx);
    else if (dz_bf16)
        hipLaunchKernelGGL((conv_dgrad_fewin_bf16_kernel<5, 3, 2>), dim3(grid), dim3(256), lds, s, dz, dz_idx, w, out, n, h, wd,
                           ty, tx);
    else
        hipLaunchKernelGGL((conv_dgrad_fewin_bf16_kernel<5, 3, 1>), dim3(grid), dim3(256), lds, s, dz, dz_idx, w, out, n, h, wd,
                           ty, tx);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_conv2d_dgrad_fewin_bf16(const float* dz, const float* w, float* out, int ci, int cz, int n, int h, int wd,
                                 int ks, void* stream) {
    return dgrad_fewin_impl(dz, nullptr, 0, w, out, ci, cz, n, h, wd, ks, stream);
}

/* the same input gradient with the output gradient given POOLED (g (n,h/2,wd/2,cz) + arg-max bytes), see
 * nimg_conv2d_wgrad_pooled_bf16 */
int nimg_conv2d_dgrad_fewin_pooled_bf16(const float* g, const unsigned char* idx, const float* w, float* out, int ci,
                                        int cz, int n, int h, int wd, int ks, void* stream) {
    return nimg_conv2d_dgrad_fewin_pooled_bf16_ex(g, idx, w, out, ci, cz, n, h, wd, ks, 0, stream);
}

int nimg_conv2d_dgrad_fewin_pooled_bf16_ex(const float* g, const unsigned char* idx, const float* w, float* out, int ci,
                                           int cz, int n, int h, int wd, int ks, int flags, void* stream) {
    if (!idx || (h & 1) || (wd & 1)) return NIMG_ERR_ARG;
    return dgrad_fewin_impl(g, idx, (flags & NIMG_BF16_DZ) ? 1 : 0, w, out, ci, cz, n, h, wd, ks, stream);
}

}  // extern "C"

#ifdef NIMG_CONV3_TIMING
extern "C" int nimg_debug_conv3_timing(unsigned long long* host, int n_words) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv3_timing), (size_t)n_words * 8) == hipSuccess ? 0 : -2;
}
#endif
