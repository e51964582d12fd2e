// Input gradient of the FAN's fused 5x5 conv + LeakyReLU + MaxPool layers (models/forensics.py:69-77 under tape.gradient) on the
// STRUCTURED-SPARSITY matrix instruction, throughput mode.  The gradient arriving at such a layer is the 2x2 un-pooling of the
// pooled gradient g (N, H/2, W/2, Cout): of the four pixels of a pooling window exactly one - the arg-max - carries a channel's
// value.  conv5_ring_kernel<TN, UNP> (conv_bf16.hip) builds that un-pooled tile in LDS and multiplies the zeros; here
//     din[2a + ey][2b + ex][ci] = sum over the 3 x 3 pooled neighbours (wy, wx), co, window positions (py, px) of
//         [g[a + wy - 1][b + wx - 1][co] if arg-max == 2 py + px]  x  w[ky][kx][ci][co],
//         ky = ey + 4 - 2 wy - py,  kx = ex + 4 - 2 wx - px   (0 outside 0..4),
// is ONE GEMM over the POOLED pixels: M = (a, b), N = (output parity class (ey, ex), ci), K = (window, co, position) - and along K
// every group of four (the positions of one window and channel) holds exactly one value: v_smfmac_f32_32x32x32_bf16 takes the
// pooled tensor as its compressed A operand (kept element 0 = g, element 1 = 0) with the arg-max byte as the index and runs at
// twice the dense rate over a K that is 36 / 25 of the dense one (the 3 x 3 x 4 window positions include 11 that no 5x5 tap
// reaches: zero weights) - 0.72 of the matrix cycles, and the A tile is the pooled 18 x 18 halo instead of the 20 x 20
// full-resolution one.  (Operand layout and index semantics: tools/probe/smfmac_probe.hip.)
//
// Workgroup = 16 x 16 pooled pixels x 128 columns = 4 parity classes x 32 input channels (N fragment ni IS class ni), 4 waves x
// (2 x 4) fragments of 32 x 32.  K runs over 8-channel chunks of co (one K = 32 instruction: 8 channels x 4 positions) and the
// 3 x 3 windows; the weights arrive one window ROW at a time (3 windows x 128 columns x 32 K = 24 KB, pre-arranged by
// dgrad5s_weights_kernel in exactly the LDS order, conflict-free planes of 16-byte pieces) by LDS-DMA into a two-slot ring,
// the compressed A tile (+ its 16-bit index words) of the next chunk is committed under the last window row - as in the ring
// kernels.  The epilogue turns the accumulators through LDS and scatters class ni to pixel (2a + (ni >> 1), 2b + (ni & 1)),
// x LeakyReLU' of the layer below (act mask), bf16 or float32.
#include <stdlib.h>

#include "common.h"

namespace {

using namespace nimg;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct Dg5sParams {
    const void* g;                 // (N, Hp, Wp, Cz) bf16 pooled gradient (already x LeakyReLU' of this layer)
    const unsigned char* idx;      // arg-max bytes, same shape
    const void* img;               // weight image, see dgrad5s_weights_kernel
    void* out;                     // (N, 2 Hp, 2 Wp, Ci) float32 or bf16
    const void* act;               // optional (N, 2 Hp, 2 Wp, Ci): out *= act > 0 ? 1 : alpha
    int Cz, Ci, N, Hp, Wp, tiles_y, tiles_x, flags;
    float alpha;
};

// img[nt][chunk][wy][wx][plane = 2 hb + part][n = 32 class + ci][8]: the B fragment of lane (n, hb) is its 16-byte pieces of
// planes (hb, 0) and (hb, 1), K index k = 16 part + 8 hb + j = 4 (co - 8 chunk) + position (the layout the probe measured)
__global__ void dgrad5s_weights_kernel(const float* __restrict__ w, __bf16* __restrict__ img, int ci_n, int cz) {
    const long total = (long)(ci_n / 32) * (cz / 8) * 9 * 4 * 128 * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), n = (int)((i >> 3) & 127), plane = (int)((i >> 10) & 3);
        long r = i >> 12;
        const int wx = (int)(r % 3);
        r /= 3;
        const int wy = (int)(r % 3);
        r /= 3;
        const int chunk = (int)(r % (cz / 8)), nt = (int)(r / (cz / 8));
        const int k = 16 * (plane & 1) + 8 * (plane >> 1) + j, co = chunk * 8 + (k >> 2), py = (k >> 1) & 1, px = k & 1;
        const int cls = n >> 5, ci = nt * 32 + (n & 31);
        const int ky = (cls >> 1) + 4 - 2 * wy - py, kx = (cls & 1) + 4 - 2 * wx - px;
        float v = 0.f;
        if (ky >= 0 && ky <= 4 && kx >= 0 && kx <= 4) v = w[((long)(ky * 5 + kx) * ci_n + ci) * cz + co];
        img[i] = (__bf16)v;
    }
}

// one LDS-DMA piece: 64 lanes x 16 B from buffer `rsrc` (per-lane byte offset voff + scalar soff) to LDS bytes [lds_addr, + 1024)
typedef unsigned int r_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(r_u32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

constexpr int S_TP = 16, S_HP = S_TP + 2, S_NPIX = S_HP * S_HP;          // pooled tile, its halo
constexpr int S_SLAB = 3 * 4 * 128 * 16;                                  // one window row of weights: 24 KB
constexpr int S_APLANE = S_NPIX * 16, S_ABUF = 2 * S_APLANE + 2 * S_NPIX * 4;     // compressed A planes + index words
constexpr int S_LDS_TILES = 2 * S_SLAB + 2 * S_ABUF;
constexpr int S_LDS_EPI = 4 * 32 * (128 + EPI_PAD) * 4;
constexpr int S_LDS = S_LDS_TILES > S_LDS_EPI ? S_LDS_TILES : S_LDS_EPI;

// MI x NI = the fragment block of a wave (8 accumulators): 2 x 4 = four waves stacked along the pixels, each across all four parity
// classes; 4 x 2 = two pixel halves x two class pairs - the dense operand (2 KB per fragment) is read half as often per matrix
// instruction (8 KB of LDS reads per 8 instructions instead of 10) and is 2 % SLOWER (profiles/r06_dgrad5s_block.txt): opt-in
template <int MI>
__global__ __launch_bounds__(256, 2) void conv5_dgrad_sparse_kernel(const Dg5sParams p) {
    constexpr int NI = 8 / MI, WAVES_M = 8 / MI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* sB = smem_raw;                                   // ring first: the LDS-DMA base (M0) stays below 64 KB
    unsigned char* sA = smem_raw + 2 * S_SLAB;
    const unsigned sB_addr = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int h = lane >> 5, nl = lane & 31;
    const int nts = p.Ci / 32, chunks = p.Cz / 8;
    int bid = xcd_order(blockIdx.x);
    const int nt = bid % nts;
    bid /= nts;
    const int tiles = p.tiles_y * p.tiles_x;
    const int tile = bid % tiles, n = bid / tiles;
    const int ty0 = (tile / p.tiles_x) * S_TP, tx0 = (tile % p.tiles_x) * S_TP;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.0f;

    // ---- A tile staging: items = halo pixels (8 channels of the chunk: 16 B of gradient + 8 arg-max bytes), 2 per thread
    constexpr int AP = (S_NPIX + 255) / 256;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.g), 0,
                                                                        (int)((long)p.N * p.Hp * p.Wp * p.Cz * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.idx), 0,
                                                                        (int)((long)p.N * p.Hp * p.Wp * p.Cz), 0x00020000);
    unsigned aoff[AP];
#pragma unroll
    for (int q = 0; q < AP; ++q) {
        const int pix = tid + q * 256;
        const int gy = ty0 - 1 + pix / S_HP, gx = tx0 - 1 + pix % S_HP;
        const bool ok = (pix < S_NPIX) & ((unsigned)gy < (unsigned)p.Hp) & ((unsigned)gx < (unsigned)p.Wp);
        aoff[q] = ok ? (unsigned)(((n * p.Hp + gy) * p.Wp + gx) * p.Cz) : 0x40000000u;      // element offset; x 2 below stays out of range
    }
    u32x4 preG[AP];
    u32x2 preK[AP];
    auto fetchA = [&](int c) {
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            preG[q] = __builtin_amdgcn_raw_buffer_load_b128(rg, aoff[q] >= 0x40000000u ? 0x80000000u : (aoff[q] + c * 8) * 2, 0, 0);
            preK[q] = __builtin_amdgcn_raw_buffer_load_b64(rk, aoff[q] >= 0x40000000u ? 0x80000000u : aoff[q] + c * 8, 0, 0);
        }
    };
    auto commitA = [&](unsigned char* buf) {
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            const int pix = tid + q * 256;
            if (pix < S_NPIX) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const unsigned v0 = preG[q][2 * hh], v1 = preG[q][2 * hh + 1], k = preK[q][hh] & 0x03030303u;
                    u32x4 c4;                                          // (value, 0) pairs of channels 4 hh .. 4 hh + 3
                    c4[0] = v0 & 0xffffu; c4[1] = v0 >> 16; c4[2] = v1 & 0xffffu; c4[3] = v1 >> 16;
                    *reinterpret_cast<u32x4*>(buf + hh * S_APLANE + pix * 16) = c4;
                    // index word: the position of channel i's value in bits 4 i .. 4 i + 1 (kept slot 2 i), slot 2 i + 1 holds a zero
                    const unsigned iw = (k & 3u) | ((k >> 4) & 0x30u) | ((k >> 8) & 0x300u) | ((k >> 12) & 0x3000u);
                    *reinterpret_cast<unsigned*>(buf + 2 * S_APLANE + (hh * S_NPIX + pix) * 4) = iw;
                }
            }
        }
    };
    // ---- weights: slab (chunk, wy) = 24 pieces of 1 KB; wave w moves the pieces w, w + 4, ...; the image is in LDS order
    const unsigned long img_addr = (unsigned long)p.img;
    const r_u32x4 rb = {(unsigned)img_addr, (unsigned)(img_addr >> 32) & 0xffffu, (unsigned)((long)nts * chunks * 3 * S_SLAB), 0x00020000u};
    auto gldsB = [&](int c, int wy, int slot) {
        const unsigned soff = (unsigned)((((long)nt * chunks + c) * 3 + wy) * S_SLAB);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int k = wave + 4 * j;
            glds16(rb, sB_addr + (unsigned)(slot * S_SLAB + k * 1024), (unsigned)(lane * 16), soff + (unsigned)(k * 1024));
        }
    };

    int abase[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int P = (wm * MI + mi) * 32 + nl;
        abase[mi] = (P / S_TP) * S_HP + (P % S_TP);                 // halo pixel of the tile pixel's (wy, wx) = (0, 0) neighbour
    }
    const int bbase = h * 2 * 2048 + nl * 16 + wn * NI * 512;

    gldsB(0, 0, 0);
    fetchA(0);
    commitA(sA);
    dma_wait();
    __syncthreads();
    for (int c = 0; c < chunks; ++c) {
        const unsigned char* ab = sA + (c & 1) * S_ABUF;
        const bool more = c + 1 < chunks;
#pragma unroll
        for (int wy = 0; wy < 3; ++wy) {
            const int slot = (c + wy) & 1;                            // (3 c + wy) & 1
            if (wy < 2) gldsB(c, wy + 1, slot ^ 1);
            else if (more) gldsB(c + 1, 0, slot ^ 1);
            if (wy == 0 && more) fetchA(c + 1);
            const unsigned char* bs = sB + slot * S_SLAB + bbase;
#pragma unroll
            for (int wx = 0; wx < 3; ++wx) {
                bf16x8 a[MI];
                int ix[MI];
                bf16x16 b[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int hp = abase[mi] + wy * S_HP + wx;
                    a[mi] = *reinterpret_cast<const bf16x8*>(ab + h * S_APLANE + hp * 16);
                    ix[mi] = *reinterpret_cast<const int*>(ab + 2 * S_APLANE + (h * S_NPIX + hp) * 4);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const u32x4 lo = *reinterpret_cast<const u32x4*>(bs + wx * 8192 + ni * 512);
                    const u32x4 hi = *reinterpret_cast<const u32x4*>(bs + wx * 8192 + ni * 512 + 2048);
                    unsigned t[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    b[ni] = *reinterpret_cast<const bf16x16*>(t);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a[mi], b[ni], acc[mi][ni], ix[mi], 0, 0);
            }
            if (wy == 2 && more) commitA(sA + ((c + 1) & 1) * S_ABUF);
            dma_wait();                                    // the next window row has landed ...
            __syncthreads();                               // ... and everyone is done with this one (slot and A buffer free)
        }
    }
    // ---- epilogue: per-wave private LDS scratch (the loop's last barrier released the tiles)
    float* elds = reinterpret_cast<float*>(smem_raw) + wave * (32 * (NI * 32 + EPI_PAD));
    const int H = 2 * p.Hp, W = 2 * p.Wp;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        epilogue_via_lds<NI, false>(acc[mi], elds, lane, [&](int row, int c, float4 v) {
            const int P = (wm * MI + mi) * 32 + row;
            const int a_ = ty0 + P / S_TP, b_ = tx0 + P % S_TP, cls = wn * NI + (c >> 5);
            if (a_ >= p.Hp || b_ >= p.Wp) return;
            const long o = (((long)n * H + 2 * a_ + (cls >> 1)) * W + 2 * b_ + (cls & 1)) * p.Ci + nt * 32 + (c & 31);
            if (p.act) {
                float4 m;
                if (p.flags & NIMG_BF16_MASK) {
                    const bf16x4 mb = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.act) + o);
                    m = make_float4((float)mb[0], (float)mb[1], (float)mb[2], (float)mb[3]);
                } else {
                    m = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.act) + o);
                }
                v.x *= m.x > 0.f ? 1.0f : p.alpha; v.y *= m.y > 0.f ? 1.0f : p.alpha;
                v.z *= m.z > 0.f ? 1.0f : p.alpha; v.w *= m.w > 0.f ? 1.0f : p.alpha;
            }
            if (p.flags & NIMG_BF16_OUT) {
                bf16x4 ob;
                ob[0] = (__bf16)v.x; ob[1] = (__bf16)v.y; ob[2] = (__bf16)v.z; ob[3] = (__bf16)v.w;
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.out) + o) = ob;
            } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + o) = v;
            }
        });
    }
}


// ---- 16-accumulator form (round 6): one workgroup per CU, a wave holds 4 x 4 fragments = TWO 16 x 16 pooled sub-tiles' share
// against all four parity classes: per window 16 matrix instructions for the same 24 KB weight slab the 8-accumulator form
// streams for 8 (profiles/r06_conv3_loop_anatomy.txt: a wave spends ~100 cycles per LDS-DMA piece it requests, 600 cycles per
// slab - what bounds the 8-accumulator form at half the pipe).  The two sub-tiles are neighbours in the linear (image, tile) order:
// side by side in an image, or the same tile of consecutive images (16 x 16 pooled layers); each keeps its own 18 x 18 halo.
constexpr int W_NT = 2;                                                   // sub-tiles per workgroup
constexpr int W_APLANE = W_NT * S_NPIX * 16, W_ABUF = 2 * W_APLANE + 2 * W_NT * S_NPIX * 4;
constexpr int W_LDS_TILES = 2 * S_SLAB + 2 * W_ABUF;
constexpr int W_LDS_EPI = 4 * 32 * (128 + EPI_PAD) * 4;
constexpr int W_LDS = W_LDS_TILES > W_LDS_EPI ? W_LDS_TILES : W_LDS_EPI;

__global__ __launch_bounds__(256, 1) void conv5_dgrad_sparse16_kernel(const Dg5sParams p) {
    constexpr int MI = 4, NI = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* sB = smem_raw;
    unsigned char* sA = smem_raw + 2 * S_SLAB;
    const unsigned sB_addr = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, nl = lane & 31;
    const int nts = p.Ci / 32, chunks = p.Cz / 8;
    int bid = xcd_order(blockIdx.x);
    const int nt = bid % nts;
    bid /= nts;
    const int tiles = p.tiles_y * p.tiles_x;
    const long total = (long)p.N * tiles;                 // sub-tiles in all
    const long L0 = (long)W_NT * bid;                     // this workgroup's first sub-tile

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.0f;

    // ---- A tile staging: items = halo pixels of both sub-tiles
    constexpr int NPIX2 = W_NT * S_NPIX;
    constexpr int AP = (NPIX2 + 255) / 256;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.g), 0,
                                                                        (int)((long)p.N * p.Hp * p.Wp * p.Cz * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.idx), 0,
                                                                        (int)((long)p.N * p.Hp * p.Wp * p.Cz), 0x00020000);
    unsigned aoff[AP];
#pragma unroll
    for (int q = 0; q < AP; ++q) {
        const int item = tid + q * 256, sub = item / S_NPIX, pix = item % S_NPIX;
        const long L = L0 + sub;
        const int n = (int)(L / tiles), tile = (int)(L % tiles);
        const int gy = (tile / p.tiles_x) * S_TP - 1 + pix / S_HP, gx = (tile % p.tiles_x) * S_TP - 1 + pix % S_HP;
        const bool ok = (item < NPIX2) & (L < total) & ((unsigned)gy < (unsigned)p.Hp) & ((unsigned)gx < (unsigned)p.Wp);
        aoff[q] = ok ? (unsigned)(((n * p.Hp + gy) * p.Wp + gx) * p.Cz) : 0x40000000u;
    }
    u32x4 preG[AP];
    u32x2 preK[AP];
    auto fetchA = [&](int c) {
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            preG[q] = __builtin_amdgcn_raw_buffer_load_b128(rg, aoff[q] >= 0x40000000u ? 0x80000000u : (aoff[q] + c * 8) * 2, 0, 0);
            preK[q] = __builtin_amdgcn_raw_buffer_load_b64(rk, aoff[q] >= 0x40000000u ? 0x80000000u : aoff[q] + c * 8, 0, 0);
        }
    };
    auto commitA = [&](unsigned char* buf) {
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            const int pix = tid + q * 256;                 // = sub * S_NPIX + halo pixel: the planes hold both sub-tiles back to back
            if (pix < NPIX2) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const unsigned v0 = preG[q][2 * hh], v1 = preG[q][2 * hh + 1], k = preK[q][hh] & 0x03030303u;
                    u32x4 c4;
                    c4[0] = v0 & 0xffffu; c4[1] = v0 >> 16; c4[2] = v1 & 0xffffu; c4[3] = v1 >> 16;
                    *reinterpret_cast<u32x4*>(buf + hh * W_APLANE + pix * 16) = c4;
                    const unsigned iw = (k & 3u) | ((k >> 4) & 0x30u) | ((k >> 8) & 0x300u) | ((k >> 12) & 0x3000u);
                    *reinterpret_cast<unsigned*>(buf + 2 * W_APLANE + (hh * NPIX2 + pix) * 4) = iw;
                }
            }
        }
    };
    const unsigned long img_addr = (unsigned long)p.img;
    const r_u32x4 rb = {(unsigned)img_addr, (unsigned)(img_addr >> 32) & 0xffffu, (unsigned)((long)nts * chunks * 3 * S_SLAB), 0x00020000u};
    auto gldsB = [&](int c, int wy, int slot) {
        const unsigned soff = (unsigned)((((long)nt * chunks + c) * 3 + wy) * S_SLAB);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int k = wave + 4 * j;
            glds16(rb, sB_addr + (unsigned)(slot * S_SLAB + k * 1024), (unsigned)(lane * 16), soff + (unsigned)(k * 1024));
        }
    };

    int abase[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int P = (wave * MI + mi) * 32 + nl;          // 0 .. 511: sub-tile P >> 8, pixel P & 255 of it
        const int loc = P & 255;
        abase[mi] = (P >> 8) * S_NPIX + (loc / S_TP) * S_HP + (loc % S_TP);
    }
    const int bbase = h * 2 * 2048 + nl * 16;

    gldsB(0, 0, 0);
    fetchA(0);
    commitA(sA);
    dma_wait();
    __syncthreads();
    for (int c = 0; c < chunks; ++c) {
        const unsigned char* ab = sA + (c & 1) * W_ABUF;
        const bool more = c + 1 < chunks;
#pragma unroll
        for (int wy = 0; wy < 3; ++wy) {
            const int slot = (c + wy) & 1;
            if (wy < 2) gldsB(c, wy + 1, slot ^ 1);
            else if (more) gldsB(c + 1, 0, slot ^ 1);
            if (wy == 0 && more) fetchA(c + 1);
            const unsigned char* bs = sB + slot * S_SLAB + bbase;
#pragma unroll
            for (int wx = 0; wx < 3; ++wx) {
                bf16x8 a[MI];
                int ix[MI];
                bf16x16 b[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int hp = abase[mi] + wy * S_HP + wx;
                    a[mi] = *reinterpret_cast<const bf16x8*>(ab + h * W_APLANE + hp * 16);
                    ix[mi] = *reinterpret_cast<const int*>(ab + 2 * W_APLANE + (h * NPIX2 + hp) * 4);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const u32x4 lo = *reinterpret_cast<const u32x4*>(bs + wx * 8192 + ni * 512);
                    const u32x4 hi = *reinterpret_cast<const u32x4*>(bs + wx * 8192 + ni * 512 + 2048);
                    unsigned t[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    b[ni] = *reinterpret_cast<const bf16x16*>(t);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a[mi], b[ni], acc[mi][ni], ix[mi], 0, 0);
            }
            if (wy == 2 && more) commitA(sA + ((c + 1) & 1) * W_ABUF);
            dma_wait();
            __syncthreads();
        }
    }
    float* elds = reinterpret_cast<float*>(smem_raw) + wave * (32 * (NI * 32 + EPI_PAD));
    const int H = 2 * p.Hp, W = 2 * p.Wp;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        epilogue_via_lds<NI, false>(acc[mi], elds, lane, [&](int row, int c, float4 v) {
            const int P = (wave * MI + mi) * 32 + row, loc = P & 255;
            const long L = L0 + (P >> 8);
            if (L >= total) return;
            const int n = (int)(L / tiles), tile = (int)(L % tiles);
            const int a_ = (tile / p.tiles_x) * S_TP + loc / S_TP, b_ = (tile % p.tiles_x) * S_TP + loc % S_TP, cls = c >> 5;
            if (a_ >= p.Hp || b_ >= p.Wp) return;
            const long o = (((long)n * H + 2 * a_ + (cls >> 1)) * W + 2 * b_ + (cls & 1)) * p.Ci + nt * 32 + (c & 31);
            if (p.act) {
                float4 m;
                if (p.flags & NIMG_BF16_MASK) {
                    const bf16x4 mb = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.act) + o);
                    m = make_float4((float)mb[0], (float)mb[1], (float)mb[2], (float)mb[3]);
                } else {
                    m = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.act) + o);
                }
                v.x *= m.x > 0.f ? 1.0f : p.alpha; v.y *= m.y > 0.f ? 1.0f : p.alpha;
                v.z *= m.z > 0.f ? 1.0f : p.alpha; v.w *= m.w > 0.f ? 1.0f : p.alpha;
            }
            if (p.flags & NIMG_BF16_OUT) {
                bf16x4 ob;
                ob[0] = (__bf16)v.x; ob[1] = (__bf16)v.y; ob[2] = (__bf16)v.z; ob[3] = (__bf16)v.w;
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.out) + o) = ob;
            } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + o) = v;
            }
        });
    }
}

}  // namespace

extern "C" {

size_t nimg_conv5_dgrad_sparse_image_bytes(int cin, int cout) {
    if (cin <= 0 || cout <= 0 || (cin % 32) || (cout % 8)) return 0;
    return (size_t)(cin / 32) * (cout / 8) * 3 * S_SLAB;
}

int nimg_conv5_dgrad_sparse_weights(const float* w, void* image, int cin, int cout, void* stream) {
    if (!w || !image || cin <= 0 || cout <= 0 || (cin % 32) || (cout % 8)) return NIMG_ERR_ARG;
    const long total = (long)nimg_conv5_dgrad_sparse_image_bytes(cin, cout) / 2;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(dgrad5s_weights_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, (hipStream_t)stream, w,
                       (__bf16*)image, cin, cout);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_conv5_dgrad_sparse(const void* g, const unsigned char* idx, int cout, const void* image, void* out, int cin,
                            const void* act_mask, int n, int h, int wd, float alpha, int flags, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!g || !idx || !image || !out || n < 0 || h < 2 || wd < 2 || (h & 1) || (wd & 1) || cin <= 0 || cout <= 0 || (cin % 32) ||
        (cout % 8) || (flags & ~(NIMG_BF16_OUT | NIMG_BF16_MASK)))
        return NIMG_ERR_ARG;
    const long g_elems = (long)n * (h / 2) * (wd / 2) * cout;
    if (g_elems * 2 >= (1l << 31) - 65536 || (long)nimg_conv5_dgrad_sparse_image_bytes(cin, cout) >= (1l << 31) - 65536) return NIMG_ERR_ARG;
    Dg5sParams p;
    p.g = g; p.idx = idx; p.img = image; p.out = out; p.act = act_mask;
    p.Cz = cout; p.Ci = cin; p.N = n; p.Hp = h / 2; p.Wp = wd / 2; p.flags = flags; p.alpha = alpha;
    p.tiles_y = nimg::cdiv(p.Hp, S_TP);
    p.tiles_x = nimg::cdiv(p.Wp, S_TP);
    const long blocks = (long)(cin / 32) * p.tiles_y * p.tiles_x * n;
    // NIMG_DGRAD5S_ACC16=1: the 16-accumulator form (two sub-tiles per workgroup, one workgroup per CU)
    static const bool acc16 = getenv("NIMG_DGRAD5S_ACC16") != nullptr;
    if (acc16) {
        const long subtiles = (long)p.tiles_y * p.tiles_x * n;
        const long blocks16 = (long)(cin / 32) * ((subtiles + W_NT - 1) / W_NT);
        (void)hipFuncSetAttribute((const void*)conv5_dgrad_sparse16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
        hipLaunchKernelGGL(conv5_dgrad_sparse16_kernel, dim3((unsigned)blocks16), dim3(256), W_LDS, (hipStream_t)stream, p);
        NIMG_CHECK_LAUNCH();
        return NIMG_OK;
    }
    static const bool block42 = getenv("NIMG_DGRAD5S_BLOCK42") != nullptr;        // A/B: the 4 x 2 fragment block (round 6, slower)
    auto kern = block42 ? conv5_dgrad_sparse_kernel<4> : conv5_dgrad_sparse_kernel<2>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), S_LDS, (hipStream_t)stream, p);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

}  // extern "C"
