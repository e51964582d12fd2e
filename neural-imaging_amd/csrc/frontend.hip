// FAN front end (models/forensics.py:62-70): ConstrainedConv2D 5x5 3->3 (models/layers.py:36-57) and the first
// convolution 5x5 3->32 + LeakyReLU + MaxPool2D, forward and backward.  320 images of 256x256 per training step make these
// the widest tensors of the whole channel (21 M pixels), while they hold under 3 % of its FLOPs: every pass here is bound by
// HBM bytes or by plain VALU issue, not by the matrix core.  Decomposition ("front end"):
//
//   cconv_kernel         ConstrainedConv2D forward (SYMMETRIC pad) AND its input gradient (zero pad, flipped / transposed
//                        filter): float32 VALU stencil, 4 px x 3 channels per thread, SGPR weights, LDS row band
//   cdgrad_border_kernel the terms of the input gradient that the SYMMETRIC pad folds back onto the 2-pixel image border
//
// This file is compiled with -fno-slp-vectorize (csrc/Makefile): the stencil wants v_fmac_f32 v, s, v; hipcc's SLP pass
// otherwise builds v_pk_fma_f32 pairs with a v_mov per operand (gfx950's VALU already retires one f32 FMA per lane and clock,
// so the packed form buys nothing and the moves cost issue slots).
#include <stdlib.h>
#include "common.h"

namespace {

using namespace nimg;

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct CConvParams {
    const float* in;      // (N,H,W,3)
    const float* w;       // [5][5][3][3] = [ky][kx][ci][co], 225 floats (wave-uniform: read through the scalar cache)
    float* out_f32;       // optional (N,H,W,3)
    uint2* out_c4;        // optional (N,H,W) x {bf16 c0, c1, c2, 1.0}: the 8-byte pixel the throughput-mode conv1 kernels read
    int N, H, W, pad_mode, tiles_y, tiles_x;
};

// Workgroup = 256 threads = TPR threads along a row x (256 / TPR) thread rows; a thread owns 4 adjacent pixels x RPT rows x 3
// output channels.  Tile = (256 / TPR * RPT) rows x (4 TPR) columns, staged with its 2-pixel halo as float32 [row][pixel][3];
// a thread's 8-pixel input window of one tile row is 6 aligned ds_read_b128 (lane stride 48 B: the sixteen lanes of every
// b128 group cover the 64 banks once), which feed 180 FMAs per kernel row with the row's 45 weights in SGPRs.
template <int TPR, int RPT>
__global__ __launch_bounds__(256) void cconv_kernel(const float* __restrict__ p_in, const float* __restrict__ p_w,
                                                    float* __restrict__ p_out_f32, uint2* __restrict__ p_out_c4,
                                                    const CConvParams p) {
    constexpr int TW = 4 * TPR, TROWS = 256 / TPR, TR = TROWS * RPT, HR = TR + 4, HC = TW + 4;
    constexpr int RS = (HC * 3 + 3) / 4 * 4;                 // floats per tile row, 16-byte multiple
    constexpr int NPIX = HR * HC, PPT = (NPIX + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [HR][RS], then the 5 x 48 filter taps
    float* wl = tile + HR * RS;                                       // [ky][48]: 45 weights of a kernel row + 3 pad
    const int tid = threadIdx.x, tx = tid % TPR, ty = tid / TPR;
    // The 225 filter weights are wave-uniform.  As SGPR operands (v_fmac_f32 v, s, v) the FMAs issue at ~0.6x the rate of the
    // all-VGPR form on gfx950 (tools/probe/fma_rate.hip: 64 vs 39 us for the same instruction count at 4 waves per SIMD), and
    // the FMA stream is what bounds this kernel - so a kernel row's 45 weights are fetched from LDS with broadcast reads
    // (every lane the same address) into VGPRs instead.
    for (int i = tid; i < 5 * 48; i += 256) wl[i] = (i % 48) < 45 ? p_w[(i / 48) * 45 + i % 48] : 0.f;
    const int tiles = p.tiles_y * p.tiles_x;
    const int total = p.N * tiles;                     // < 2^31 / (12 * 32): the entry point bounds the batch

    // Staging.  Halo pixels travel as ONE 12-byte buffer load each; zero-pad pixels and everything outside the tensor are
    // offsets beyond num_records, which the hardware answers with zeros (a predicated flat load costs hipcc a saveexec /
    // branch / zero-fill scaffold per pixel).  A thread owns one tile COLUMN (its x mapping is resolved once per tile) and
    // walks down the rows (row mapping = a handful of scalar-ish selects), so a pixel costs one add + one load.  The few
    // columns beyond the thread grid (HC - NCOLT) are one extra pixel for the first threads.  Tensor < 2 GB (entry point).
    typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NCOLT = HC <= 128 ? 128 : 256, RG = 256 / NCOLT, RPTS = (HR + RG - 1) / RG;
    constexpr int EXC = HC > NCOLT ? HC - NCOLT : 0, EXN = EXC * HR;
    static_assert(EXN <= 256, "extra halo columns: one pixel per thread");
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p_in), 0,
                                                                         (int)((long)p.N * p.H * p.W * 12), 0x00020000);
    const int sc = tid % NCOLT, sr = tid / NCOLT;                    // staging column / first row of this thread
    const int er = EXC ? tid / (EXC ? EXC : 1) : 0, ec = EXC ? NCOLT + tid % (EXC ? EXC : 1) : 0;
    u32x3 pre[RPTS], pre_x;
    auto byte_x = [&](int gx) {
        const bool ok = map_coord(gx, p.W, p.pad_mode);
        return ok ? (unsigned)(gx * 12) : OOB;
    };
    auto fetch = [&](int t) {
        const int n = t / tiles, tl = t % tiles;
        const int y0 = (tl / p.tiles_x) * TR, x0 = (tl % p.tiles_x) * TW;
        const unsigned xb = sc < HC ? byte_x(x0 - 2 + sc) : OOB;
#pragma unroll
        for (int q = 0; q < RPTS; ++q) {
            int gy = y0 - 2 + q * RG + sr;
            const bool ok = map_coord(gy, p.H, p.pad_mode) & (q * RG + sr < HR);
            const unsigned rowb = (unsigned)((n * p.H + gy) * p.W * 12);
            pre[q] = __builtin_amdgcn_raw_buffer_load_b96(rin, (rowb + xb) | (ok ? 0u : OOB), 0, 0);   // rowb < 2^31: an OOB xb stays OOB
        }
        if constexpr (EXN > 0) {
            int gy = y0 - 2 + er;
            const bool ok = map_coord(gy, p.H, p.pad_mode) & (tid < EXN);
            const unsigned rowb = (unsigned)((n * p.H + gy) * p.W * 12);
            pre_x = __builtin_amdgcn_raw_buffer_load_b96(rin, (rowb + byte_x(x0 - 2 + ec)) | (ok ? 0u : OOB), 0, 0);
        }
    };
    const int first = xcd_order(blockIdx.x);
    if (first < total) fetch(first);
    for (int t = first; t < total; t += gridDim.x) {
        const int n = t / tiles, tl = t % tiles;
        const int y0 = (tl / p.tiles_x) * TR, x0 = (tl % p.tiles_x) * TW;
        __syncthreads();
        if (sc < HC) {
#pragma unroll
            for (int q = 0; q < RPTS; ++q) {
                if (q * RG + sr < HR) {
                    float* d = tile + (q * RG + sr) * RS + sc * 3;
                    d[0] = __uint_as_float(pre[q][0]); d[1] = __uint_as_float(pre[q][1]); d[2] = __uint_as_float(pre[q][2]);
                }
            }
        }
        if constexpr (EXN > 0) {
            if (tid < EXN) {
                float* d = tile + er * RS + ec * 3;
                d[0] = __uint_as_float(pre_x[0]); d[1] = __uint_as_float(pre_x[1]); d[2] = __uint_as_float(pre_x[2]);
            }
        }
        __syncthreads();
        if (t + (int)gridDim.x < total) fetch(t + gridDim.x);

        float acc[RPT][4][3];
#pragma unroll
        for (int r = 0; r < RPT; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[r][j][0] = acc[r][j][1] = acc[r][j][2] = 0.f;
#pragma unroll 1
        for (int ky = 0; ky < 5; ++ky) {
            float wr[48];                                       // [kx][ci][co]: this kernel row's 45 weights (+3 pad)
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const float4 w4 = *reinterpret_cast<const float4*>(wl + ky * 48 + 4 * i);
                wr[4 * i] = w4.x; wr[4 * i + 1] = w4.y; wr[4 * i + 2] = w4.z; wr[4 * i + 3] = w4.w;
            }
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const float4* row = reinterpret_cast<const float4*>(tile + (ty * RPT + r + ky) * RS + tx * 12);
                float v[24];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const float4 q4 = row[i];
                    v[4 * i] = q4.x; v[4 * i + 1] = q4.y; v[4 * i + 2] = q4.z; v[4 * i + 3] = q4.w;
                }
#pragma unroll
                for (int kx = 0; kx < 5; ++kx)
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float xv = v[(j + kx) * 3 + ci];
                            acc[r][j][0] = fmaf(xv, wr[(kx * 3 + ci) * 3 + 0], acc[r][j][0]);
                            acc[r][j][1] = fmaf(xv, wr[(kx * 3 + ci) * 3 + 1], acc[r][j][1]);
                            acc[r][j][2] = fmaf(xv, wr[(kx * 3 + ci) * 3 + 2], acc[r][j][2]);
                        }
            }
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int oy = y0 + ty * RPT + r, ox = x0 + tx * 4;
            if (oy >= p.H || ox >= p.W) continue;
            const long o = ((long)n * p.H + oy) * p.W + ox;
            if (ox + 3 < p.W && (p.W & 3) == 0) {               // whole group inside the row, 16-byte aligned
                if (p_out_f32) {
                    float4* d = reinterpret_cast<float4*>(p_out_f32 + o * 3);
                    d[0] = make_float4(acc[r][0][0], acc[r][0][1], acc[r][0][2], acc[r][1][0]);
                    d[1] = make_float4(acc[r][1][1], acc[r][1][2], acc[r][2][0], acc[r][2][1]);
                    d[2] = make_float4(acc[r][2][2], acc[r][3][0], acc[r][3][1], acc[r][3][2]);
                }
                if (p_out_c4) {
                    bf16x4 c[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        c[j][0] = (__bf16)acc[r][j][0]; c[j][1] = (__bf16)acc[r][j][1]; c[j][2] = (__bf16)acc[r][j][2];
                        c[j][3] = (__bf16)1.0f;
                    }
                    uint4* d = reinterpret_cast<uint4*>(p_out_c4 + o);
                    const uint2 a0 = *reinterpret_cast<const uint2*>(&c[0]), a1 = *reinterpret_cast<const uint2*>(&c[1]);
                    const uint2 a2 = *reinterpret_cast<const uint2*>(&c[2]), a3 = *reinterpret_cast<const uint2*>(&c[3]);
                    d[0] = make_uint4(a0.x, a0.y, a1.x, a1.y);
                    d[1] = make_uint4(a2.x, a2.y, a3.x, a3.y);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (ox + j >= p.W) break;
                    if (p_out_f32) {
                        p_out_f32[(o + j) * 3] = acc[r][j][0];
                        p_out_f32[(o + j) * 3 + 1] = acc[r][j][1];
                        p_out_f32[(o + j) * 3 + 2] = acc[r][j][2];
                    }
                    if (p_out_c4) {
                        bf16x4 c;
                        c[0] = (__bf16)acc[r][j][0]; c[1] = (__bf16)acc[r][j][1]; c[2] = (__bf16)acc[r][j][2];
                        c[3] = (__bf16)1.0f;
                        p_out_c4[o + j] = *reinterpret_cast<const uint2*>(&c);
                    }
                }
            }
        }
    }
}

// Input gradient of the SYMMETRIC-padded filter, border part.  With xp = pad(x): dxp[u][v][i] = sum_{ky,kx,o}
// dc[u-ky][v-kx][o] nf[ky][kx][i][o] on the (H+4) x (W+4) padded domain, and dx[y][x] = sum of dxp over every padded position
// that mirrors onto (y, x): (y+2, x+2) itself - that term is cconv_kernel with the flipped filter and zero padding - plus, on
// the two outermost rows / columns, the mirror images u in {1-y, 2H+1-y}, v in {1-x, 2W+1-x}.  One thread per border pixel
// adds those extra terms to dx (4(H+W)-16 pixels per image: the work is negligible, the reads come from L2).
__global__ void cdgrad_border_kernel(const float* __restrict__ dc, const float* __restrict__ nf, float* __restrict__ dx, int N,
                                     int H, int W) {
    const int per = 4 * W + 4 * (H - 4);                        // 2 top + 2 bottom rows, then 2 + 2 columns of the other rows
    const long total = (long)N * per;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int n = (int)(t / per), b = (int)(t % per);
        int y, x;
        if (b < 4 * W) {
            const int r = b / W;
            y = r < 2 ? r : H - 4 + r;
            x = b % W;
        } else {
            const int c = (b - 4 * W) & 3;
            y = 2 + (b - 4 * W) / 4;
            x = c < 2 ? c : W - 4 + c;
        }
        int us[3], vs[3], nu = 0, nv = 0;
        us[nu++] = y + 2;
        if (y < 2) us[nu++] = 1 - y;
        if (y >= H - 2) us[nu++] = 2 * H + 1 - y;
        vs[nv++] = x + 2;
        if (x < 2) vs[nv++] = 1 - x;
        if (x >= W - 2) vs[nv++] = 2 * W + 1 - x;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int iu = 0; iu < nu; ++iu)
            for (int iv = 0; iv < nv; ++iv) {
                if (iu == 0 && iv == 0) continue;               // the direct term belongs to the main kernel
                const int u = us[iu], v = vs[iv];
                for (int ky = 0; ky < 5; ++ky) {
                    const int py = u - ky;
                    if (py < 0 || py >= H) continue;
                    for (int kx = 0; kx < 5; ++kx) {
                        const int px = v - kx;
                        if (px < 0 || px >= W) continue;
                        const float* g = dc + (((long)n * H + py) * W + px) * 3;
                        const float* wv = nf + (ky * 5 + kx) * 9;            // [i][o]
#pragma unroll
                        for (int o = 0; o < 3; ++o) {
                            a0 = fmaf(g[o], wv[o], a0);
                            a1 = fmaf(g[o], wv[3 + o], a1);
                            a2 = fmaf(g[o], wv[6 + o], a2);
                        }
                    }
                }
            }
        float* d = dx + (((long)n * H + y) * W + x) * 3;
        d[0] += a0; d[1] += a1; d[2] += a2;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// First FAN convolution, throughput mode: 5x5, 3 -> 32, SAME (zeros), + bias + LeakyReLU + MaxPool2D(2) in one pass
// (models/forensics.py:69-70) over the 8-byte bf16 pixels {c0, c1, c2, 1} written by cconv_kernel.
//
// GEMM view per output row: D[cout 32][pixel 32] += A[cout][k] B[k][pixel], v_mfma_f32_32x32x16_bf16, K = 7 steps of 16 =
// 28 (tap, 4 channel) slots for the 25 taps (the 4th channel and the 3 spare slots meet zero weights).  The weights (A) stay
// in 28 VGPRs for the whole kernel; a pixel's B operand is two 8-byte LDS reads of whole pixels at immediate tap offsets -
// no gather, no conversion, no packing.  Slot order: k-steps 0..4 = kernel rows (lane half h takes kx = 2h, 2h+1), k-step 5
// = column kx = 4 of rows 2h, 2h+1, k-step 6 = tap (4,4): inside a k-step the two lane halves differ by ONE constant LDS
// offset, so three per-lane base addresses serve every read.
// Output layout D^T (couts along the accumulator registers, pixels along the lanes).  A unit is 2 output rows x 64 columns in
// FOUR accumulators: lane column nl computes pixel columns 2 nl and 2 nl + 1 (the B operand of the odd column is the even
// column's window one pixel on: 3 reads serve both), so the whole 2x2 pooling window of 16 output channels sits in ONE lane -
// 9 compare / select instructions per pooled value, no lane exchange, nothing computed twice (the first version pooled through
// a DPP exchange with the neighbouring lane, both lanes of a pair doing the same 17 instructions per value: the kernel was
// bound by VALU issue, 394 instructions per 64 pixels; now ~150).  LeakyReLU is monotonic (0 < alpha <= 1, checked by the entry
// point), so the window is pooled on the raw sums and bias + activation are applied to the winner only.  The weight rows are
// permuted (MFMA row m <-> output channel 16 ((m >> 2) & 1) + 4 (m >> 3) + (m & 3)) so that a lane's 16 registers are 16
// CONSECUTIVE channels of one pooled pixel: two 16-byte stores of bf16 values and one of arg-max bytes per lane, every lane.
#ifndef CONV1_WGS
#define CONV1_WGS 3       // resident workgroups per CU the register allocation of conv1_pool_fwd_kernel aims at (A/B: EXTRA=-DCONV1_WGS=2)
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float dpp_swap1(float v) {           // value of lane ^ 1 (quad_perm [1,0,3,2])
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 r;
    r[0] = (__bf16)a; r[1] = (__bf16)b;
    return *reinterpret_cast<const unsigned*>(&r);
}

// tap (ky, kx) of K slot (k-step ks, lane half h, j in {0,1}); ky < 0: spare slot (zero weight)
__device__ __forceinline__ void conv1_slot(int ks, int h, int j, int& ky, int& kx) {
    if (ks < 5) { ky = ks; kx = 2 * h + j; }
    else if (ks == 5) { ky = 2 * h + j; kx = 4; }
    else { ky = (h == 0 && j == 0) ? 4 : -1; kx = 4; }
}

template <int TWD, bool OUT_BF16>
__global__ __launch_bounds__(256, CONV1_WGS) void conv1_pool_fwd_kernel(const void* __restrict__ c4, const float* __restrict__ w,
                                                             const float* __restrict__ bias, void* __restrict__ pooled,
                                                             unsigned char* __restrict__ pidx, int N, int H, int W,
                                                             float alpha, int tiles_y, int tiles_x) {
    constexpr int TRD = 8, HR = TRD + 4, HC = TWD + 4, NPC = HC / 2;            // tile rows / halo rows / halo cols / pixel pairs
    constexpr int PT = NPC <= 64 ? 64 : 128, RG = 256 / PT, RPTS = (HR + RG - 1) / RG;
    constexpr int EXC = NPC > PT ? NPC - PT : 0, EXN = EXC * HR;
    constexpr int UC = TWD / 64, UNITS = (TRD / 2) * UC;
    constexpr unsigned OOB = 0x80000000u;
    static_assert(EXN <= 256 && HC % 2 == 0, "staging layout");
    __shared__ __attribute__((aligned(16))) uint4 tile[HR * NPC];                 // [row][pixel pair]: 8 B per pixel
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, nl = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = tiles_y * tiles_x, total = N * tiles;

    // A operand: MFMA row m = nl holds output channel co_a, K values of the lane half, for the 7 k-steps
    const int co_a = 16 * ((nl >> 2) & 1) + 4 * (nl >> 3) + (nl & 3);
    bf16x8 wa[7];
#pragma unroll
    for (int ks = 0; ks < 7; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int ky, kx;
            conv1_slot(ks, h, j, ky, kx);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                wa[ks][4 * j + c] = (__bf16)((ky >= 0 && c < 3) ? w[((ky * 5 + kx) * 3 + c) * 32 + co_a] : 0.f);
        }
    float br[16];                                     // bias of the 16 couts this lane's accumulator registers hold: 16 h + r
#pragma unroll
    for (int r = 0; r < 16; ++r) br[r] = bias ? bias[16 * h + r] : 0.f;

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(c4), 0, (int)((long)N * H * W * 8),
                                                                         0x00020000);
    const int sp = tid % PT, sr = tid / PT;           // staging: pixel-pair column / first row of this thread
    const int er = EXC ? tid / (EXC ? EXC : 1) : 0, ep = EXC ? PT + tid % (EXC ? EXC : 1) : 0;
    u32x4 pre[RPTS], pre_x;
    auto fetch = [&](int t) {
        const int n = t / tiles, tl = t % tiles;
        const int y0 = (tl / tiles_x) * TRD, x0 = (tl % tiles_x) * TWD;
        const int gx = x0 - 2 + 2 * sp;               // W and x0 are even: a pair is inside or outside as a whole
        const unsigned xb = (sp < NPC && (unsigned)gx < (unsigned)W) ? (unsigned)(gx * 8) : OOB;
#pragma unroll
        for (int q = 0; q < RPTS; ++q) {
            const int gy = y0 - 2 + q * RG + sr;
            const bool ok = ((unsigned)gy < (unsigned)H) & (q * RG + sr < HR);
            pre[q] = __builtin_amdgcn_raw_buffer_load_b128(rin, ((unsigned)((n * H + gy) * W * 8) + xb) | (ok ? 0u : OOB), 0, 0);
        }
        if constexpr (EXN > 0) {
            const int gy = y0 - 2 + er, gx2 = x0 - 2 + 2 * ep;
            const bool ok = ((unsigned)gy < (unsigned)H) & ((unsigned)gx2 < (unsigned)W) & (tid < EXN);
            pre_x = __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? (unsigned)(((n * H + gy) * W + gx2) * 8) : OOB, 0, 0);
        }
    };
    const int first = xcd_order(blockIdx.x);
    if (first < total) fetch(first);
    const int Hp = H >> 1, Wp = W >> 1;
    for (int t = first; t < total; t += gridDim.x) {
        const int n = t / tiles, tl = t % tiles;
        const int y0 = (tl / tiles_x) * TRD, x0 = (tl % tiles_x) * TWD;
        __syncthreads();
        if (sp < NPC) {
#pragma unroll
            for (int q = 0; q < RPTS; ++q)
                if (q * RG + sr < HR) tile[(q * RG + sr) * NPC + sp] = *reinterpret_cast<const uint4*>(&pre[q]);
        }
        if constexpr (EXN > 0) {
            if (tid < EXN) tile[er * NPC + ep] = *reinterpret_cast<const uint4*>(&pre_x);
        }
        __syncthreads();
        if (t + (int)gridDim.x < total) fetch(t + gridDim.x);

        const unsigned char* tb = reinterpret_cast<const unsigned char*>(tile);
        for (int u = wave; u < UNITS; u += 4) {
            const int ur = u / UC, uc = u % UC;
            // pixel (row 2ur [+1], column 64uc + 2nl [+1]) tap (ky, kx) sits at tile[(2ur [+1] + ky)][64uc + 2nl [+1] + kx]
            const int a0 = ((2 * ur) * HC + 64 * uc + 2 * nl) * 8;
            const int baseA = a0 + h * 16, baseB = a0 + h * (2 * HC * 8);
            f32x16 acc[2][2];                                      // [output row][column parity]
            auto rd = [&](int off) { return *reinterpret_cast<const u32x2*>(tb + off); };
            auto op = [](u32x2 lo, u32x2 hi) {
                u32x4 b;
                b[0] = lo[0]; b[1] = lo[1]; b[2] = hi[0]; b[3] = hi[1];
                return *reinterpret_cast<const bf16x8*>(&b);
            };
#pragma unroll
            for (int ks = 0; ks < 7; ++ks)
#pragma unroll
                for (int row = 0; row < 2; ++row) {
                    bf16x8 b0, b1;                                 // B operands of the even / odd pixel column
                    if (ks < 5) {                                  // taps kx = 2h, 2h + 1 (even column) / one pixel on (odd column)
                        const int o = baseA + (ks + row) * HC * 8;
                        const u32x2 q0 = rd(o), q1 = rd(o + 8), q2 = rd(o + 16);
                        b0 = op(q0, q1);
                        b1 = op(q1, q2);
                    } else if (ks == 5) {                          // kx = 4 of kernel rows 2h, 2h + 1
                        const int o = baseB + row * HC * 8 + 4 * 8;
                        b0 = op(rd(o), rd(o + HC * 8));
                        b1 = op(rd(o + 8), rd(o + HC * 8 + 8));
                    } else {                                       // tap (4, 4)
                        const int o = a0 + ((4 + row) * HC + 4) * 8;
                        const u32x2 q0 = rd(o), q1 = rd(o + 8);
                        b0 = op(q0, q0);
                        b1 = op(q1, q1);
                    }
                    if (ks == 0) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[row][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0], b0, zero, 0, 0, 0);
                        acc[row][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0], b1, zero, 0, 0, 0);
                    } else {
                        acc[row][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks], b0, acc[row][0], 0, 0, 0);
                        acc[row][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks], b1, acc[row][1], 0, 0, 0);
                    }
                }
            // the 2x2 window in window order (top-left, top-right, bottom-left, bottom-right): the first maximum wins, like
            // nimg_maxpool2_fwd; compare + select throughout (each compare also yields the arg-max bit)
            float mx[16];
            unsigned kk[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float tl = acc[0][0][r], tr = acc[0][1][r], bl = acc[1][0][r], bq = acc[1][1][r];
                const bool rt = tr > tl, rb = bq > bl;
                const float mt = rt ? tr : tl, mb = rb ? bq : bl;
                const unsigned kt = rt ? 1u : 0u, kb = rb ? 3u : 2u;
                const bool bot = mb > mt;
                const float v = (bot ? mb : mt) + br[r];
                mx[r] = __builtin_fmaxf(v, alpha * v);            // LeakyReLU, 0 < alpha <= 1
                kk[r] = bot ? kb : kt;
            }
            const int py = (y0 + 2 * ur) >> 1, px = ((x0 + 64 * uc) >> 1) + nl;
            if (py < Hp && px < Wp) {
                const long po = ((long)(n * Hp + py) * Wp + px) * 32 + 16 * h;
                if (pidx) {                       // 2 bits per channel: byte c >> 2 of the pixel's 8, bits 2 (c & 3)
                    unsigned ib = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) ib |= kk[r] << (2 * r);
                    *reinterpret_cast<unsigned*>(pidx + (po >> 2)) = ib;
                }
                if constexpr (OUT_BF16) {
                    unsigned pb[8];
#pragma unroll
                    for (int g = 0; g < 8; ++g) pb[g] = pk_bf16(mx[2 * g], mx[2 * g + 1]);
                    uint4* d = reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(pooled) + po);
                    d[0] = make_uint4(pb[0], pb[1], pb[2], pb[3]);
                    d[1] = make_uint4(pb[4], pb[5], pb[6], pb[7]);
                } else {
                    float4* d = reinterpret_cast<float4*>(reinterpret_cast<float*>(pooled) + po);
#pragma unroll
                    for (int g = 0; g < 4; ++g) d[g] = make_float4(mx[4 * g], mx[4 * g + 1], mx[4 * g + 2], mx[4 * g + 3]);
                }
            }
        }
    }
}

template <int TPR, int RPT>
int launch_cconv(CConvParams p, hipStream_t s) {
    constexpr int TW = 4 * TPR, TR = (256 / TPR) * RPT, HR = TR + 4, HC = TW + 4, RS = (HC * 3 + 3) / 4 * 4;
    constexpr size_t lds = ((size_t)HR * RS + 5 * 48) * sizeof(float);
    p.tiles_y = cdiv(p.H, TR);
    p.tiles_x = cdiv(p.W, TW);
    const long total = (long)p.N * p.tiles_y * p.tiles_x;
    static const int capmul = getenv("NIMG_CCONV_CAP") ? atoi(getenv("NIMG_CCONV_CAP")) : 0;
    const long cap = 256L * (capmul > 0 ? capmul : (lds > 40000 ? 3 : (lds > 20000 ? 4 : 8)));   // persistent: the resident workgroups
    auto k = cconv_kernel<TPR, RPT>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)(total < cap ? total : cap)), dim3(256), lds, s, p.in, p.w, p.out_f32, p.out_c4, p);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight + bias gradient of the first FAN convolution from the POOLED gradient (throughput mode):
//   dw[ky][kx][ci][o] = sum_{n,y,x} c[n][y+ky-2][x+kx-2][ci] dz[n][y][x][o],   db[o] = sum dz[n][y][x][o],
//   dz[y][x][o] = g[y/2][x/2][o] if argmax[y/2][x/2][o] == 2 (y & 1) + (x & 1) else 0      (MaxPool2D routing; g carries LReLU')
// GEMM view: D[(tap, 4 ch) 128][o 32] += A[(tap, ch)][16 pixels] B[16 pixels][o]: 4 M-fragments x v_mfma_f32_32x32x16_bf16 per
// run of 16 pixels.  Both operands are pixel-major in memory and K(=pixel)-major in the MFMA, which is what
// ds_read_b64_tr_b16 delivers: A straight from the 8-byte {c0,c1,c2,1} pixels (a 16-lane group reads 4 pixels x 4 taps x 4
// channels; the tap is a per-lane offset), B from the un-pooled, arg-max-masked gradient tile the staging pass builds in LDS
// with packed bit operations (no float conversions).  The constant channel of the pixel makes row (tap (2,2), ch 3) of D the
// bias gradient.  Waves split the pixel runs of a tile; every wave keeps its 64 accumulator registers over all its tiles and
// writes one slab (final dw layout) at the end; a fixed-order slab reduction follows (deterministic).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// ds_read_b64_tr_b16: in a 16-lane group lane g supplies the 8-byte aligned address of row g >> 2, columns 4 (g & 3) .. + 3 of
// a 4 x 16 block of 16-bit elements and receives column g (4 elements).  Two reads = the 8 K values of an MFMA operand.
__device__ __forceinline__ bf16x8 tr_read8(const unsigned char* p0, const unsigned char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p1);
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return *reinterpret_cast<bf16x8*>(&r);
}

// The arg-max of the first pooling layer is stored as 2 bits per value (8 bytes per pooled pixel of 32 channels instead of 32:
// -0.38 GB per step over conv1_pool_fwd / conv1_wgrad_pooled / conv1_dgrad_pooled, whose traffic bounds them).  The 16 bits of
// an 8-channel chunk -> the two words of byte-per-channel codes the routing below works on.
__device__ __forceinline__ u32x2 argmax2_bytes(unsigned short k16) {
    const unsigned k = k16;
    auto spread = [](unsigned x) {                        // x = 4 fields of 2 bits -> one field per byte
        const unsigned t = (x | (x << 12)) & 0x000F000Fu;
        return (t | (t << 6)) & 0x03030303u;
    };
    u32x2 r;
    r[0] = spread(k & 0xFFu);
    r[1] = spread(k >> 8);
    return r;
}

// 4 arg-max bytes (values 0..3) -> 0xFF in every byte that equals pos
__device__ __forceinline__ unsigned eq_bytes(unsigned k, unsigned pos) {
    const unsigned x = k ^ (pos * 0x01010101u);
    return (((x | (x >> 1)) & 0x01010101u) ^ 0x01010101u) * 0xFFu;
}

constexpr int F_TR = 8, F_TW = 64, F_HR = F_TR + 4, F_HC = F_TW + 4;          // tile rows / cols, with the 2-pixel halo
constexpr int F_C4_BYTES = F_HR * F_HC * 8, F_DZ_BYTES = F_TR * F_TW * 64;
constexpr int F_DW = 25 * 3 * 32;                                             // floats of one dw slab

template <bool GB>      // GB: the pooled gradient is stored as bf16 (else float32)
__global__ __launch_bounds__(256) void conv1_wgrad_pooled_kernel(const void* __restrict__ c4, const void* __restrict__ gp,
                                                                 const unsigned char* __restrict__ pidx,
                                                                 float* __restrict__ dw_slabs, float* __restrict__ db_slabs,
                                                                 int N, int H, int W, int tiles_y, int tiles_x, int per_wg) {
    constexpr unsigned OOB = 0x80000000u;
    __shared__ __attribute__((aligned(16))) unsigned char smem[F_C4_BYTES + F_DZ_BYTES];
    unsigned char* sC = smem;                       // [F_HR][F_HC] x 8 B
    unsigned char* sZ = smem + F_C4_BYTES;          // [F_TR][F_TW] x 64 B (32 o, bf16)
    const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5, q1 = (lane >> 4) & 1, g16 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = tiles_y * tiles_x, total = N * tiles;
    const int Hp = H >> 1, Wp = W >> 1;
    const int w_begin = blockIdx.x * per_wg, w_end = min(total, w_begin + per_wg);

    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(c4), 0, (int)((long)N * H * W * 8),
                                                                        0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(gp), 0,
                                                                        (int)((long)N * Hp * Wp * 32 * (GB ? 2 : 4)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(pidx), 0,
                                                                        (int)((long)N * Hp * Wp * 8), 0x00020000);
    // per-lane parts of the transpose-read addresses
    const int tl = 4 * q1 + (g16 & 3);                               // this lane's tap inside an 8-tap M fragment
    int a_off[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int tap = min(8 * f + tl, 24);                         // spare slots re-read tap 24 (their D rows are dropped)
        a_off[f] = ((tap / 5) * F_HC + (tap % 5) + 8 * kh + (g16 >> 2)) * 8;
    }
    const int z_off = (8 * kh + (g16 >> 2)) * 64 + (16 * q1 + 4 * (g16 & 3)) * 2;

    f32x16 acc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    constexpr int C_ITEMS = F_HR * (F_HC / 2), C_PT = (C_ITEMS + 255) / 256;      // 16-byte pixel pairs of the c tile
    constexpr int G_PT = (F_TR / 2) * (F_TW / 2) * 4 / 256;                       // (pooled pixel, 8-channel chunk) items
    static_assert(G_PT * 256 == (F_TR / 2) * (F_TW / 2) * 4, "gradient tile must divide over the threads");
    u32x4 pc[C_PT], pg[G_PT][GB ? 1 : 2];
    u32x2 pk[G_PT];
    auto fetch = [&](int t) {
        const int n = t / tiles, tile = t % tiles;
        const int y0 = (tile / tiles_x) * F_TR, x0 = (tile % tiles_x) * F_TW;
#pragma unroll
        for (int q = 0; q < C_PT; ++q) {
            const int item = tid + q * 256, row = item / (F_HC / 2), pr = item % (F_HC / 2);
            const int gy = y0 - 2 + row, gx = x0 - 2 + 2 * pr;                    // W, x0 even: a pair is in or out as a whole
            const bool ok = (item < C_ITEMS) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
            pc[q] = __builtin_amdgcn_raw_buffer_load_b128(rc, ok ? (unsigned)(((n * H + gy) * W + gx) * 8) : OOB, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < G_PT; ++q) {
            const int item = tid + q * 256, pp = item >> 2, ch = item & 3;
            const int py = (y0 >> 1) + pp / (F_TW / 2), px = (x0 >> 1) + pp % (F_TW / 2);
            const bool ok = (py < Hp) & (px < Wp);
            const unsigned e = (unsigned)(((n * Hp + py) * Wp + px) * 32 + ch * 8);      // element index of the 8-channel chunk
            if constexpr (GB) {
                pg[q][0] = __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? e * 2 : OOB, 0, 0);
            } else {
                pg[q][0] = __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? e * 4 : OOB, 0, 0);
                pg[q][1] = __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? e * 4 + 16 : OOB, 0, 0);
            }
            pk[q] = argmax2_bytes(__builtin_amdgcn_raw_buffer_load_b16(rk, ok ? e >> 2 : OOB, 0, 0));
        }
    };
    if (w_begin < w_end) fetch(w_begin);
    for (int t = w_begin; t < w_end; ++t) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < C_PT; ++q) {
            const int item = tid + q * 256;
            if (item < C_ITEMS) *reinterpret_cast<uint4*>(sC + item * 16) = *reinterpret_cast<const uint4*>(&pc[q]);
        }
#pragma unroll
        for (int q = 0; q < G_PT; ++q) {
            const int item = tid + q * 256, pp = item >> 2, ch = item & 3;
            const int ry = 2 * (pp / (F_TW / 2)), rx = 2 * (pp % (F_TW / 2));
            unsigned d[4];
            if constexpr (GB) {
                d[0] = pg[q][0][0]; d[1] = pg[q][0][1]; d[2] = pg[q][0][2]; d[3] = pg[q][0][3];
            } else {
                d[0] = pk_bf16(__uint_as_float(pg[q][0][0]), __uint_as_float(pg[q][0][1]));
                d[1] = pk_bf16(__uint_as_float(pg[q][0][2]), __uint_as_float(pg[q][0][3]));
                d[2] = pk_bf16(__uint_as_float(pg[q][1][0]), __uint_as_float(pg[q][1][1]));
                d[3] = pk_bf16(__uint_as_float(pg[q][1][2]), __uint_as_float(pg[q][1][3]));
            }
#pragma unroll
            for (int pos = 0; pos < 4; ++pos) {
                const unsigned m0 = eq_bytes(pk[q][0], pos), m1 = eq_bytes(pk[q][1], pos);
                uint4 v;                                               // byte masks -> one 16-bit mask per bf16
                v.x = d[0] & __builtin_amdgcn_perm(m0, m0, 0x01010000u);
                v.y = d[1] & __builtin_amdgcn_perm(m0, m0, 0x03030202u);
                v.z = d[2] & __builtin_amdgcn_perm(m1, m1, 0x01010000u);
                v.w = d[3] & __builtin_amdgcn_perm(m1, m1, 0x03030202u);
                *reinterpret_cast<uint4*>(sZ + ((ry + (pos >> 1)) * F_TW + rx + (pos & 1)) * 64 + ch * 16) = v;
            }
        }
        __syncthreads();
        if (t + 1 < w_end) fetch(t + 1);
#pragma unroll 2
        for (int ks = wave; ks < F_TR * (F_TW / 16); ks += 4) {
            const int r = ks / (F_TW / 16), xr = ks % (F_TW / 16);
            const unsigned char* zb = sZ + (r * F_TW + 16 * xr) * 64 + z_off;
            const bf16x8 b = tr_read8(zb, zb + 4 * 64);
            const unsigned char* ab = sC + (r * F_HC + 16 * xr) * 8;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const bf16x8 a = tr_read8(ab + a_off[f], ab + a_off[f] + 32);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[f], 0, 0, 0);
            }
        }
    }
    // The four waves hold partial sums over different pixel runs: fold them through LDS (waves 2, 3 -> 0, 1, then 1 -> 0; the
    // tiles are dead) so the workgroup writes ONE slab - the slab reduction reads 1024 instead of 4096 slabs (51 -> 15 us).
    float* red = reinterpret_cast<float*>(smem);                     // [2 waves][64 registers][64 lanes] = 32 KB
    static_assert(F_C4_BYTES + F_DZ_BYTES >= 2 * 64 * 64 * 4, "reduction scratch");
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const int senders = round == 0 ? 2 : 1;                      // round 0: waves 2, 3 send; round 1: wave 1 sends
        __syncthreads();
        if (wave >= senders && wave < 2 * senders) {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wave - senders) * 64 + f * 16 + r) * 64 + lane] = acc[f][r];
        }
        __syncthreads();
        if (wave < senders) {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[f][r] += red[(wave * 64 + f * 16 + r) * 64 + lane];
        }
    }
    if (wave != 0) return;
    // D row m = (reg & 3) + 8 (reg >> 2) + 4 kh = 4 (tap - 8 f) + ch, column o = lane & 31
    float* dws = dw_slabs + (long)blockIdx.x * F_DW;
    const int o = lane & 31;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = r & 3, tap = 8 * f + 2 * (r >> 2) + kh;
            if (ch < 3 && tap < 25) dws[(tap * 3 + ch) * 32 + o] = acc[f][r];
            if (ch == 3 && tap == 12 && db_slabs) db_slabs[(long)blockIdx.x * 32 + o] = acc[f][r];
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// Input gradient of the first FAN convolution from the POOLED gradient (throughput mode), 32 -> 3 channels:
//   dc[y][x][i] = sum_{ky,kx,o} dz[y+2-ky][x+2-kx][o] w[ky][kx][i][o]          (dz = MaxPool2D routing of g, as above)
// kx is folded into the MFMA M dimension: P[(kx,i) 15 of 16][q] = sum_{ky,o} w[ky][kx][i][o] dz[y+2-ky][q][o] is five
// v_mfma_f32_16x16x32_bf16 per 16 positions q (K = the 32 channels, contiguous in the [pixel][o] tile: plain 16-byte operand
// reads, XOR-swizzled so that the sixteen lanes of a b128 group cover the 64 banks), the weights (A) live in 20 VGPRs, and
// dc[y][x][i] = sum_kx P[(kx,i)][x+2-kx] is a shift-add through a 4 KB per-wave LDS strip.  The gradient tile is un-pooled
// and arg-max-masked once per tile by the staging pass (packed bit operations), then read five times (once per ky).
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int E_TR = 8, E_TW = 64, E_HR = E_TR + 4, E_HC = E_TW + 4;
constexpr int E_DZ_BYTES = E_HR * E_HC * 64 + 12 * 64;                       // + slack: the last fragment over-reads 12 pixels
constexpr int E_PS = E_HC;                                                   // floats per P row: (4 kg + reg) * 68 = 16 kg + 4 reg
constexpr int E_P_BYTES = 16 * E_PS * 4;                                     // (mod 32 banks) -> both accesses conflict-free

template <bool GB>
__global__ __launch_bounds__(256) void conv1_dgrad_pooled_kernel(const void* __restrict__ gp, const unsigned char* __restrict__ pidx,
                                                                 const float* __restrict__ w, float* __restrict__ dc, int N, int H,
                                                                 int W, int tiles_y, int tiles_x) {
    constexpr unsigned OOB = 0x80000000u;
    __shared__ __attribute__((aligned(16))) unsigned char smem[E_DZ_BYTES + 4 * E_P_BYTES];
    unsigned char* sZ = smem;                                  // [E_HR][E_HC] x 4 chunks of 8 channels (bf16), chunk c of
    const int tid = threadIdx.x, lane = tid & 63;              // tile column q stored at chunk slot c ^ (((q >> 3) & 1) << 1)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* sP = reinterpret_cast<float*>(smem + E_DZ_BYTES + wave * E_P_BYTES);
    const int nl = lane & 15, kg = lane >> 4;
    const int tiles = tiles_y * tiles_x, total = N * tiles;
    const int Hp = H >> 1, Wp = W >> 1;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(gp), 0,
                                                                        (int)((long)N * Hp * Wp * 32 * (GB ? 2 : 4)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(pidx), 0,
                                                                        (int)((long)N * Hp * Wp * 8), 0x00020000);
    // A operand: row m = (kx, i) = nl (row 15 is empty), K values o = 8 kg .. 8 kg + 7, one fragment per ky
    bf16x8 wa[5];
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int e = 0; e < 8; ++e)
            wa[ky][e] = (__bf16)(nl < 15 ? w[((ky * 5 + nl / 3) * 3 + nl % 3) * 32 + 8 * kg + e] : 0.f);

    constexpr int PR = E_HR / 2, PCOLS = E_HC / 2, G_ITEMS = PR * PCOLS * 4, G_PT = (G_ITEMS + 255) / 256;
    u32x4 pg[G_PT][GB ? 1 : 2];
    u32x2 pk[G_PT];
    auto fetch = [&](int t) {
        const int n = t / tiles, tile = t % tiles;
        const int y0 = (tile / tiles_x) * E_TR, x0 = (tile % tiles_x) * E_TW;
#pragma unroll
        for (int q = 0; q < G_PT; ++q) {
            const int item = tid + q * 256, pp = item >> 2, ch = item & 3;
            const int py = ((y0 - 2) >> 1) + pp / PCOLS, px = ((x0 - 2) >> 1) + pp % PCOLS;     // arithmetic shift: -1 at the border
            const bool ok = (item < G_ITEMS) & ((unsigned)py < (unsigned)Hp) & ((unsigned)px < (unsigned)Wp);
            const unsigned e = (unsigned)(((n * Hp + py) * Wp + px) * 32 + ch * 8);
            if constexpr (GB) {
                pg[q][0] = __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? e * 2 : OOB, 0, 0);
            } else {
                pg[q][0] = __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? e * 4 : OOB, 0, 0);
                pg[q][1] = __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? e * 4 + 16 : OOB, 0, 0);
            }
            pk[q] = argmax2_bytes(__builtin_amdgcn_raw_buffer_load_b16(rk, ok ? e >> 2 : OOB, 0, 0));
        }
    };
    const int first = xcd_order(blockIdx.x);
    if (first < total) fetch(first);
    for (int t = first; t < total; t += gridDim.x) {
        const int n = t / tiles, tile = t % tiles;
        const int y0 = (tile / tiles_x) * E_TR, x0 = (tile % tiles_x) * E_TW;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < G_PT; ++q) {
            const int item = tid + q * 256, pp = item >> 2, ch = item & 3;
            if (item < G_ITEMS) {
                const int ry = 2 * (pp / PCOLS), rx = 2 * (pp % PCOLS);
                unsigned d[4];
                if constexpr (GB) {
                    d[0] = pg[q][0][0]; d[1] = pg[q][0][1]; d[2] = pg[q][0][2]; d[3] = pg[q][0][3];
                } else {
                    d[0] = pk_bf16(__uint_as_float(pg[q][0][0]), __uint_as_float(pg[q][0][1]));
                    d[1] = pk_bf16(__uint_as_float(pg[q][0][2]), __uint_as_float(pg[q][0][3]));
                    d[2] = pk_bf16(__uint_as_float(pg[q][1][0]), __uint_as_float(pg[q][1][1]));
                    d[3] = pk_bf16(__uint_as_float(pg[q][1][2]), __uint_as_float(pg[q][1][3]));
                }
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) {
                    const unsigned m0 = eq_bytes(pk[q][0], pos), m1 = eq_bytes(pk[q][1], pos);
                    uint4 v;
                    v.x = d[0] & __builtin_amdgcn_perm(m0, m0, 0x01010000u);
                    v.y = d[1] & __builtin_amdgcn_perm(m0, m0, 0x03030202u);
                    v.z = d[2] & __builtin_amdgcn_perm(m1, m1, 0x01010000u);
                    v.w = d[3] & __builtin_amdgcn_perm(m1, m1, 0x03030202u);
                    const int col = rx + (pos & 1);
                    *reinterpret_cast<uint4*>(sZ + ((ry + (pos >> 1)) * E_HC + col) * 64 + ((ch ^ (((col >> 3) & 1) << 1)) * 16)) = v;
                }
            }
        }
        __syncthreads();
        if (t + (int)gridDim.x < total) fetch(t + gridDim.x);

#ifndef NIMG_E_NO_PAIR
        // a wave owns two ADJACENT output rows: their taps read the tile rows r0 .. r0 + 5, each fragment ONCE for both rows (30
        // operand reads per 50 matrix instructions instead of 50; every accumulator still sums ky = 0 .. 4 in that order: same bits)
        static_assert(E_TR == 8, "four waves x two adjacent rows");
        {
            const int r0 = 2 * wave;
            f32x4 acc2[2][5];
#pragma unroll
            for (int fq = 0; fq < 5; ++fq) {
                const int q = 16 * fq + nl;
                const int slot = (kg ^ (((q >> 3) & 1) << 1)) * 16;
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 6; ++j) {                            // tile row r0 + 5 - j: ky = j for row r0 + 1, j - 1 for row r0
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(sZ + ((r0 + 5 - j) * E_HC + q) * 64 + slot);
                    if (j < 5) a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j], b, a1, 0, 0, 0);
                    if (j >= 1) a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j - 1], b, a0, 0, 0, 0);
                }
                acc2[0][fq] = a0;
                acc2[1][fq] = a1;
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
                for (int fq = 0; fq < 5; ++fq) {
                    const int q = 16 * fq + nl;
                    if (q < E_HC) {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) sP[(4 * kg + reg) * E_PS + q] = acc2[rr][fq][reg];
                    }
                }
                __builtin_amdgcn_wave_barrier();
                const int oy = y0 + r0 + rr;
                float o3[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int kx = 0; kx < 5; ++kx)
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) o3[ci] += sP[(kx * 3 + ci) * E_PS + lane + 4 - kx];
                if (oy < H && x0 + lane < W) {
                    float* d = dc + ((long)(n * H + oy) * W + x0 + lane) * 3;
                    d[0] = o3[0]; d[1] = o3[1]; d[2] = o3[2];
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
#else
        for (int r = wave; r < E_TR; r += 4) {
            // output row y0 + r, tap ky reads gradient row y0 + r + 2 - ky = tile row r + 4 - ky; position q = tile column
#pragma unroll
            for (int fq = 0; fq < 5; ++fq) {
                const int q = 16 * fq + nl;                              // fragment 4 over-reads 12 columns (never gathered)
                const int slot = (kg ^ (((q >> 3) & 1) << 1)) * 16;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ky = 0; ky < 5; ++ky) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(sZ + ((r + 4 - ky) * E_HC + q) * 64 + slot);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ky], b, acc, 0, 0, 0);
                }
                // D: column = position q (lane & 15), rows m = 4 kg + reg -> P[m][q] (m-major: the 16 lanes of a row group write
                // 16 consecutive floats, the two groups of a 32-lane pass sit 16 banks apart)
                if (q < E_HC) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) sP[(4 * kg + reg) * E_PS + q] = acc[reg];
                }
            }
            __builtin_amdgcn_wave_barrier();                             // sP is private to the wave: LDS ops complete in order
            const int oy = y0 + r;
            // lane = output column x0 + lane, all three channels: every read is 64 consecutive floats of one P row
            float o3[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int kx = 0; kx < 5; ++kx)
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) o3[ci] += sP[(kx * 3 + ci) * E_PS + lane + 4 - kx];     // q = x + 2 - kx (+2 halo)
            if (oy < H && x0 + lane < W) {
                float* d = dc + ((long)(n * H + oy) * W + x0 + lane) * 3;
                d[0] = o3[0]; d[1] = o3[1]; d[2] = o3[2];
            }
            __builtin_amdgcn_wave_barrier();
        }
#endif
    }
}

constexpr int F1_WGS = 1024;             // persistent workgroups of the conv1 weight gradient (one slab each)

template <int TWD>
int launch_conv1_pool(const void* c4, const float* w, const float* bias, void* pooled, unsigned char* pidx, int n, int h,
                      int wd, float alpha, int out_bf16, hipStream_t s) {
    const int tiles_y = cdiv(h, 8), tiles_x = cdiv(wd, TWD);
    const long total = (long)n * tiles_y * tiles_x;
    static const int capmul = getenv("NIMG_CONV1_CAP") ? atoi(getenv("NIMG_CONV1_CAP")) : 0;
    const long cap = 256L * (capmul > 0 ? capmul : CONV1_WGS);      // persistent: the resident workgroups (167 registers per lane)
    const dim3 grid((unsigned)(total < cap ? total : cap));
    if (out_bf16)
        hipLaunchKernelGGL((conv1_pool_fwd_kernel<TWD, true>), grid, dim3(256), 0, s, c4, w, bias, pooled, pidx, n, h, wd, alpha,
                           tiles_y, tiles_x);
    else
        hipLaunchKernelGGL((conv1_pool_fwd_kernel<TWD, false>), grid, dim3(256), 0, s, c4, w, bias, pooled, pidx, n, h, wd,
                           alpha, tiles_y, tiles_x);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}


// ------------------------------------------------------------------------------------------------------------------
// The zero-padded 5x5, 3 -> 3 convolution on the matrix core - the main term of the ConstrainedConv2D's input gradient in
// throughput mode (bf16 operands, float32 accumulation, like every other input gradient there; the FORWARD filter stays on the
// float32 stencil above: its -100 centre taps cancel against the rest, which bf16 inputs cannot carry).  cconv_kernel is bound
// by VALU issue (225 FMAs per pixel, 168 us for the 21 M pixels of a step, 2.9 TB/s); here four horizontally adjacent output
// pixels x 3 channels are the N dimension of v_mfma_f32_16x16x32_bf16 (12 of 16 columns), 16 such groups = 64 pixels of an
// image row are M, and K = one window row: 8 input pixels x 4 channels ({c0, c1, c2, 0} bf16 pixels, 8 bytes) - the window of
// group g starts at the even column 4 g - 2, so a lane's 8 K values are ONE aligned 16-byte LDS read.  The weights become five
// banded (Toeplitz) 32 x 16 operands, one per kernel row, built once per wave in registers: five reads + five matrix
// instructions per 64 output pixels; the result tile turns around through a 768-byte LDS row so that every lane stores one
// pixel's 12 bytes.  What remains is the byte stream: 12 B in + 12 B out per pixel.
typedef float f32x4c __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));

template <int ABL>      // diagnostic bits (wrong results): 1 no staging loads, 2 no output stores, 4 no matrix work
__global__ __launch_bounds__(256) void conv5c3_mfma_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                           float* __restrict__ out, int N, int H, int W, int tiles_y) {
    constexpr int TR = 16, TC = 64, HR = TR + 4, HC = TC + 8;            // halo tile: 20 rows x 72 c4 pixels (68 used)
    __shared__ __attribute__((aligned(16))) uint2 tile[HR * HC];
    __shared__ __attribute__((aligned(16))) float orow[4][TC * 3];       // per-wave output row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = xcd_order(blockIdx.x);
    const int tiles_x = W / TC;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y, n = bid / tiles_y;
    const int y0 = ty * TR, x0 = tx * TC;
    // stage the halo tile: float32 NHWC3 -> {c0, c1, c2, 0} bf16 pixels, zeros outside the image.  All loads of a thread are issued
    // before the first is used, from clamped addresses and without branches around them (a guarded load in a rolled loop makes
    // hipcc wait for every element before it requests the next: the 252 MB input then costs 76 us on top of the rest)
    constexpr int ITEMS = HR * (TC + 4), SQ = (ITEMS + 255) / 256;
    float sv[SQ][3];
    bool sok[SQ];
#pragma unroll
    for (int q = 0; q < SQ; ++q) {
        const int i = tid + 256 * q, r = i / (TC + 4), c = i % (TC + 4);
        const int gy = y0 - 2 + r, gx = x0 - 2 + c;
        sok[q] = !(ABL & 1) && i < ITEMS && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const float* src = in + (sok[q] ? (((long)n * H + gy) * W + gx) * 3 : 0L);
        sv[q][0] = src[0]; sv[q][1] = src[1]; sv[q][2] = src[2];
    }
#pragma unroll
    for (int q = 0; q < SQ; ++q) {
        const int i = tid + 256 * q, r = i / (TC + 4), c = i % (TC + 4);
        if (i < ITEMS) {
            const __bf16 b0 = (__bf16)(sok[q] ? sv[q][0] : 0.f), b1 = (__bf16)(sok[q] ? sv[q][1] : 0.f), b2 = (__bf16)(sok[q] ? sv[q][2] : 0.f);
            tile[r * HC + c] = make_uint2((unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16),
                                          (unsigned)__builtin_bit_cast(unsigned short, b2));
        }
    }
    // banded weight operands: lane (col = lane & 15 = 3 j + co, kg = lane >> 4) holds k = 8 kg .. 8 kg + 7 = window columns 2 kg,
    // 2 kg + 1 x 4 channels; output pixel j of a group reads window column wc with tap kx = wc - j.  The 225 weights go through
    // LDS (one coalesced load), the operands are built from there with clamped indices and selects - 40 guarded global loads
    // per lane were 40 dependent round trips in front of every tile
    __shared__ float sw[232];
    if (tid < 225) sw[tid] = w[tid];
    if (tid >= 225 && tid < 232) sw[tid] = 0.f;
    const int col = lane & 15, kg = lane >> 4;
    const int j = col / 3, co = col % 3;
    __syncthreads();
    bf16x8c bw[5];
#pragma unroll
    for (int wr = 0; wr < 5; ++wr)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int wc = 2 * kg + (e >> 2), ci = e & 3, kx = wc - j;
            const bool on = col < 12 && ci < 3 && kx >= 0 && kx < 5;
            const float v = sw[on ? ((wr * 5 + kx) * 3 + ci) * 3 + co : 225];
            bw[wr][e] = (__bf16)v;
        }
    const int g = lane & 15;                                               // A operand: pixel group g, window columns 2 kg, 2 kg + 1
    const uint2* abase = tile + 4 * g + 2 * kg;
#pragma unroll
    for (int q = 0; q < TR / 4; ++q) {
        const int r = wave + 4 * q;                                        // output row of the tile
        f32x4c acc = {0.f, 0.f, 0.f, 0.f};
        if (!(ABL & 4))
#pragma unroll
        for (int wr = 0; wr < 5; ++wr) {
            const uint4 v = *reinterpret_cast<const uint4*>(abase + (r + wr) * HC);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8c*>(&v), bw[wr], acc, 0, 0, 0);
        }
        // D[m = group][n = 3 j + co]: lane holds rows 4 (lane >> 4) + e, column lane & 15 -> float index 12 m + n of the row
        __builtin_amdgcn_wave_barrier();
        if (col < 12) {
#pragma unroll
            for (int e = 0; e < 4; ++e) orow[wave][12 * (4 * kg + e) + col] = acc[e];
        }
        __builtin_amdgcn_wave_barrier();
        const int gy = y0 + r;
        if (!(ABL & 2) && gy < H && lane < 48) // the row's 64 pixels = 768 contiguous, 16-byte aligned bytes: 48 float4 stores
            *reinterpret_cast<float4*>(out + (((long)n * H + gy) * W + x0) * 3 + 4 * lane) =
                *reinterpret_cast<const float4*>(&orow[wave][4 * lane]);
    }
}

}  // namespace

extern "C" {

int nimg_conv1_pool_fwd_c4(const void* c4, const float* w, const float* bias, void* pooled, unsigned char* pool_idx, int n,
                           int h, int wd, float alpha, int out_bf16, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!c4 || !w || !pooled || n < 0 || h < 2 || wd < 2 || (h & 1) || (wd & 1) || !(alpha > 0.f && alpha <= 1.f))
        return NIMG_ERR_ARG;
    const long per_image = (long)h * wd * 8;
    if (per_image > 0x7fffffffL) return NIMG_ERR_ARG;
    const int chunk = (int)(0x7fffffffL / per_image);           // buffer descriptors address < 2 GB
    const size_t esz = out_bf16 ? 2 : 4;
    for (int n0 = 0; n0 < n; n0 += chunk) {
        const int nn = n - n0 < chunk ? n - n0 : chunk;
        const long px0 = (long)n0 * h * wd, pp0 = (long)n0 * (h / 2) * (wd / 2) * 32;
        const void* src = (const char*)c4 + px0 * 8;
        void* dst = (char*)pooled + pp0 * esz;
        unsigned char* di = pool_idx ? pool_idx + (pp0 >> 2) : nullptr;       // 2 bits per value
        const int rc = wd > 96 ? launch_conv1_pool<256>(src, w, bias, dst, di, nn, h, wd, alpha, out_bf16, (hipStream_t)stream)
                               : launch_conv1_pool<64>(src, w, bias, dst, di, nn, h, wd, alpha, out_bf16, (hipStream_t)stream);
        if (rc != NIMG_OK) return rc;
    }
    return NIMG_OK;
}

int nimg_conv1_dgrad_pooled(const void* g, const unsigned char* pool_idx, const float* w, float* dc, int n, int h, int wd,
                            int g_bf16, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!g || !pool_idx || !w || !dc || n < 0 || h < 2 || wd < 2 || (h & 1) || (wd & 1)) return NIMG_ERR_ARG;
    if ((long)n * (h / 2) * (wd / 2) * 32 * (g_bf16 ? 2 : 4) > 0x7fffffffL) return NIMG_ERR_ARG;      // one descriptor, < 2 GB
    const int tiles_y = cdiv(h, E_TR), tiles_x = cdiv(wd, E_TW);
    const long total = (long)n * tiles_y * tiles_x;
    const dim3 grid((unsigned)(total < 512 ? total : 512));              // persistent: two 70 KB workgroups per CU
    if (g_bf16)
        hipLaunchKernelGGL((conv1_dgrad_pooled_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, g, pool_idx, w, dc, n, h, wd,
                           tiles_y, tiles_x);
    else
        hipLaunchKernelGGL((conv1_dgrad_pooled_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, g, pool_idx, w, dc, n, h,
                           wd, tiles_y, tiles_x);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

size_t nimg_conv1_wgrad_c4_workspace_bytes(void) { return (size_t)F1_WGS * (F_DW + 32) * sizeof(float); }

int nimg_conv1_wgrad_c4(const void* c4, const void* g, const unsigned char* pool_idx, float* dw, float* db, int n, int h, int wd,
                        int g_bf16, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!c4 || !g || !pool_idx || !dw || !workspace || n < 0 || h < 2 || wd < 2 || (h & 1) || (wd & 1)) return NIMG_ERR_ARG;
    if (workspace_bytes < nimg_conv1_wgrad_c4_workspace_bytes()) return NIMG_ERR_WORKSPACE;
    if ((long)n * h * wd * 8 > 0x7fffffffL) return NIMG_ERR_ARG;          // one buffer descriptor per tensor (< 2 GB)
    const int tiles_y = cdiv(h, F_TR), tiles_x = cdiv(wd, F_TW);
    const long total = (long)n * tiles_y * tiles_x;
    const int wgs = (int)(total < F1_WGS ? total : F1_WGS), per = cdiv(total, wgs);
    const int used = cdiv(total, per);                                    // workgroups that own at least one tile
    float* dws = (float*)workspace;
    float* dbs = dws + (size_t)F1_WGS * F_DW;
    if (g_bf16)
        hipLaunchKernelGGL((conv1_wgrad_pooled_kernel<true>), dim3(used), dim3(256), 0, (hipStream_t)stream, c4, g, pool_idx, dws,
                           db ? dbs : nullptr, n, h, wd, tiles_y, tiles_x, per);
    else
        hipLaunchKernelGGL((conv1_wgrad_pooled_kernel<false>), dim3(used), dim3(256), 0, (hipStream_t)stream, c4, g, pool_idx,
                           dws, db ? dbs : nullptr, n, h, wd, tiles_y, tiles_x, per);
    NIMG_CHECK_LAUNCH();
    launch_reduce2(dws, dw, F_DW, used, db ? dbs : nullptr, db, 32, used, accumulate, (hipStream_t)stream);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_cconv3(const float* in, const float* w, float* out_f32, void* out_c4, int n, int h, int wd, int pad_mode,
                void* stream) {
    if (n == 0) return NIMG_OK;
    if (!in || !w || (!out_f32 && !out_c4) || n < 0 || h < 1 || wd < 1 || pad_mode < 0 || pad_mode > 2) return NIMG_ERR_ARG;
    if (pad_mode != 0 && (h < 2 || wd < 2)) return NIMG_ERR_ARG;
    const long per_image = (long)h * wd * 12;
    if (per_image > 0x7fffffffL) return NIMG_ERR_ARG;
    const int chunk = (int)(0x7fffffffL / per_image);           // buffer descriptors address < 2 GB
    for (int n0 = 0; n0 < n; n0 += chunk) {
        CConvParams p;
        const long px0 = (long)n0 * h * wd;
        p.in = in + px0 * 3; p.w = w; p.out_f32 = out_f32 ? out_f32 + px0 * 3 : nullptr;
        p.out_c4 = out_c4 ? (uint2*)out_c4 + px0 : nullptr;
        p.N = n - n0 < chunk ? n - n0 : chunk; p.H = h; p.W = wd; p.pad_mode = pad_mode; p.tiles_y = p.tiles_x = 0;
        static const int variant = getenv("NIMG_CCONV_VARIANT") ? atoi(getenv("NIMG_CCONV_VARIANT")) : 0;   // timing experiments
        const int rc = wd <= 96 ? launch_cconv<16, 1>(p, (hipStream_t)stream)
                     : variant == 1 ? launch_cconv<64, 1>(p, (hipStream_t)stream)
                     : variant == 2 ? launch_cconv<64, 4>(p, (hipStream_t)stream)
                     : variant == 3 ? launch_cconv<32, 2>(p, (hipStream_t)stream)
                                    : launch_cconv<64, 2>(p, (hipStream_t)stream);
        if (rc != NIMG_OK) return rc;
    }
    return NIMG_OK;
}

int nimg_conv5c3_bf16(const float* in, const float* w, float* out, int n, int h, int wd, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!in || !w || !out || n < 0 || h < 1 || wd < 64 || (wd & 63)) return NIMG_ERR_ARG;
    const int tiles_y = (h + 15) / 16;
    const long blocks = (long)n * tiles_y * (wd / 64);
    if (blocks > 0x7fffffffL) return NIMG_ERR_ARG;
    static const int abl = getenv("NIMG_C5C3_ABL") ? atoi(getenv("NIMG_C5C3_ABL")) : 0;
    auto kern = abl == 1 ? conv5c3_mfma_kernel<1> : abl == 2 ? conv5c3_mfma_kernel<2> : abl == 4 ? conv5c3_mfma_kernel<4>
              : abl == 3 ? conv5c3_mfma_kernel<3> : conv5c3_mfma_kernel<0>;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, w, out, n, h, wd, tiles_y);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_cconv3_dgrad_border(const float* dc, const float* nf, float* dx, int n, int h, int wd, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!dc || !nf || !dx || n < 0 || h < 4 || wd < 4) return NIMG_ERR_ARG;
    const long total = (long)n * (4 * wd + 4 * (h - 4));
    const long g = (total + 255) / 256;
    hipLaunchKernelGGL(cdgrad_border_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream, dc, nf,
                       dx, n, h, wd);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

}  // extern "C"
