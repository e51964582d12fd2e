// Convolutions with very few OUTPUT channels (Cout <= 4) on the vector ALU, float32:
//   ConstrainedConv2D forward 5x5 3->3 with SYMMETRIC padding          (models/layers.py:56-57)
//   its input gradient on the padded domain 3->3                        (backward of the same)
//   the input gradient of the FAN's first convolution, 5x5 32->3        (models/forensics.py:69, backward)
// A 32-wide MFMA N tile would idle >= 90 % of the matrix core here, and in float32 the VALU and MFMA peaks are the same
// 157 TFLOP/s, so these run as a register-blocked direct convolution: one thread owns PW = 4 horizontally adjacent output
// pixels x all Cout channels, every input sample fetched from the LDS halo tile feeds up to KS x Cout FMAs, and the
// weights - indexed only by loop counters - live in SGPRs (scalar loads), so the inner loop is pure v_fmac.
#include "common.h"

namespace {

using namespace nimg;

struct SmallParams {
    const float* in;
    const float* w;       // [KS*KS][Cin][COUT]
    const float* bias;    // optional
    float* out;
    int N, H, W, Cin, Hout, Wout, pad_t, pad_l, pad_mode;
    int tiles_y, tiles_x;
};

constexpr int S_TH = 16, S_TWT = 16, S_PW = 4;      // 16 rows x (16 threads x 4 px) = 16 x 64 output tile
constexpr int S_TW = S_TWT * S_PW;

template <int KS, int COUT, int CK>
__global__ __launch_bounds__(256) void conv_fewout_kernel(const SmallParams p) {
    constexpr int THH = S_TH + KS - 1, TWH = S_TW + KS - 1;
    constexpr int CKP = CK + (CK % 2 == 0 ? 1 : 0);            // odd pixel stride => conflict-free column walks
    extern __shared__ __attribute__((aligned(16))) float smem[];    // [THH][TWH][CKP]
    const int tid = threadIdx.x;
    const int tx = tid % S_TWT, ty = tid / S_TWT;
    int bid = xcd_order(blockIdx.x);
    const int tiles = p.tiles_y * p.tiles_x;
    const int n = bid / tiles, tile = bid % tiles;
    const int ty0 = (tile / p.tiles_x) * S_TH, tx0 = (tile % p.tiles_x) * S_TW;

    float acc[S_PW][COUT];
#pragma unroll
    for (int j = 0; j < S_PW; ++j)
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[j][o] = 0.f;

    // Many-channel case (CK = 8: the FAN conv1 input gradient in float32 mode): the weights of the current channel chunk
    // sit in LDS and one kernel row of them at a time in VGPRs (wave-uniform addresses = broadcast reads) instead of being
    // wave-uniform scalar loads in the FMA loop: +1.9 % float32-mode step.  For the 3-channel filter (225 weights) the same
    // change is neutral stand-alone and costs 1.7 % of the bf16-mode step (more LDS per workgroup next to the kernels it
    // overlaps with), so that case keeps the scalar loads (three same-box A/B pairs each).
    constexpr bool WLDS = (CK != 3);
    constexpr int WROW = KS * CK * COUT, WROWP = (WROW + 3) / 4 * 4;     // weights of one kernel row and channel chunk
    float* wl = smem + (THH * TWH * CKP + 3) / 4 * 4;                   // [KS][kx][k][o] padded to WROWP, behind the halo tile

    auto stage_weights = [&](int c0) {                      // this chunk's weights: w[(ky*KS + kx)*Cin + c0 + k][o]
        for (int i = tid; i < KS * WROWP; i += 256) {
            const int ky = i / WROWP, r = i % WROWP;
            const int o = r % COUT, k = (r / COUT) % CK, kx = r / (COUT * CK);
            wl[i] = (r < WROW && c0 + k < p.Cin) ? p.w[((long)(ky * KS + kx) * p.Cin + c0 + k) * COUT + o] : 0.f;
        }
    };
    for (int c0 = 0; c0 < p.Cin; c0 += CK) {
        __syncthreads();
        if constexpr (WLDS) stage_weights(c0);
        // staged in batches of 8 independent loads per thread (a rolled loop would serialise 16 global round trips)
        constexpr int NITEM = THH * TWH * CK, NIT = (NITEM + 255) / 256;
#pragma unroll 8
        for (int it = 0; it < NIT; ++it) {
            const int item = tid + it * 256;
            const int k = item % CK, pix = item / CK;
            int gy = ty0 - p.pad_t + pix / TWH, gx = tx0 - p.pad_l + pix % TWH;
            float v = 0.f;
            if (item < NITEM && c0 + k < p.Cin && map_coord(gy, p.H, p.pad_mode) && map_coord(gx, p.W, p.pad_mode))
                v = p.in[(((long)n * p.H + gy) * p.W + gx) * p.Cin + c0 + k];
            if (item < NITEM) smem[pix * CKP + k] = v;
        }
        __syncthreads();
        const int ck = min(CK, p.Cin - c0);
        // fully unrolled for the 3-channel case: the weights are wave-uniform scalar loads, and only an unrolled body lets
        // the compiler issue the s_loads of the next (ky, ci) pairs while the current FMAs run
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const float* row = smem + ((ty + ky) * TWH + tx * S_PW) * CKP;
            float wrow[WLDS ? WROWP : 4];                   // this kernel row's weights: wave-uniform LDS reads (broadcast)
            if constexpr (WLDS) {
#pragma unroll
                for (int i = 0; i < WROWP / 4; ++i) {
                    const float4 w4 = *reinterpret_cast<const float4*>(wl + ky * WROWP + 4 * i);
                    wrow[4 * i] = w4.x; wrow[4 * i + 1] = w4.y; wrow[4 * i + 2] = w4.z; wrow[4 * i + 3] = w4.w;
                }
            }
#pragma unroll
            for (int k = 0; k < CK; ++k) {
                if (k >= ck) break;
                const float* wk = p.w + ((long)(ky * KS) * p.Cin + c0 + k) * COUT;     // + kx * Cin * COUT
#pragma unroll
                for (int c = 0; c < S_PW + KS - 1; ++c) {
                    const float v = row[c * CKP + k];
#pragma unroll
                    for (int j = 0; j < S_PW; ++j) {
                        const int kx = c - j;
                        if (kx >= 0 && kx < KS) {
#pragma unroll
                            for (int o = 0; o < COUT; ++o) {
                                if constexpr (WLDS) acc[j][o] = fmaf(v, wrow[(kx * CK + k) * COUT + o], acc[j][o]);
                                else acc[j][o] = fmaf(v, wk[(long)kx * p.Cin * COUT + o], acc[j][o]);
                            }
                        }
                    }
                }
            }
        }
    }
    const int oy = ty0 + ty;
    if (oy >= p.Hout) return;
#pragma unroll
    for (int j = 0; j < S_PW; ++j) {
        const int ox = tx0 + tx * S_PW + j;
        if (ox >= p.Wout) continue;
        float* o = p.out + (((long)n * p.Hout + oy) * p.Wout + ox) * COUT;
#pragma unroll
        for (int c = 0; c < COUT; ++c) o[c] = acc[j][c] + (p.bias ? p.bias[c] : 0.f);
    }
}

template <int KS, int COUT, int CK>
int launch_fewout(SmallParams p, hipStream_t s) {
    constexpr int THH = S_TH + KS - 1, TWH = S_TW + KS - 1;
    constexpr int CKP = CK + (CK % 2 == 0 ? 1 : 0);
    constexpr size_t lds = ((size_t)THH * TWH * CKP + 4 + (CK != 3 ? KS * ((KS * CK * COUT + 3) / 4 * 4) : 0)) * sizeof(float);
    p.tiles_y = cdiv(p.Hout, S_TH);
    p.tiles_x = cdiv(p.Wout, S_TW);
    const long blocks = (long)p.N * p.tiles_y * p.tiles_x;
    auto k = conv_fewout_kernel<KS, COUT, CK>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), lds, s, p);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

}  // namespace

// internal entry used by nimg_conv2d_fwd's dispatcher (conv_mfma.hip); not part of the public ABI
int nimg_internal_conv_fewout(const float* in, int cin, const float* w, const float* bias, float* out, int cout, int n,
                              int h, int wd, int ks, int pad_t, int pad_l, int pad_mode, int hout, int wout,
                              hipStream_t s) {
    SmallParams p;
    p.in = in; p.w = w; p.bias = bias; p.out = out; p.N = n; p.H = h; p.W = wd; p.Cin = cin; p.Hout = hout; p.Wout = wout;
    p.pad_t = pad_t; p.pad_l = pad_l; p.pad_mode = pad_mode; p.tiles_y = p.tiles_x = 0;
    if (cout == 3 && ks == 5) return cin <= 4 ? launch_fewout<5, 3, 3>(p, s) : launch_fewout<5, 3, 8>(p, s);
    if (cout == 3 && ks == 3) return cin <= 4 ? launch_fewout<3, 3, 3>(p, s) : launch_fewout<3, 3, 8>(p, s);
    return NIMG_ERR_ARG;
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of a tiny convolution (Cin <= 4 AND Cout <= 4: the ConstrainedConv2D 5x5x3x3 filter, 225 numbers):
// one thread per (tap, ci, co) walks the pixels of a 32x32 tile held in LDS (both operands are broadcast / conflict-free
// reads), persistent workgroups loop over tiles and write one partial per workgroup; a fixed-order reduction follows.
// HBM-bound on paper (reads x and dz once, 2 x 0.25 GB for 320 images), exact float32 in both compute modes.
namespace {

struct TinyWParams {
    const float* in;
    const float* dz;
    float* partial;      // [gridDim.x][taps*cin*cout]
    int N, H, W, pad, pad_mode, tiles_y, tiles_x;
};

template <int KS, int CIN, int COUT>
__global__ __launch_bounds__(256) void conv_wgrad_tiny_kernel(const TinyWParams p) {
    constexpr int T = 32, TH = T + KS - 1, NOUT = KS * KS * CIN * COUT;
    static_assert(NOUT <= 256, "one thread per weight");
    __shared__ float sx[TH * TH * CIN];
    __shared__ float sd[T * T * COUT];
    const int tid = threadIdx.x;
    const int co = tid % COUT, ci = (tid / COUT) % CIN, tap = tid / (COUT * CIN);
    const int ky = tap / KS, kx = tap % KS;
    const long total = (long)p.N * p.tiles_y * p.tiles_x;
    float acc = 0.f;
    for (long t = xcd_order(blockIdx.x); t < total; t += gridDim.x) {
        const int n = (int)(t / (p.tiles_y * p.tiles_x)), tile = (int)(t % (p.tiles_y * p.tiles_x));
        const int y0 = (tile / p.tiles_x) * T, x0 = (tile % p.tiles_x) * T;
        __syncthreads();
        for (int i = tid; i < TH * TH; i += 256) {
            int gy = y0 - p.pad + i / TH, gx = x0 - p.pad + i % TH;
            const bool ok = map_coord(gy, p.H, p.pad_mode) && map_coord(gx, p.W, p.pad_mode);
            const float* src = p.in + (((long)n * p.H + gy) * p.W + gx) * CIN;
#pragma unroll
            for (int c = 0; c < CIN; ++c) sx[i * CIN + c] = ok ? src[c] : 0.f;
        }
        for (int i = tid; i < T * T; i += 256) {
            const int gy = y0 + i / T, gx = x0 + i % T;
            const bool ok = gy < p.H && gx < p.W;
            const float* src = p.dz + (((long)n * p.H + gy) * p.W + gx) * COUT;
#pragma unroll
            for (int c = 0; c < COUT; ++c) sd[i * COUT + c] = ok ? src[c] : 0.f;
        }
        __syncthreads();
        if (tid < NOUT) {
            for (int y = 0; y < T; ++y) {
                const float* xr = sx + ((y + ky) * TH + kx) * CIN + ci;
                const float* dr = sd + y * T * COUT + co;
#pragma unroll 8
                for (int x = 0; x < T; ++x) acc = fmaf(xr[x * CIN], dr[x * COUT], acc);
            }
        }
    }
    if (tid < NOUT) p.partial[(long)blockIdx.x * NOUT + tid] = acc;
}

// The ConstrainedConv2D filter itself (5x5, 3 -> 3, same-size output): the one-thread-per-weight walk above issues two
// LDS reads per FMA.  Here a thread owns one kernel ROW ky and one output row of a 48 x 32-pixel tile and slides along
// it: each step loads ONE new input pixel and ONE dz pixel (16-byte LDS reads, row strides padded to 37 / 33 float4 so
// that 16 lanes hit 64 distinct banks) and feeds 5 taps x 3 x 3 = 45 FMAs from registers.  240 of 256 threads are
// busy; the next tile is fetched into registers while the current one is processed; the 48 row-partials of every
// weight are summed through LDS once per workgroup, in a fixed order.
__global__ __launch_bounds__(256, 2) void conv_wgrad_c3k5_kernel(const TinyWParams p) {
    constexpr int KS = 5, TR = 48, TC = 32, XR = TR + KS - 1, XC = TC + KS - 1, XS = 37, DS = 33, NOUT = 225;
    constexpr int NXP = (XR * XC + 255) / 256, NDP = (TR * TC + 255) / 256;
    constexpr int SX = XR * XS, SD = TR * DS;                  // float4 entries
    static_assert((SX + SD) * 4 >= 240 * 45, "reduction scratch must fit the tiles");
    __shared__ float4 smem4[SX + SD];
    float4* sx = smem4;
    float4* sd = smem4 + SX;
    const int tid = threadIdx.x;
    const int ky = tid / TR, row = tid % TR;
    const bool active = tid < KS * TR;
    const long total = (long)p.N * p.tiles_y * p.tiles_x;
    // 45 FMAs per tile column as 25 instructions (the VALU retires v_pk_fma_f32 at the rate of v_fma_f32): the outputs (co0, co1)
    // of a (kx, ci) pair up against the gradient pair {d.x, d.y} with the input value broadcast (op_sel), output co2 of
    // (ci0, ci1) pairs up against the input pair {w.x, w.y} with d.z broadcast, (ci2, co2) stays scalar.  Same products in the
    // same order per accumulator as the scalar form.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 a01[KS][3], a2p[KS];
    float a22[KS];
#pragma unroll
    for (int a = 0; a < KS; ++a) {
        a01[a][0] = a01[a][1] = a01[a][2] = a2p[a] = f32x2{0.f, 0.f};
        a22[a] = 0.f;
    }
    float px[NXP][3], pd[NDP][3];
    auto fetch = [&](long t) {
        const int n = (int)(t / (p.tiles_y * p.tiles_x)), tile = (int)(t % (p.tiles_y * p.tiles_x));
        const int y0 = (tile / p.tiles_x) * TR, x0 = (tile % p.tiles_x) * TC;
#pragma unroll
        for (int q = 0; q < NXP; ++q) {
            const int i = tid + q * 256;
            int gy = y0 - p.pad + i / XC, gx = x0 - p.pad + i % XC;
            const bool ok = i < XR * XC && map_coord(gy, p.H, p.pad_mode) && map_coord(gx, p.W, p.pad_mode);
            const float* src = p.in + (((long)n * p.H + (ok ? gy : 0)) * p.W + (ok ? gx : 0)) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) px[q][c] = ok ? src[c] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < NDP; ++q) {
            const int i = tid + q * 256;
            const int gy = y0 + i / TC, gx = x0 + i % TC;
            const bool ok = i < TR * TC && gy < p.H && gx < p.W;
            const float* src = p.dz + (((long)n * p.H + (ok ? gy : 0)) * p.W + (ok ? gx : 0)) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) pd[q][c] = ok ? src[c] : 0.f;
        }
    };
    const long first_tile = xcd_order(blockIdx.x);
    if (first_tile < total) fetch(first_tile);
    for (long t = first_tile; t < total; t += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NXP; ++q) {
            const int i = tid + q * 256;
            if (i < XR * XC) sx[(i / XC) * XS + i % XC] = make_float4(px[q][0], px[q][1], px[q][2], 0.f);
        }
#pragma unroll
        for (int q = 0; q < NDP; ++q) {
            const int i = tid + q * 256;
            if (i < TR * TC) sd[(i / TC) * DS + i % TC] = make_float4(pd[q][0], pd[q][1], pd[q][2], 0.f);
        }
        __syncthreads();
        if (t + gridDim.x < total) fetch(t + gridDim.x);
        if (active) {
            const float4* xr = sx + (row + ky) * XS;
            const float4* dr = sd + row * DS;
            float4 w[KS];
#pragma unroll
            for (int k = 0; k < KS - 1; ++k) w[k + 1] = xr[k];
#pragma unroll 4
            for (int c = 0; c < TC; ++c) {
#pragma unroll
                for (int k = 0; k < KS - 1; ++k) w[k] = w[k + 1];
                w[KS - 1] = xr[c + KS - 1];
                const float4 d = dr[c];
                const f32x2 d01 = {d.x, d.y}, dzz = {d.z, d.z};
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    a01[kx][0] = __builtin_elementwise_fma(f32x2{w[kx].x, w[kx].x}, d01, a01[kx][0]);
                    a01[kx][1] = __builtin_elementwise_fma(f32x2{w[kx].y, w[kx].y}, d01, a01[kx][1]);
                    a01[kx][2] = __builtin_elementwise_fma(f32x2{w[kx].z, w[kx].z}, d01, a01[kx][2]);
                    a2p[kx] = __builtin_elementwise_fma(f32x2{w[kx].x, w[kx].y}, dzz, a2p[kx]);
                    a22[kx] = fmaf(w[kx].z, d.z, a22[kx]);
                }
            }
        }
    }
    float acc[KS][3][3];
#pragma unroll
    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            acc[kx][ci][0] = a01[kx][ci][0];
            acc[kx][ci][1] = a01[kx][ci][1];
            acc[kx][ci][2] = ci < 2 ? a2p[kx][ci] : a22[kx];
        }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem4);              // [45][240]
    if (active) {
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                for (int co = 0; co < 3; ++co) red[((kx * 3 + ci) * 3 + co) * (KS * TR) + tid] = acc[kx][ci][co];
    }
    __syncthreads();
    if (tid < NOUT) {
        const int kyo = tid / 45, rest = tid % 45;               // dw index = ((ky*5 + kx)*3 + ci)*3 + co = ky*45 + rest
        const float* src = red + rest * (KS * TR) + kyo * TR;
        float s = 0.f;
        for (int r = 0; r < TR; ++r) s += src[r];
        p.partial[(long)blockIdx.x * NOUT + tid] = s;
    }
}

// The same weight gradient on the matrix core (throughput mode).  The vector kernel above is VALU-bound: 225 FMAs per pixel,
// 230 us for the 21 M pixels of a training step while its 504 MB would stream in ~100.  bf16 operands are what every other weight
// gradient of the throughput mode multiplies (float32 accumulation; nothing here cancels like the filter's forward pass does).
// All 225 weights are ONE 16 x 16 accumulator tile:
//     D[m = (kx, ci)][n = (ky, co)] = sum over p = (v, x) of  A[m][p] B[p][n],
//     A[(kx, ci)][(v, x)] = xp[v][x + kx][ci],   B[(v, x)][(ky, co)] = dz[v - ky][x][co]      (v = y + ky: padded row, 0 .. H + 3)
// i.e. the kernel-row shift sits in the gradient operand (whole rows of a planar tile with a 4-row zero apron: aligned) and the
// kernel-column shift in the input operand, which is staged as FIVE pre-shifted planar copies (a shift by one bf16 pixel would
// otherwise misalign the 16-byte fragment reads): v_mfma_f32_16x16x32_bf16 with K = 32 consecutive pixels of a row, 15 x 15 of
// the 16 x 16 tile used, one instruction per 32 pixels.  What is left is staging: float32 NHWC3 -> bf16 planes, two pixels per
// lane packed into dwords (odd shifts take the neighbour lane's pixel by a wave shuffle).  A workgroup walks tiles of TR padded
// rows x 64 columns; the next tile's loads are in flight during the current tile's staging + matrix work.
typedef float f32x4w __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const __bf16 l = (__bf16)lo, h = (__bf16)hi;
    return (unsigned)__builtin_bit_cast(unsigned short, l) | ((unsigned)__builtin_bit_cast(unsigned short, h) << 16);
}

template <int TR>
__global__ __launch_bounds__(256, 2) void conv_wgrad_c3k5_mfma_kernel(const TinyWParams p) {
    constexpr int TC = 64, DR = TR + 4;                           // tile: TR padded rows x 64 columns; gradient rows incl. apron
    // bytes: one row / one (kx, ci) plane of the shifted input copies.  A row carries 16 B in front of and behind its 64 pixels:
    // the shifted copies of the 34 staged pairs land on dwords -2 .. 33 of a row - with the margin every store is in bounds
    // and needs no predicate of its own (the margins are never read)
    constexpr int XMARGIN = 16, XROW = TC * 2 + 2 * XMARGIN, XPLANE = TR * XROW;
    constexpr int ZROW = TC * 2, ZPLANE = DR * ZROW;              // gradient planes (co)
    constexpr int XBYTES = 15 * XPLANE, ZBYTES = 3 * ZPLANE;
    constexpr int XQ = TR / 4, ZQ = (DR * 32 + 255) / 256, SPW = TR * 2 / 4;       // staging passes, matrix steps per wave
    __shared__ __attribute__((aligned(16))) unsigned char smem[XBYTES + ZBYTES > 4096 ? XBYTES + ZBYTES : 4096];
    unsigned char* sx = smem;
    unsigned char* sz = smem + XBYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rows_p = p.H + 4;                                   // padded rows v
    const int tiles_v = (rows_p + TR - 1) / TR, tiles_x = p.W / TC;
    const long total = (long)p.N * tiles_v * tiles_x;

    // staging roles.  Input: wave w stages padded rows w, w + 4, ... of the tile; lane j < 34 holds the pixel pair (2 j, 2 j + 1)
    // of the 68 padded columns the five shifts need.  Gradient: DR rows x 32 pairs, item = tid + 256 q.
    float px[XQ][6], pz[ZQ][6];
    auto fetch = [&](long t) {
        const int n = (int)(t / (tiles_v * tiles_x)), tile = (int)(t % (tiles_v * tiles_x));
        const int v0 = (tile / tiles_x) * TR, x0 = (tile % tiles_x) * TC;
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int v = v0 + wave + 4 * q;
            int gy = v - 2;
            const bool oky = v < rows_p && lane < 34 && map_coord(gy, p.H, p.pad_mode);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int gx = x0 + 2 * lane + e - 2;
                const bool ok = oky && map_coord(gx, p.W, p.pad_mode);
                // (clamped address, unconditional loads, select afterwards: a guarded load makes hipcc wait per element)
                const float* src = p.in + (ok ? (((long)n * p.H + gy) * p.W + gx) * 3 : 0L);
                const float l0 = src[0], l1 = src[1], l2 = src[2];
                px[q][3 * e] = ok ? l0 : 0.f; px[q][3 * e + 1] = ok ? l1 : 0.f; px[q][3 * e + 2] = ok ? l2 : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < ZQ; ++q) {
            const int item = tid + 256 * q, r = item >> 5, j = item & 31;
            const int gy = v0 - 4 + r, gx = x0 + 2 * j;
            const bool ok = item < DR * 32 && (unsigned)gy < (unsigned)p.H;
            const float* src = p.dz + (ok ? (((long)n * p.H + gy) * p.W + gx) * 3 : 0L);      // the pair is 24 contiguous bytes
            float l[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) l[c] = src[c];
#pragma unroll
            for (int c = 0; c < 6; ++c) pz[q][c] = ok ? l[c] : 0.f;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int vr = wave + 4 * q;
            unsigned e[3], o[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                e[c] = pack_bf16x2(px[q][c], px[q][3 + c]);                           // padded columns 2 j, 2 j + 1
                const unsigned nx = (unsigned)__shfl_down((int)e[c], 1, 64);          // the next pair (lane 33's is never used)
                o[c] = (e[c] >> 16) | (nx << 16);                                      // padded columns 2 j + 1, 2 j + 2
            }
            // copy kx holds xp[..][x + kx]: tile column x = padded column - kx.  Even kx: pair j lands on dword j - kx / 2;
            // odd kx: the (odd, even) pair starting at padded column 2 j + 1 lands on dword j - (kx - 1) / 2
            if (lane < 34) {
                unsigned char* row = sx + vr * XROW + XMARGIN + lane * 4;
#pragma unroll
                for (int kx = 0; kx < 5; ++kx)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        *reinterpret_cast<unsigned*>(row + (kx * 3 + c) * XPLANE - (kx >> 1) * 4) = (kx & 1) ? o[c] : e[c];
            }
        }
#pragma unroll
        for (int q = 0; q < ZQ; ++q) {
            const int item = tid + 256 * q, r = item >> 5, j = item & 31;
            if (item < DR * 32) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    *reinterpret_cast<unsigned*>(sz + c * ZPLANE + r * ZROW + j * 4) = pack_bf16x2(pz[q][c], pz[q][3 + c]);
            }
        }
    };
    // fragment addresses: lane (i = lane & 15, kg = lane >> 4) reads the 8 pixels 8 kg .. 8 kg + 7 of a 32-pixel row segment of
    // plane i (A: i = 3 kx + ci; B: i = 3 ky + co, i.e. gradient plane co at row vr - ky + 4).  i = 15 is the unused row / column
    // of the tile: it reads plane 14's data, its products land in accumulator entries nobody stores.
    const int fi = (lane & 15) < 15 ? (lane & 15) : 14, kg = lane >> 4;
    const int a_lane = fi * XPLANE + XMARGIN + kg * 16;
    const int b_lane = (fi % 3) * ZPLANE + (4 - fi / 3) * ZROW + kg * 16;
    f32x4w acc = {0.f, 0.f, 0.f, 0.f};
    const long first = xcd_order(blockIdx.x);
    if (first < total) fetch(first);
    for (long t = first; t < total; t += gridDim.x) {
        __syncthreads();                                   // the previous tile's fragment reads are done
        commit();
        __syncthreads();
        if (t + gridDim.x < total) fetch(t + gridDim.x);
        const int v0 = (int)((t % (tiles_v * tiles_x)) / tiles_x) * TR;
#pragma unroll
        for (int s8 = 0; s8 < SPW; ++s8) {                 // TR rows x 2 segments of 32 pixels, split over the four waves
            const int step = wave * SPW + s8, vr = step >> 1, seg = step & 1;
            if (v0 + vr < rows_p) {                        // wave-uniform: the last row tile of an image is partial
                const bf16x8w a = *reinterpret_cast<const bf16x8w*>(sx + a_lane + vr * XROW + seg * 64);
                const bf16x8w b = *reinterpret_cast<const bf16x8w*>(sz + b_lane + vr * ZROW + seg * 64);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
            }
        }
    }
    // D[m][n]: lane holds rows m = 4 (lane >> 4) + r, column n = lane & 15.  The four waves' tiles are added in a fixed order.
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);           // [4 waves][16 m][16 n]
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
    __syncthreads();
    if (tid < 225) {
        const int ky = tid / 45, rest = tid % 45, kx = rest / 9, ci = (rest / 3) % 3, co = rest % 3;
        const int m = kx * 3 + ci, n = ky * 3 + co;
        p.partial[(long)blockIdx.x * 225 + tid] = (red[m * 16 + n] + red[256 + m * 16 + n]) + (red[512 + m * 16 + n] + red[768 + m * 16 + n]);
    }
}

// one wave per weight: lane l adds the partials l, l+64, ... in order, then a fixed-shape butterfly => deterministic
__global__ __launch_bounds__(64) void tiny_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                         int count, int blocks, int accumulate) {
    const int i = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int b = lane; b < blocks; b += 64) s += partial[(long)b * count + i];
    s = wave_sum(s);
    if (lane == 0) dw[i] = accumulate ? dw[i] + s : s;
}

}  // namespace

constexpr int TINY_BLOCKS = 1024;

size_t nimg_internal_wgrad_tiny_bytes(int ks, int cin, int cout) { return (size_t)2 * TINY_BLOCKS * ks * ks * cin * cout * sizeof(float); }

// internal entry used by the weight-gradient dispatchers; same-size output (stride 1, pad = (ks-1)/2), any pad mode
int nimg_internal_conv_wgrad_tiny(const float* in, const float* dz, float* dw, int cin, int cout, int n, int h, int wd,
                                  int ks, int pad, int pad_mode, int accumulate, void* workspace, hipStream_t s, bool bf16_ok) {
    TinyWParams p;
    p.in = in; p.dz = dz; p.partial = (float*)workspace; p.N = n; p.H = h; p.W = wd; p.pad = pad; p.pad_mode = pad_mode;
    const bool c3k5 = ks == 5 && cin == 3 && cout == 3;
    p.tiles_y = cdiv(h, c3k5 ? 48 : 32); p.tiles_x = cdiv(wd, 32);
    const long total = (long)n * p.tiles_y * p.tiles_x;
    int blocks = (int)(total < TINY_BLOCKS ? total : TINY_BLOCKS);
    // bf16_ok: the caller runs the throughput mode (bf16 matrix operands everywhere): the matrix-core form where the columns tile
    static const bool no_mfma = getenv("NIMG_NO_C3K5_MFMA") != nullptr;
    // 8-row tiles: 24 KB of LDS, six workgroups per CU cover each other's load latency (16-row tiles, three per CU: 180 us)
    static const int tr8 = getenv("NIMG_C3K5_TR8") ? atoi(getenv("NIMG_C3K5_TR8")) : 1;
    if (c3k5 && bf16_ok && !no_mfma && pad == 2 && wd % 64 == 0 && h >= 4) {
        const int tr = tr8 ? 8 : 16;
        const long tot2 = (long)n * ((h + 4 + tr - 1) / tr) * (wd / 64);
        const long cap = tr8 ? 2 * TINY_BLOCKS : TINY_BLOCKS;
        blocks = (int)(tot2 < cap ? tot2 : cap);
        if (tr8) hipLaunchKernelGGL(conv_wgrad_c3k5_mfma_kernel<8>, dim3(blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(conv_wgrad_c3k5_mfma_kernel<16>, dim3(blocks), dim3(256), 0, s, p);
    } else if (c3k5) hipLaunchKernelGGL(conv_wgrad_c3k5_kernel, dim3(blocks), dim3(256), 0, s, p);
    else if (ks == 3 && cin == 3 && cout == 3) hipLaunchKernelGGL((conv_wgrad_tiny_kernel<3, 3, 3>), dim3(blocks), dim3(256), 0, s, p);
    else return NIMG_ERR_ARG;
    NIMG_CHECK_LAUNCH();
    const int count = ks * ks * cin * cout;
    hipLaunchKernelGGL(tiny_reduce_kernel, dim3(count), dim3(64), 0, s, (const float*)workspace, dw, count, blocks,
                       accumulate);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}
