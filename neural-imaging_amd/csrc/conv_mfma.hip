// NHWC implicit-GEMM convolution on the CDNA4 matrix cores, float32 in / float32 accumulate
// (v_mfma_f32_32x32x2_f32: exact f32, bitwise an fmaf chain - the parity mode of the channel).
//
// Replaces the tf.keras.layers.Conv2D call sites of the reference:
//   UNet 3x3 s1 SAME (+bias +LeakyReLU)      models/pipelines.py:191-192,207-208,216
//   FAN  5x5 s1 SAME (+bias +LeakyReLU), 1x1 models/forensics.py:69,76
//   DCN  5x5 s2 SAME (TF asymmetric pad), 3x3 models/compression.py:221-237,247-265
// and their input gradients (dgrad = the same kernel run on flipped/transposed weights, see
// nimg_conv_flip_weights), with the LeakyReLU derivative of the PREVIOUS layer fused in the epilogue.
//
// im2col-free: GEMM  M = output pixels (32 per MFMA tile = one 2-D patch row group), N = Cout, K = taps x Cin.
//   A operand  A[i][k] = in[pixel_i + tap][ci0 + k]   read from an LDS halo tile stored CHANNEL-MAJOR
//              [ci][pixel] (plane stride = 2 mod 32 banks): lanes 0-31 read 32 consecutive pixels of channel k,
//              lanes 32-63 the same pixels of channel k+1  -> conflict-free ds_read_b32
//   B operand  B[k][j] = w[tap][ci0 + k][co0 + j]      LDS [tap][ci][co], lanes read consecutive co
//   HBM reads  float4 per lane along the channel axis (NHWC => coalesced), transposed on the LDS write.
//   Concat-free: the input may be split over two tensors (UNet skip connections, pipelines.py:206,211),
//   the output over two tensors (dgrad of a concat input).
#include "common.h"

// defined in conv_small.hip
int nimg_internal_conv_fewout(const float* in, int cin, const float* w, const float* bias, float* out, int cout, int n,
                              int h, int wd, int ks, int pad_t, int pad_l, int pad_mode, int hout, int wout,
                              hipStream_t s);

namespace {

using namespace nimg;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const float* in1;
    const float* in2;
    const float* w;       // [KS*KS][Cin][Cout]
    const float* bias;    // [Cout] or nullptr
    float* out1;
    float* out2;
    const float* act1;    // optional saved activation, same shape as out1: out1 *= lrelu'(act1)
    float* pool_out;      // optional: fused activation + 2x2 max-pool output [N][Hout/2][Wout/2][Cout] (replaces out1)
    unsigned char* pool_idx;   // optional argmax (0..3) per pooled element
    int C1, C2, O1, O2;   // Cin = C1 + C2, Cout = O1 + O2
    int N, H, W, Hout, Wout, pad_t, pad_l;
    int tiles_y, tiles_x;
    int act;              // 0 none, 1 leaky relu
    int pad_mode;         // 0 zeros, 1 SYMMETRIC, 2 REFLECT (tf.pad modes folded into the tile load)
    float alpha;
};

template <int KS, int STRIDE, int TH, int TW, int NB, int TN, int CK, bool VEC>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const ConvParams p) {
    constexpr int THH = (TH - 1) * STRIDE + KS, TWH = (TW - 1) * STRIDE + KS;
    constexpr int NPIXH = NB * THH * TWH;
    constexpr int PS = ((NPIXH + 31) / 32) * 32 + 2;
    constexpr int MFRAGS = NB * TH * TW / 32;
    constexpr int NFRAGS = TN / 32;
    constexpr int WAVES_M = MFRAGS >= 4 ? 4 : MFRAGS;
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int MI = MFRAGS / WAVES_M;
    constexpr int NI = NFRAGS / WAVES_N;
    constexpr int TAPS = KS * KS;
    static_assert(MFRAGS % WAVES_M == 0 && NFRAGS % WAVES_N == 0 && NI >= 1, "bad tile configuration");
    static_assert((NB * TH * TW) % 32 == 0 && CK % 4 == 0, "bad tile configuration");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                 // [CK][PS]
    float* sB = smem + CK * PS;       // [TAPS][CK][TN]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int Cin = p.C1 + p.C2, Cout = p.O1 + p.O2;

    // block -> (co tile, spatial tile, image group); co tile fastest so neighbours reuse the input tile in L2/MALL
    const int cot = (Cout + TN - 1) / TN;
    int bid = xcd_order(blockIdx.x);
    const int co0 = (bid % cot) * TN;
    bid /= cot;
    const int tiles = p.tiles_y * p.tiles_x;
    const int tile = bid % tiles, grp = bid / tiles;
    const int ty0 = (tile / p.tiles_x) * TH, tx0 = (tile % p.tiles_x) * TW;
    const int iy0 = ty0 * STRIDE - p.pad_t, ix0 = tx0 * STRIDE - p.pad_l;

    // per-lane A base offsets (pixel part) for its MI fragments
    int abase[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int P = (wm * MI + mi) * 32 + (lane & 31);
        const int img = P / (TH * TW), rem = P % (TH * TW);
        abase[mi] = img * (THH * TWH) + (rem / TW) * STRIDE * TWH + (rem % TW) * STRIDE;
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.0f;

    for (int ci0 = 0; ci0 < Cin; ci0 += CK) {
        __syncthreads();   // previous chunk's MFMA reads are done before the tiles are overwritten
        // ---- stage A: halo tile of CK input channels, transposed to channel-major
        if (VEC) {
            constexpr int C4 = CK / 4;
            for (int item = tid; item < NPIXH * C4; item += 256) {
                const int pix = item / C4, c = ci0 + (item % C4) * 4;
                const int img = pix / (THH * TWH), rem = pix % (THH * TWH);
                int gy = iy0 + rem / TWH, gx = ix0 + rem % TWH;
                const int n = grp * NB + img;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < p.N && c < Cin && map_coord(gy, p.H, p.pad_mode) && map_coord(gx, p.W, p.pad_mode)) {
                    const long pixoff = ((long)n * p.H + gy) * p.W + gx;
                    v = c < p.C1 ? *reinterpret_cast<const float4*>(p.in1 + pixoff * p.C1 + c)
                                 : *reinterpret_cast<const float4*>(p.in2 + pixoff * p.C2 + (c - p.C1));
                }
                float* d = sA + ((item % C4) * 4) * PS + pix;
                d[0] = v.x; d[PS] = v.y; d[2 * PS] = v.z; d[3 * PS] = v.w;
            }
        } else {
            for (int item = tid; item < NPIXH * CK; item += 256) {
                const int pix = item / CK, k = item % CK, c = ci0 + k;
                const int img = pix / (THH * TWH), rem = pix % (THH * TWH);
                int gy = iy0 + rem / TWH, gx = ix0 + rem % TWH;
                const int n = grp * NB + img;
                float v = 0.f;
                if (n < p.N && c < Cin && map_coord(gy, p.H, p.pad_mode) && map_coord(gx, p.W, p.pad_mode)) {
                    const long pixoff = ((long)n * p.H + gy) * p.W + gx;
                    v = c < p.C1 ? p.in1[pixoff * p.C1 + c] : p.in2[pixoff * p.C2 + (c - p.C1)];
                }
                sA[k * PS + pix] = v;
            }
        }
        // ---- stage B: weights of this channel chunk, all taps
        if (VEC) {
            constexpr int J4 = TN / 4;
            for (int item = tid; item < TAPS * CK * J4; item += 256) {
                const int j = (item % J4) * 4, k = (item / J4) % CK, tap = item / (J4 * CK);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ci0 + k < Cin && co0 + j < Cout)
                    v = *reinterpret_cast<const float4*>(p.w + ((long)tap * Cin + ci0 + k) * Cout + co0 + j);
                *reinterpret_cast<float4*>(sB + (tap * CK + k) * TN + j) = v;
            }
        } else {
            for (int item = tid; item < TAPS * CK * TN; item += 256) {
                const int j = item % TN, k = (item / TN) % CK, tap = item / (TN * CK);
                float v = 0.f;
                if (ci0 + k < Cin && co0 + j < Cout) v = p.w[((long)tap * Cin + ci0 + k) * Cout + co0 + j];
                sB[(tap * CK + k) * TN + j] = v;
            }
        }
        __syncthreads();
        // ---- MFMA: taps x channel pairs
        const int kpairs = min(CK, (Cin - ci0 + 1) & ~1) / 2;
        const float* aL = sA + (lane >> 5) * PS;
        const float* bL = sB + (lane >> 5) * TN + wn * NI * 32 + (lane & 31);
#pragma unroll 1
        for (int tap = 0; tap < TAPS; ++tap) {
            const int toff = (tap / KS) * TWH + (tap % KS);
            for (int kp = 0; kp < kpairs; ++kp) {
                float a[MI], b[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) a[mi] = aL[(2 * kp) * PS + abase[mi] + toff];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) b[ni] = bL[(tap * CK + 2 * kp) * TN + ni * 32];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            }
        }
    }

    // ---- epilogue, fused activation + 2x2 max-pool (16x16 tiles only; the entry point guarantees even Hout/Wout, Cout % 4 == 0)
    if constexpr (TW == 16 && NB == 1 && STRIDE == 1) {
        if (p.pool_out) {
            float* elds = reinterpret_cast<float*>(smem) + wave * (32 * (NI * 32 + EPI_PAD));
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
                        *reinterpret_cast<float4*>(p.pool_out + o) = v;
                        if (p.pool_idx) *reinterpret_cast<uchar4*>(p.pool_idx + o) = k;
                    });
            }
            return;
        }
    }
    // ---- epilogue, vector form: accumulators turned around through LDS so each lane stores 16 B along the channels
    if ((p.O1 & 3) == 0 && (p.O2 & 3) == 0) {
        float* elds = reinterpret_cast<float*>(smem) + wave * (32 * (NI * 32 + EPI_PAD));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            epilogue_via_lds<NI>(acc[mi], elds, lane, [&](int row, int c, float4 v) {
                const int co = co0 + wn * NI * 32 + c;
                if (co >= Cout) return;
                const int P = (wm * MI + mi) * 32 + row;
                const int img = P / (TH * TW), rem = P % (TH * TW);
                const int oy = ty0 + rem / TW, ox = tx0 + rem % TW, n = grp * NB + img;
                if (n >= p.N || oy >= p.Hout || ox >= p.Wout) return;
                const long pixoff = ((long)n * p.Hout + oy) * p.Wout + ox;
                if (p.bias) {
                    const float4 b = *reinterpret_cast<const float4*>(p.bias + co);
                    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                }
                if (p.act == 1) {
                    v.x = lrelu(v.x, p.alpha); v.y = lrelu(v.y, p.alpha); v.z = lrelu(v.z, p.alpha); v.w = lrelu(v.w, p.alpha);
                }
                if (co < p.O1) {
                    if (p.act1) {
                        const float4 m = *reinterpret_cast<const float4*>(p.act1 + pixoff * p.O1 + co);
                        v.x *= m.x > 0.f ? 1.0f : p.alpha; v.y *= m.y > 0.f ? 1.0f : p.alpha;
                        v.z *= m.z > 0.f ? 1.0f : p.alpha; v.w *= m.w > 0.f ? 1.0f : p.alpha;
                    }
                    *reinterpret_cast<float4*>(p.out1 + pixoff * p.O1 + co) = v;
                } else {
                    *reinterpret_cast<float4*>(p.out2 + pixoff * p.O2 + (co - p.O1)) = v;
                }
            });
        }
        return;
    }
    // ---- epilogue: bias, activation, optional lrelu' mask of the previous layer, NHWC store (128 B per half-wave)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int co = co0 + (wn * NI + ni) * 32 + (lane & 31);
        if (co >= Cout) continue;
        const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int row = (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
                const int P = (wm * MI + mi) * 32 + row;
                const int img = P / (TH * TW), rem = P % (TH * TW);
                const int oy = ty0 + rem / TW, ox = tx0 + rem % TW, n = grp * NB + img;
                if (n >= p.N || oy >= p.Hout || ox >= p.Wout) continue;
                const long pixoff = ((long)n * p.Hout + oy) * p.Wout + ox;
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
}

// ------------------------------------------------------------------------------------------------------------------
// Few input channels (Cin <= 4: FAN conv1 3->32, UNet ec11 4->32, ConstrainedConv2D input): the reduction index packs
// (tap, ci) densely, K = KS*KS*Cin, so no MFMA k-slot is wasted on channel padding and the weights (which are
// contiguous in exactly that order, [tap][ci][co]) are staged ONCE per workgroup; the workgroup then walks
// TILES_PER_WG spatial tiles.  Stride 1, one input tensor.
template <int KS, int CINP, int TN>
__global__ __launch_bounds__(256) void conv_fwd_packed_kernel(const ConvParams p, int tiles_per_wg) {
    constexpr int TH = 16, TW = 16;
    constexpr int THH = TH + KS - 1, TWH = TW + KS - 1;
    constexpr int NPIXH = THH * TWH;
    constexpr int PS = ((NPIXH + 31) / 32) * 32 + 2;
    constexpr int KTOT = KS * KS * CINP;
    constexpr int KPAIRS = (KTOT + 1) / 2;
    constexpr int NI = TN / 32;
    constexpr int MI = 2;                              // 8 M fragments over 4 waves
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                                  // [CINP][PS]
    float* sB = smem + CINP * PS;                      // [2*KPAIRS][TN]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cout = p.O1 + p.O2;
    const int cot = (Cout + TN - 1) / TN;
    const int xbid = xcd_order(blockIdx.x);
    const int co0 = (xbid % cot) * TN;
    const int wg = xbid / cot;
    const int tiles = p.tiles_y * p.tiles_x;
    const long total_tiles = (long)tiles * p.N;

    // weights: rows k = tap*CINP + ci are contiguous in memory; pad row (if KTOT is odd) is zero
    for (int item = tid; item < 2 * KPAIRS * TN; item += 256) {
        const int j = item % TN, k = item / TN;
        sB[item] = (k < KTOT && co0 + j < Cout) ? p.w[(long)k * Cout + co0 + j] : 0.f;
    }
    int abase[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int P = (wave * MI + mi) * 32 + (lane & 31);
        abase[mi] = (P / TW) * TWH + (P % TW);
    }
    for (int tt = 0; tt < tiles_per_wg; ++tt) {
        const long gt = (long)wg * tiles_per_wg + tt;
        if (gt >= total_tiles) break;
        const int n = (int)(gt / tiles), tile = (int)(gt % tiles);
        const int ty0 = (tile / p.tiles_x) * TH, tx0 = (tile % p.tiles_x) * TW;
        __syncthreads();
        for (int pix = tid; pix < NPIXH; pix += 256) {
            int gy = ty0 - p.pad_t + pix / TWH, gx = tx0 - p.pad_l + pix % TWH;
            const bool ok = map_coord(gy, p.H, p.pad_mode) && map_coord(gx, p.W, p.pad_mode);
            const float* src = p.in1 + (((long)n * p.H + gy) * p.W + gx) * CINP;
#pragma unroll
            for (int c = 0; c < CINP; ++c) sA[c * PS + pix] = ok ? src[c] : 0.f;
        }
        __syncthreads();
        f32x16 acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.0f;
        const float* bL = sB + (lane >> 5) * TN + (lane & 31);
#pragma unroll 2
        for (int kp = 0; kp < KPAIRS; ++kp) {
            int k = 2 * kp + (lane >> 5);
            k = k < KTOT ? k : KTOT - 1;               // the padded slot multiplies a zero weight row
            const int tap = k / CINP, ci = k - tap * CINP;
            const int aoff = ci * PS + (tap / KS) * TWH + (tap % KS);
            float a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = sA[aoff + abase[mi]];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = bL[(2 * kp) * TN + ni * 32];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (p.pool_out) {                  // fused activation + 2x2 max-pool (see common.h); private per-wave scratch
            float* elds = smem + CINP * PS + 2 * KPAIRS * TN + wave * (32 * (NI * 32 + EPI_PAD));
            const int Hp = p.Hout >> 1, Wp = p.Wout >> 1;
            const float al = p.act == 1 ? p.alpha : 1.0f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int py = (ty0 >> 1) + wave * MI + mi;
                pool_via_lds<NI, false>(acc[mi], elds, lane, al,
                    [&](int c) {
                        return (p.bias && co0 + c < Cout) ? *reinterpret_cast<const float4*>(p.bias + co0 + c)
                                                          : make_float4(0.f, 0.f, 0.f, 0.f);
                    },
                    [&](int pc, int c, float4 v, uchar4 k) {
                        const int co = co0 + c, px = (tx0 >> 1) + pc;
                        if (co >= Cout || py >= Hp || px >= Wp) return;
                        const long o = (((long)n * Hp + py) * Wp + px) * Cout + co;
                        *reinterpret_cast<float4*>(p.pool_out + o) = v;
                        if (p.pool_idx) *reinterpret_cast<uchar4*>(p.pool_idx + o) = k;
                    });
            }
            continue;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int co = co0 + ni * 32 + (lane & 31);
            if (co >= Cout) continue;
            const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int P = (wave * MI + mi) * 32 + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
                    const int oy = ty0 + P / TW, ox = tx0 + P % TW;
                    if (oy >= p.Hout || ox >= p.Wout) continue;
                    const long pixoff = ((long)n * p.Hout + oy) * p.Wout + ox;
                    float v = acc[mi][ni][j] + bv;
                    if (p.act == 1) v = lrelu(v, p.alpha);
                    if (p.act1) v *= (p.act1[pixoff * Cout + co] > 0.f ? 1.0f : p.alpha);
                    p.out1[pixoff * Cout + co] = v;
                }
        }
    }
}

template <int KS, int CINP, int TN>
int launch_conv_packed(const ConvParams& p, hipStream_t stream) {
    constexpr int THH = 16 + KS - 1, NPIXH = THH * THH;
    constexpr int PS = ((NPIXH + 31) / 32) * 32 + 2;
    constexpr int KP = (KS * KS * CINP + 1) / 2;
    constexpr size_t lds = (size_t)(CINP * PS + 2 * KP * TN + 4 * 32 * (TN + EPI_PAD)) * sizeof(float);
    ConvParams q = p;
    q.tiles_y = cdiv(p.Hout, 16);
    q.tiles_x = cdiv(p.Wout, 16);
    const long total_tiles = (long)q.tiles_y * q.tiles_x * p.N;
    const int tpw = total_tiles >= 8192 ? 8 : (total_tiles >= 2048 ? 2 : 1);
    const long blocks = cdiv(total_tiles, tpw) * (long)cdiv(p.O1 + p.O2, TN);
    hipLaunchKernelGGL((conv_fwd_packed_kernel<KS, CINP, TN>), dim3((unsigned)blocks), dim3(256), lds, stream, q, tpw);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

template <int KS, int STRIDE, int TH, int TW, int NB, int TN, int CK, bool VEC>
int launch_conv(const ConvParams& p, hipStream_t stream) {
    constexpr int THH = (TH - 1) * STRIDE + KS, TWH = (TW - 1) * STRIDE + KS;
    constexpr int NPIXH = NB * THH * TWH;
    constexpr int PS = ((NPIXH + 31) / 32) * 32 + 2;
    constexpr size_t lds_tiles = (size_t)(CK * PS + KS * KS * CK * TN) * sizeof(float);
    constexpr size_t lds_epi = (size_t)4 * 32 * (TN + EPI_PAD) * sizeof(float);
    constexpr size_t lds = lds_tiles > lds_epi ? lds_tiles : lds_epi;
    ConvParams q = p;
    q.tiles_y = cdiv(p.Hout, TH);
    q.tiles_x = cdiv(p.Wout, TW);
    const int Cout = p.O1 + p.O2;
    const long blocks = (long)cdiv(Cout, TN) * q.tiles_y * q.tiles_x * cdiv(p.N, NB);
    auto kern = conv_fwd_kernel<KS, STRIDE, TH, TW, NB, TN, CK, VEC>;
    static bool attr_set = false;   // opt in to > 64 KiB dynamic LDS once per instantiation
    if (!attr_set) {
        // the 6x6 ... 11x11 tiles take 93 - 121 KB: a part with less LDS than gfx950's 160 KB refuses here - report the
        // configuration as unsupported instead of failing at the launch
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return NIMG_ERR_ARG;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, q);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

template <int KS, int STRIDE, int CK>
int dispatch_tiles(const ConvParams& p, bool vec, hipStream_t stream) {
    const int Cout = p.O1 + p.O2;
    const bool small = (p.Hout <= 8 && p.Wout <= 8);
    // narrow N tile when the problem is short of workgroups (keeps >= ~2 blocks per CU)
    const long blocks64 = (long)cdiv(Cout, 64) * cdiv(p.Hout, small ? 8 : 16) * cdiv(p.Wout, small ? 8 : 16) *
                          cdiv(p.N, small ? 4 : 1);
    const bool tn32 = (Cout <= 32) || (blocks64 < 512 && Cout % 64 != 0) || (blocks64 < 384);
#define NIMG_LAUNCH(TH_, TW_, NB_, TN_)                                                            \
    (vec ? launch_conv<KS, STRIDE, TH_, TW_, NB_, TN_, CK, true>(p, stream)                          \
         : launch_conv<KS, STRIDE, TH_, TW_, NB_, TN_, CK, false>(p, stream))
    if (small) return tn32 ? NIMG_LAUNCH(8, 8, 4, 32) : NIMG_LAUNCH(8, 8, 4, 64);
    return tn32 ? NIMG_LAUNCH(16, 16, 1, 32) : NIMG_LAUNCH(16, 16, 1, 64);
#undef NIMG_LAUNCH
}

__global__ void flip_weights_kernel(const float* __restrict__ w, float* __restrict__ wt, int taps, int cin,
                                    int cout) {
    // wt[taps-1-t][co][ci] = w[t][ci][co]
    const long total = (long)taps * cin * cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin);
        const int co = (int)((i / cin) % cout);
        const int t = (int)(i / ((long)cin * cout));
        wt[i] = w[((long)(taps - 1 - t) * cin + ci) * cout + co];
    }
}

}  // namespace

extern "C" {

int nimg_conv2d_fwd(const float* in1, int c1, const float* in2, int c2, const float* w, const float* bias,
                    float* out1, int o1, float* out2, int o2, const float* act_mask, int n, int h, int wd, int ks,
                    int stride, int pad_t, int pad_l, int pad_mode, int hout, int wout, int act, float alpha,
                    void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!in1 || !w || !out1 || c1 <= 0 || c2 < 0 || o1 <= 0 || o2 < 0 || n < 0 || h <= 0 || wd <= 0) return NIMG_ERR_ARG;
    if ((c2 > 0 && !in2) || (o2 > 0 && !out2) || hout <= 0 || wout <= 0 || pad_t < 0 || pad_l < 0) return NIMG_ERR_ARG;
    if (act < 0 || act > 1 || pad_mode < 0 || pad_mode > 2) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    ConvParams p;
    p.in1 = in1; p.in2 = in2; p.w = w; p.bias = bias; p.out1 = out1; p.out2 = out2; p.act1 = act_mask;
    p.pool_out = nullptr; p.pool_idx = nullptr;
    p.C1 = c1; p.C2 = c2; p.O1 = o1; p.O2 = o2; p.N = n; p.H = h; p.W = wd; p.Hout = hout; p.Wout = wout;
    p.pad_t = pad_t; p.pad_l = pad_l; p.tiles_y = p.tiles_x = 0; p.act = act; p.alpha = alpha; p.pad_mode = pad_mode;
    const bool vec = (c1 % 4 == 0) && (c2 % 4 == 0) && ((o1 + o2) % 4 == 0);
    hipStream_t s = (hipStream_t)stream;
    const bool plain = (c2 == 0 && o2 == 0 && stride == 1);
    // few output channels (<= 4): register-blocked VALU kernel (conv_small.hip)
    if (plain && o1 == 3 && (ks == 3 || ks == 5) && !act_mask && act == 0)
        return nimg_internal_conv_fewout(in1, c1, w, bias, out1, o1, n, h, wd, ks, pad_t, pad_l, pad_mode, hout, wout, s);
    // few input channels (<= 4): (tap, ci)-packed reduction on the matrix core
    if (plain && (c1 == 3 || c1 == 4) && (ks == 3 || ks == 5) && o1 >= 8) {
        const bool n64 = o1 > 32;
#define NIMG_PACKED(KS_, C_) (n64 ? launch_conv_packed<KS_, C_, 64>(p, s) : launch_conv_packed<KS_, C_, 32>(p, s))
        if (ks == 5) return c1 == 3 ? NIMG_PACKED(5, 3) : NIMG_PACKED(5, 4);
        return c1 == 3 ? NIMG_PACKED(3, 3) : NIMG_PACKED(3, 4);
#undef NIMG_PACKED
    }
    if (stride == 1) {
        if (ks == 1) return dispatch_tiles<1, 1, 16>(p, vec, s);
        if (ks == 3) return dispatch_tiles<3, 1, 16>(p, vec, s);
        if (ks == 5) return dispatch_tiles<5, 1, 8>(p, vec, s);
        // 7x7 ... 11x11 (FAN `kernel`, forensics.py:51; the demosaicing filters, pipelines.py:242,419): the same kernel with a
        // 4-channel K chunk - the weight tile of a chunk is KS^2 x 4 x TN floats (11x11, TN = 64: 121 KB of LDS)
        if (ks == 7) return dispatch_tiles<7, 1, 4>(p, vec, s);
        if (ks == 9) return dispatch_tiles<9, 1, 4>(p, vec, s);
        if (ks == 11) return dispatch_tiles<11, 1, 4>(p, vec, s);
        // even sizes (the FAN's `kernel` is any integer 3 .. 11): TF's SAME padding puts the extra row / column AFTER the image -
        // the pads are explicit arguments here, so it is the same kernel (the input gradient takes pads KS - 1 - pad)
        if (ks == 4) return dispatch_tiles<4, 1, 8>(p, vec, s);
        if (ks == 6) return dispatch_tiles<6, 1, 4>(p, vec, s);
        if (ks == 8) return dispatch_tiles<8, 1, 4>(p, vec, s);
        if (ks == 10) return dispatch_tiles<10, 1, 4>(p, vec, s);
    } else if (stride == 2) {
        if (ks == 2) return dispatch_tiles<2, 2, 16>(p, vec, s);
        if (ks == 5) return dispatch_tiles<5, 2, 8>(p, vec, s);
    }
    return NIMG_ERR_ARG;
}

int nimg_conv2d_pool_fwd(const float* in, int cin, const float* w, const float* bias, float* pool_out,
                         unsigned char* pool_idx, int cout, int n, int h, int wd, int ks, int act, float alpha,
                         void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!in || !w || !pool_out || cin <= 0 || cout <= 0 || (cout & 3) || n < 0 || h <= 0 || wd <= 0) return NIMG_ERR_ARG;
    if ((h & 1) || (wd & 1) || (ks != 3 && ks != 5) || act < 0 || act > 1) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    ConvParams p;
    p.in1 = in; p.in2 = nullptr; p.w = w; p.bias = bias; p.out1 = nullptr; p.out2 = nullptr; p.act1 = nullptr;
    p.pool_out = pool_out; p.pool_idx = pool_idx;
    p.C1 = cin; p.C2 = 0; p.O1 = cout; p.O2 = 0; p.N = n; p.H = h; p.W = wd; p.Hout = h; p.Wout = wd;
    p.pad_t = p.pad_l = (ks - 1) / 2; p.tiles_y = p.tiles_x = 0; p.act = act; p.alpha = alpha; p.pad_mode = 0;
    hipStream_t s = (hipStream_t)stream;
    if (cin == 3 || cin == 4) {
        const bool n64 = cout > 32;
#define NIMG_PACKED(KS_, C_) (n64 ? launch_conv_packed<KS_, C_, 64>(p, s) : launch_conv_packed<KS_, C_, 32>(p, s))
        if (ks == 5) return cin == 3 ? NIMG_PACKED(5, 3) : NIMG_PACKED(5, 4);
        return cin == 3 ? NIMG_PACKED(3, 3) : NIMG_PACKED(3, 4);
#undef NIMG_PACKED
    }
    const bool vec = cin % 4 == 0;
    const bool tn32 = cout <= 32 || (long)cdiv(cout, 64) * cdiv(h, 16) * cdiv(wd, 16) * n < 384;
#define NIMG_POOL(KS_, CK_)                                                                                   \
    (vec ? (tn32 ? launch_conv<KS_, 1, 16, 16, 1, 32, CK_, true>(p, s) : launch_conv<KS_, 1, 16, 16, 1, 64, CK_, true>(p, s)) \
         : (tn32 ? launch_conv<KS_, 1, 16, 16, 1, 32, CK_, false>(p, s) : launch_conv<KS_, 1, 16, 16, 1, 64, CK_, false>(p, s)))
    return ks == 3 ? NIMG_POOL(3, 16) : NIMG_POOL(5, 8);
#undef NIMG_POOL
}

int nimg_conv_flip_weights(const float* w, float* wt, int ks_h, int ks_w, int cin, int cout, void* stream) {
    if (!w || !wt || ks_h <= 0 || ks_w <= 0 || cin <= 0 || cout <= 0) return NIMG_ERR_ARG;
    const long total = (long)ks_h * ks_w * cin * cout;
    const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(flip_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, wt, ks_h * ks_w, cin,
                       cout);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

}  // extern "C"
