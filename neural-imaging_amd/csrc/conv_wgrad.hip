// Weight (and bias) gradients of the NHWC convolutions on the CDNA4 matrix cores, float32.
//
//   dw[tap][ci][co] = sum_{n,y,x} in[n, y*s + ky - pt, x*s + kx - pl, ci] * dz[n, y, x, co]
//
// i.e. the tape.gradient(..., weights) half of the reference's training steps
// (models/pipelines.py:84-88, models/forensics.py:118-124, workflows/manipulation_classification.py:280).
//
// GEMM view: M = ci (32 / MFMA tile), N = co (32 / tile), K = output pixels (2 per v_mfma_f32_32x32x2_f32).
// Both operands are read from LDS tiles kept in their natural NHWC order (channel contiguous), so staging is a straight
// float4 copy and the per-lane ds_read_b32 is conflict-free (lanes 0-31 = 32 consecutive channels of one pixel, lanes
// 32-63 the next pixel).  The dz fragment is shared by all taps; the taps are split over the 4 waves of a workgroup so
// the accumulators (taps/4 x 2 x 16 registers) stay in the register file for the whole K loop.
// Split-K over (image, tile) ranges: every workgroup writes its partial dw to a workspace slab and a second kernel
// reduces the slabs in a fixed order (deterministic; no float atomics) and also produces the bias gradient.
#include "common.h"

// defined in conv_small.hip
size_t nimg_internal_wgrad_tiny_bytes(int ks, int cin, int cout);
int nimg_internal_conv_wgrad_tiny(const float* in, const float* dz, float* dw, int cin, int cout, int n, int h, int wd,
                                  int ks, int pad, int pad_mode, int accumulate, void* workspace, hipStream_t s, bool bf16_ok);


namespace {

using namespace nimg;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradParams {
    const float* in1;
    const float* in2;
    const float* dz;
    float* partial;       // [splits][taps][Cin][Cout]
    float* db_partial;    // optional [splits][Cout]: fused bias gradient (column sums of dz)
    int C1, C2, Cout;
    int N, H, W, Hout, Wout, pad_t, pad_l;
    int tiles_y, tiles_x, splits, work_per_split, pad_mode;
    int tap0;             // first tap of this launch (kernels above 5x5 take their taps in passes of TCH)
};

constexpr int WG_TH = 8, WG_TW = 16;     // output-pixel tile per K iteration
constexpr int CI_T = 32, CO_T = 64;      // dw block per workgroup

// NW waves share the taps (4, or 8 for 5x5: 4 waves would each hold 7 taps x 32 = 224 accumulator registers and run
// alone on their SIMD; with 8 it is 128 and two waves per SIMD hide each other's LDS / global latency)
// TCH < KS * KS (7x7 ... 11x11): the launch covers taps [p.tap0, p.tap0 + TCH) only - 49 ... 121 taps x 2 x 16 accumulators do
// not fit the register file, so the host walks the taps in passes over the same tiles (each pass its own part of the slabs;
// the bias sums come from pass 0)
template <int KS, int STRIDE, bool VEC, int NW, int TCH = KS * KS>
__global__ __launch_bounds__(NW * 64, 2) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int TAPS = KS * KS, NTHR = NW * 64;
    constexpr int NT = (TCH + NW - 1) / NW;            // taps per wave
    constexpr bool PASSES = TCH < TAPS;
    const int tap_first = PASSES ? p.tap0 : 0, tap_end = PASSES ? min(TAPS, p.tap0 + TCH) : TAPS;
    constexpr int THH = (WG_TH - 1) * STRIDE + KS, TWH = (WG_TW - 1) * STRIDE + KS;
    constexpr int NPIXH = THH * TWH, NPIX = WG_TH * WG_TW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sI = smem;                  // [NPIXH][CI_T]
    float* sZ = smem + NPIXH * CI_T;   // [NPIX][CO_T]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int Cin = p.C1 + p.C2;
    const int cib = (Cin + CI_T - 1) / CI_T, cob = (p.Cout + CO_T - 1) / CO_T;
    int bid = xcd_order(blockIdx.x);
    const int ci0 = (bid % cib) * CI_T;
    bid /= cib;
    const int co0 = (bid % cob) * CO_T;
    const int split = bid / cob;

    f32x16 acc[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[t][ni][j] = 0.0f;

    const int tiles = p.tiles_y * p.tiles_x;
    const int work_total = p.N * tiles;                 // < 2^31 (checked by the entry point)
    const int w_begin = split * p.work_per_split;
    const int w_end = min(work_total, w_begin + p.work_per_split);
    const bool do_bias = p.db_partial && ci0 == 0 && tap_first == 0;
    float bsum = 0.f;

    for (int wk = w_begin; wk < w_end; ++wk) {
        const int n = (int)(wk / tiles), tile = (int)(wk % tiles);
        const int ty0 = (tile / p.tiles_x) * WG_TH, tx0 = (tile % p.tiles_x) * WG_TW;
        const int iy0 = ty0 * STRIDE - p.pad_t, ix0 = tx0 * STRIDE - p.pad_l;
        __syncthreads();
        // ---- stage the input halo tile [pixel][ci] and the dz tile [pixel][co]
        if (VEC) {
            for (int item = tid; item < NPIXH * (CI_T / 4); item += NTHR) {
                const int pix = item / (CI_T / 4), c = ci0 + (item % (CI_T / 4)) * 4;
                int gy = iy0 + pix / TWH, gx = ix0 + pix % TWH;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < Cin && map_coord(gy, p.H, p.pad_mode) && map_coord(gx, p.W, p.pad_mode)) {
                    const long pixoff = ((long)n * p.H + gy) * p.W + gx;
                    v = c < p.C1 ? *reinterpret_cast<const float4*>(p.in1 + pixoff * p.C1 + c)
                                 : *reinterpret_cast<const float4*>(p.in2 + pixoff * p.C2 + (c - p.C1));
                }
                *reinterpret_cast<float4*>(sI + pix * CI_T + (item % (CI_T / 4)) * 4) = v;
            }
            for (int item = tid; item < NPIX * (CO_T / 4); item += NTHR) {
                const int pix = item / (CO_T / 4), c = co0 + (item % (CO_T / 4)) * 4;
                const int oy = ty0 + pix / WG_TW, ox = tx0 + pix % WG_TW;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (oy < p.Hout && ox < p.Wout && c < p.Cout)
                    v = *reinterpret_cast<const float4*>(p.dz + (((long)n * p.Hout + oy) * p.Wout + ox) * p.Cout + c);
                *reinterpret_cast<float4*>(sZ + pix * CO_T + (item % (CO_T / 4)) * 4) = v;
            }
        } else {
            for (int item = tid; item < NPIXH * CI_T; item += NTHR) {
                const int pix = item / CI_T, c = ci0 + item % CI_T;
                int gy = iy0 + pix / TWH, gx = ix0 + pix % TWH;
                float v = 0.f;
                if (c < Cin && map_coord(gy, p.H, p.pad_mode) && map_coord(gx, p.W, p.pad_mode)) {
                    const long pixoff = ((long)n * p.H + gy) * p.W + gx;
                    v = c < p.C1 ? p.in1[pixoff * p.C1 + c] : p.in2[pixoff * p.C2 + (c - p.C1)];
                }
                sI[item] = v;
            }
            for (int item = tid; item < NPIX * CO_T; item += NTHR) {
                const int pix = item / CO_T, c = co0 + item % CO_T;
                const int oy = ty0 + pix / WG_TW, ox = tx0 + pix % WG_TW;
                float v = 0.f;
                if (oy < p.Hout && ox < p.Wout && c < p.Cout)
                    v = p.dz[(((long)n * p.Hout + oy) * p.Wout + ox) * p.Cout + c];
                sZ[item] = v;
            }
        }
        __syncthreads();
        if (do_bias && tid < CO_T) {                       // fused bias gradient: column sums of the dz tile
#pragma unroll 8
            for (int px = 0; px < NPIX; ++px) bsum += sZ[px * CO_T + tid];
        }
        // ---- K loop over pixel pairs; lane half (lane>>5) selects the pixel of the pair
        const float* zL = sZ + (lane >> 5) * CO_T + (lane & 31);
        const float* iL = sI + (lane >> 5) * STRIDE * CI_T + (lane & 31);
#pragma unroll 1
        for (int r = 0; r < WG_TH; ++r) {
#pragma unroll 2
            for (int cp = 0; cp < WG_TW / 2; ++cp) {
                const int pix = r * WG_TW + 2 * cp;
                const float b0 = zL[pix * CO_T], b1 = zL[pix * CO_T + 32];
                const int ibase = (r * STRIDE * TWH + 2 * cp * STRIDE) * CI_T;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int tap = tap_first + wave + NW * t;
                    if (tap < tap_end) {
                        const float a = iL[ibase + ((tap / KS) * TWH + (tap % KS)) * CI_T];
                        acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[t][0], 0, 0, 0);
                        acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[t][1], 0, 0, 0);
                    }
                }
            }
        }
    }

    if (do_bias && tid < CO_T && co0 + tid < p.Cout) p.db_partial[(long)split * p.Cout + co0 + tid] = bsum;
    // ---- write the partial slab: rows = ci (M), cols = co (N)
    float* slab = p.partial + (long)split * TAPS * Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tap = tap_first + wave + NW * t;
        if (tap >= tap_end) continue;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int co = co0 + ni * 32 + (lane & 31);
            if (co >= p.Cout) continue;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int ci = ci0 + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
                if (ci < Cin) slab[((long)tap * Cin + ci) * p.Cout + co] = acc[t][ni][j];
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Few input channels (Cin <= 4): the M dimension packs (tap, ci) - 10 taps x 3 channels or 8 taps x 4 channels per
// 32-row MFMA tile - instead of padding 3 channels to 32.  Every wave holds ALL the accumulators and the 4 waves
// split the rows of each pixel tile, so a workgroup produces 4 partial slabs (slab index = split * 4 + wave).
template <int KS, int CINP, int NI>
__global__ __launch_bounds__(256) void conv_wgrad_packed_kernel(const WgradParams p) {
    constexpr int TAPS = KS * KS;
    constexpr int TPF = 32 / CINP;                       // taps per M fragment
    constexpr int MF = (TAPS + TPF - 1) / TPF;           // M fragments
    constexpr int THH = WG_TH + KS - 1, TWH = WG_TW + KS - 1;
    constexpr int NPIXH = THH * TWH, NPIX = WG_TH * WG_TW;
    constexpr int COT = 32 * NI;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sI = smem;                    // [NPIXH][CINP]
    float* sZ = smem + NPIXH * CINP;     // [NPIX][COT]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cob = (p.Cout + COT - 1) / COT;
    const int xbid = xcd_order(blockIdx.x);
    const int co0 = (xbid % cob) * COT;
    const int split = xbid / cob;

    // per-lane (tap, ci) of each M fragment -> offset inside the halo tile, or -1 when the slot is padding
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
    const int w_begin = split * p.work_per_split;
    const int w_end = min(work_total, w_begin + p.work_per_split);
    const bool do_bias = p.db_partial != nullptr;
    float bsum = 0.f;
    for (int wk = w_begin; wk < w_end; ++wk) {
        const int n = (int)(wk / tiles), tile = (int)(wk % tiles);
        const int ty0 = (tile / p.tiles_x) * WG_TH, tx0 = (tile % p.tiles_x) * WG_TW;
        __syncthreads();
        for (int pix = tid; pix < NPIXH; pix += 256) {
            int gy = ty0 - p.pad_t + pix / TWH, gx = tx0 - p.pad_l + pix % TWH;
            const bool ok = map_coord(gy, p.H, p.pad_mode) && map_coord(gx, p.W, p.pad_mode);
            const float* src = p.in1 + (((long)n * p.H + gy) * p.W + gx) * CINP;
#pragma unroll
            for (int c = 0; c < CINP; ++c) sI[pix * CINP + c] = ok ? src[c] : 0.f;
        }
        for (int item = tid; item < NPIX * COT; item += 256) {
            const int pix = item / COT, c = co0 + item % COT;
            const int oy = ty0 + pix / WG_TW, ox = tx0 + pix % WG_TW;
            float v = 0.f;
            if (oy < p.Hout && ox < p.Wout && c < p.Cout)
                v = p.dz[(((long)n * p.Hout + oy) * p.Wout + ox) * p.Cout + c];
            sZ[item] = v;
        }
        __syncthreads();
        if (do_bias && tid < COT) {
#pragma unroll 8
            for (int px = 0; px < NPIX; ++px) bsum += sZ[px * COT + tid];
        }
        const float* zL = sZ + (lane >> 5) * COT + (lane & 31);
        for (int r = wave; r < WG_TH; r += 4) {
#pragma unroll 2
            for (int cp = 0; cp < WG_TW / 2; ++cp) {
                const int pix = r * WG_TW + 2 * cp;
                float b[NI];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) b[ni] = zL[pix * COT + ni * 32];
                const int ibase = (r * TWH + 2 * cp + (lane >> 5)) * CINP;
#pragma unroll
                for (int f = 0; f < MF; ++f) {
                    const float a = aoff[f] >= 0 ? sI[ibase + aoff[f]] : 0.f;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[f][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[ni], acc[f][ni], 0, 0, 0);
                }
            }
        }
    }
    if (do_bias && tid < COT && co0 + tid < p.Cout) p.db_partial[(long)split * p.Cout + co0 + tid] = bsum;
    float* slab = p.partial + ((long)split * 4 + wave) * TAPS * CINP * p.Cout;
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int co = co0 + ni * 32 + (lane & 31);
            if (co >= p.Cout) continue;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int i = (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
                const int tl = i / CINP, ci = i % CINP, tap = f * TPF + tl;
                if (tl < TPF && tap < TAPS) slab[((long)tap * CINP + ci) * p.Cout + co] = acc[f][ni][j];
            }
        }
}

// db[co] = sum_pixels dz[pixel][co]: one workgroup per slice of pixels, two-stage (partials then fixed-order reduce)
__global__ __launch_bounds__(256) void bias_grad_partial_kernel(const float* __restrict__ dz,
                                                                float* __restrict__ partial, long npix, int cout,
                                                                long pix_per_block) {
    // thread -> channel (tid % cpad), pixel phase (tid / cpad)
    extern __shared__ float red[];
    const int tid = threadIdx.x;
    const int cpad = cout >= 256 ? 256 : (cout > 128 ? 256 : (cout > 64 ? 128 : (cout > 32 ? 64 : 32)));
    const int phases = 256 / cpad;
    const int c = tid % cpad, ph = tid / cpad;
    const long p0 = (long)blockIdx.x * pix_per_block, p1 = min(npix, p0 + pix_per_block);
    for (int cb = 0; cb < cout; cb += cpad) {
        float s = 0.f;
        if (cb + c < cout)
            for (long px = p0 + ph; px < p1; px += phases) s += dz[px * cout + cb + c];
        red[tid] = s;
        __syncthreads();
        if (ph == 0 && cb + c < cout) {
            for (int k = 1; k < phases; ++k) s += red[k * cpad + c];
            partial[(long)blockIdx.x * cout + cb + c] = s;
        }
        __syncthreads();
    }
}

// float4 variant (cout % 4 == 0 and cout/4 divides 256): 16-byte loads, 4 independent accumulators per thread
// BF: dz is stored as bf16 (8-byte loads of 4 channels, summed in float32)
typedef __bf16 bg_bf16x4 __attribute__((ext_vector_type(4)));
template <bool BF>
struct BiasSrc {
    const void* base;
    __device__ __forceinline__ float4 operator[](long i) const {
        if constexpr (BF) {
            const bg_bf16x4 v = reinterpret_cast<const bg_bf16x4*>(base)[i];
            return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
        } else {
            return reinterpret_cast<const float4*>(base)[i];
        }
    }
};
template <bool BF>
__global__ __launch_bounds__(256) void bias_grad_partial4_kernel(const float* __restrict__ dz,
                                                                 float* __restrict__ partial, long npix, int cout,
                                                                 long pix_per_block) {
    __shared__ float4 red4[256];
    const int tid = threadIdx.x, cq = cout >> 2, phases = 256 / cq;
    const int c = tid % cq, ph = tid / cq;
    const long p0 = (long)blockIdx.x * pix_per_block, p1 = min(npix, p0 + pix_per_block);
    const BiasSrc<BF> srcb{dz};
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    long px = p0 + ph;
    for (; px + 3L * phases < p1; px += 4L * phases) {
        const float4 v0 = srcb[px * cq + c], v1 = srcb[(px + phases) * cq + c], v2 = srcb[(px + 2L * phases) * cq + c],
                     v3 = srcb[(px + 3L * phases) * cq + c];
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
        a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; px < p1; px += phases) {
        const float4 v = srcb[px * cq + c];
        a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w;
    }
    red4[tid] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y),
                            (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
    __syncthreads();
    if (ph == 0) {
        float4 s = red4[c];
        for (int k = 1; k < phases; ++k) {
            const float4 v = red4[k * cq + c];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4*>(partial + (long)blockIdx.x * cout)[c] = s;
    }
}

}  // namespace

extern "C" {

// number of split-K slabs the launcher will use for this problem (so the caller can size the workspace)
static int wgrad_splits(int cin, int cout, int n, int hout, int wout) {
    const long blocks_io = (long)nimg::cdiv(cin, CI_T) * nimg::cdiv(cout, CO_T);
    const long work = (long)n * nimg::cdiv(hout, WG_TH) * nimg::cdiv(wout, WG_TW);
    long splits = (512 + blocks_io - 1) / blocks_io;
    if (splits > work) splits = work;
    if (splits < 1) splits = 1;
    const long wps = (work + splits - 1) / splits;
    return (int)((work + wps - 1) / wps);
}

static int packed_splits(int cout, int n, int hout, int wout) {
    const long blocks_io = nimg::cdiv(cout, cout <= 32 ? 32 : 64);
    const long work = (long)n * nimg::cdiv(hout, WG_TH) * nimg::cdiv(wout, WG_TW);
    long splits = (512 + blocks_io - 1) / blocks_io;
    if (splits > work) splits = work;
    if (splits < 1) splits = 1;
    const long wps = (work + splits - 1) / splits;
    return (int)((work + wps - 1) / wps);
}

size_t nimg_conv2d_wgrad_workspace_bytes(int cin, int cout, int ks_h, int ks_w, int n, int hout, int wout) {
    if (cin <= 0 || cout <= 0 || n <= 0) return 0;
    const size_t slab = (size_t)ks_h * ks_w * cin * cout * sizeof(float);
    const size_t generic = (slab + cout * sizeof(float)) * wgrad_splits(cin, cout, n, hout, wout);
    const size_t packed = cin <= 4 ? (4 * slab + cout * sizeof(float)) * packed_splits(cout, n, hout, wout) : 0;
    const size_t tiny = (cin <= 4 && cout <= 4) ? nimg_internal_wgrad_tiny_bytes(ks_h, cin, cout) : 0;
    const size_t m = generic > packed ? generic : packed;
    return m > tiny ? m : tiny;
}

int nimg_conv2d_wgrad(const float* in1, int c1, const float* in2, int c2, const float* dz, int cout, float* dw,
                      float* db, int n, int h, int wd, int ks, int stride, int pad_t, int pad_l, int pad_mode, int hout,
                      int wout, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (!in1 || !dz || !dw || c1 <= 0 || c2 < 0 || cout <= 0 || n <= 0 || h <= 0 || wd <= 0) return NIMG_ERR_ARG;
    if ((c2 > 0 && !in2) || hout <= 0 || wout <= 0 || !workspace || pad_mode < 0 || pad_mode > 2) return NIMG_ERR_ARG;
    const int cin = c1 + c2;
    if (workspace_bytes < nimg_conv2d_wgrad_workspace_bytes(cin, cout, ks, ks, n, hout, wout)) return NIMG_ERR_WORKSPACE;
    WgradParams p;
    p.in1 = in1; p.in2 = in2; p.dz = dz; p.partial = (float*)workspace; p.db_partial = nullptr;
    p.C1 = c1; p.C2 = c2; p.Cout = cout; p.N = n; p.H = h; p.W = wd; p.Hout = hout; p.Wout = wout;
    p.pad_t = pad_t; p.pad_l = pad_l; p.pad_mode = pad_mode;
    p.tiles_y = nimg::cdiv(hout, WG_TH); p.tiles_x = nimg::cdiv(wout, WG_TW);
    p.splits = wgrad_splits(cin, cout, n, hout, wout);
    const long work = (long)n * p.tiles_y * p.tiles_x;
    p.work_per_split = (int)((work + p.splits - 1) / p.splits);
    const bool vec = (c1 % 4 == 0) && (c2 % 4 == 0) && (cout % 4 == 0);
    hipStream_t s = (hipStream_t)stream;
    const long count = (long)ks * ks * cin * cout;

    // tiny filters (Cin, Cout <= 4): one thread per weight, exact f32 (conv_small.hip)
    if (c2 == 0 && c1 == 3 && cout == 3 && stride == 1 && (ks == 3 || ks == 5) && hout == h && wout == wd &&
        pad_t == (ks - 1) / 2 && pad_l == pad_t && !db)
        return nimg_internal_conv_wgrad_tiny(in1, dz, dw, c1, cout, n, h, wd, ks, pad_t, pad_mode, accumulate, workspace, s, false);
    // few input channels: (tap, ci)-packed M dimension
    if (c2 == 0 && (c1 == 3 || c1 == 4) && stride == 1 && (ks == 3 || ks == 5)) {
        const int ni = cout <= 32 ? 1 : 2;
        p.splits = packed_splits(cout, n, hout, wout);
        p.work_per_split = (int)((work + p.splits - 1) / p.splits);
        if (db) p.db_partial = p.partial + (size_t)4 * p.splits * count;
        const long pblocks = (long)nimg::cdiv(cout, 32 * ni) * p.splits;
#define NIMG_WGP(KS_, C_, NI_)                                                                                 \
        do {                                                                                                  \
            constexpr size_t lds = (size_t)((WG_TH + KS_ - 1) * (WG_TW + KS_ - 1) * C_ + WG_TH * WG_TW * 32 * NI_) * \
                                   sizeof(float);                                                             \
            hipLaunchKernelGGL((conv_wgrad_packed_kernel<KS_, C_, NI_>), dim3((unsigned)pblocks), dim3(256), lds, s, p); \
        } while (0)
        if (ks == 5 && c1 == 3) { if (ni == 1) NIMG_WGP(5, 3, 1); else NIMG_WGP(5, 3, 2); }
        else if (ks == 5) { if (ni == 1) NIMG_WGP(5, 4, 1); else NIMG_WGP(5, 4, 2); }
        else if (c1 == 3) { if (ni == 1) NIMG_WGP(3, 3, 1); else NIMG_WGP(3, 3, 2); }
        else { if (ni == 1) NIMG_WGP(3, 4, 1); else NIMG_WGP(3, 4, 2); }
#undef NIMG_WGP
        NIMG_CHECK_LAUNCH();
        nimg::launch_reduce2((const float*)workspace, dw, count, 4 * p.splits, db ? (const float*)p.db_partial : nullptr,
                             db, (long)cout, p.splits, accumulate, s);
        NIMG_CHECK_LAUNCH();
        return NIMG_OK;
    }
    if (db) p.db_partial = p.partial + (size_t)p.splits * count;
    const long blocks = (long)nimg::cdiv(cin, CI_T) * nimg::cdiv(cout, CO_T) * p.splits;

#define NIMG_WG(KS_, ST_)                                                                                     \
    do {                                                                                                      \
        constexpr int THH = (WG_TH - 1) * ST_ + KS_, TWH = (WG_TW - 1) * ST_ + KS_;                           \
        constexpr size_t lds = (size_t)(THH * TWH * CI_T + WG_TH * WG_TW * CO_T) * sizeof(float);             \
        constexpr int NW = KS_ == 5 ? 8 : 4;                                                                  \
        if (vec) {                                                                                            \
            auto k = conv_wgrad_kernel<KS_, ST_, true, NW>;                                                   \
            (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
            hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(NW * 64), lds, s, p);                          \
        } else {                                                                                              \
            auto k = conv_wgrad_kernel<KS_, ST_, false, NW>;                                                  \
            (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
            hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(NW * 64), lds, s, p);                          \
        }                                                                                                     \
    } while (0)

    // 7x7 ... 11x11: ceil(taps / 32) passes of <= 32 taps (8 waves x 4 taps: the register footprint of the 5x5 kernel)
#define NIMG_WG_BIG(KS_)                                                                                      \
    do {                                                                                                      \
        constexpr int THH = WG_TH - 1 + KS_, TWH = WG_TW - 1 + KS_;                                           \
        constexpr size_t lds = (size_t)(THH * TWH * CI_T + WG_TH * WG_TW * CO_T) * sizeof(float);             \
        constexpr int PASSES_ = (KS_ * KS_ + 31) / 32, TCH_ = (KS_ * KS_ + PASSES_ - 1) / PASSES_;            \
        for (int ps = 0; ps < PASSES_; ++ps) {                                                                \
            p.tap0 = ps * TCH_;                                                                               \
            if (vec) {                                                                                        \
                auto k = conv_wgrad_kernel<KS_, 1, true, 8, TCH_>;                                            \
                (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
                hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(512), lds, s, p);                          \
            } else {                                                                                          \
                auto k = conv_wgrad_kernel<KS_, 1, false, 8, TCH_>;                                           \
                (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
                hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(512), lds, s, p);                          \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)
    p.tap0 = 0;
    if (stride == 1 && ks == 1) NIMG_WG(1, 1);
    else if (stride == 1 && ks == 3) NIMG_WG(3, 1);
    else if (stride == 1 && ks == 5) NIMG_WG(5, 1);
    else if (stride == 2 && ks == 2) NIMG_WG(2, 2);
    else if (stride == 2 && ks == 5) NIMG_WG(5, 2);
    else if (stride == 1 && ks == 4) NIMG_WG(4, 1);
    else if (stride == 1 && ks == 6) NIMG_WG_BIG(6);
    else if (stride == 1 && ks == 8) NIMG_WG_BIG(8);
    else if (stride == 1 && ks == 10) NIMG_WG_BIG(10);
    else if (stride == 1 && ks == 7) NIMG_WG_BIG(7);
    else if (stride == 1 && ks == 9) NIMG_WG_BIG(9);
    else if (stride == 1 && ks == 11) NIMG_WG_BIG(11);
    else return NIMG_ERR_ARG;
#undef NIMG_WG
#undef NIMG_WG_BIG
    NIMG_CHECK_LAUNCH();
    nimg::launch_reduce2((const float*)workspace, dw, count, p.splits, db ? (const float*)p.db_partial : nullptr, db,
                         (long)cout, p.splits, accumulate, s);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

static long bias_grad_blocks(long npix) { return npix < 512 ? 1 : (npix / 512 > 2048 ? 2048 : npix / 512); }

size_t nimg_bias_grad_workspace_bytes(long npix, int cout) {
    return (size_t)bias_grad_blocks(npix) * cout * sizeof(float);
}

int nimg_bias_grad(const float* dz, float* db, long npix, int cout, int accumulate, void* workspace,
                   size_t workspace_bytes, void* stream) {
    return nimg_bias_grad_ex(dz, db, npix, cout, accumulate, workspace, workspace_bytes, 0, stream);
}

/* flags: NIMG_BF16_DZ = dz is stored as bf16 (cout % 4 == 0 and cout / 4 divides 256) */
int nimg_bias_grad_ex(const float* dz, float* db, long npix, int cout, int accumulate, void* workspace,
                      size_t workspace_bytes, int flags, void* stream) {
    if (!dz || !db || npix <= 0 || cout <= 0 || !workspace || (flags & ~NIMG_BF16_DZ)) return NIMG_ERR_ARG;
    if ((flags & NIMG_BF16_DZ) && !((cout & 3) == 0 && (cout >> 2) <= 256 && 256 % (cout >> 2) == 0)) return NIMG_ERR_ARG;
    if (workspace_bytes < nimg_bias_grad_workspace_bytes(npix, cout)) return NIMG_ERR_WORKSPACE;
    const long blocks = bias_grad_blocks(npix);
    const long ppb = (npix + blocks - 1) / blocks;
    hipStream_t s = (hipStream_t)stream;
    if (flags & NIMG_BF16_DZ)
        hipLaunchKernelGGL(bias_grad_partial4_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, dz, (float*)workspace,
                           npix, cout, ppb);
    else if ((cout & 3) == 0 && (cout >> 2) <= 256 && 256 % (cout >> 2) == 0)
        hipLaunchKernelGGL(bias_grad_partial4_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, dz, (float*)workspace,
                           npix, cout, ppb);
    else
        hipLaunchKernelGGL(bias_grad_partial_kernel, dim3((unsigned)blocks), dim3(256), 256 * sizeof(float), s, dz,
                           (float*)workspace, npix, cout, ppb);
    NIMG_CHECK_LAUNCH();
    nimg::launch_reduce2((const float*)workspace, db, (long)cout, (int)blocks, nullptr, nullptr, 0, 0, accumulate, s);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

}  // extern "C"
