// Fused differentiable JPEG for gfx950 (CDNA4): colour transform, 8x8 block DCT, quantise, IDCT, inverse colour
// and clip in ONE kernel per direction.  Replaces models/jpeg.py:91-159 (about 25 TensorFlow ops with >=12
// materialised intermediates).  HBM-bound: forward reads 12 B + writes 12 B (+1 B mask) per pixel.
//
// Work decomposition (wave64-native):
//   one wavefront = one strip of 8 image rows x 64 pixels = 8 JPEG blocks x 3 channels
//   lane (b = lane>>3, r = lane&7) owns row r of block b for all three channels (24 values in registers)
//   pass 1 (along the row, in registers) -> 8x8 transpose through LDS (stride-9 tiles, conflict-free)
//   pass 2 (along the column, in registers) -> quantise -> inverse pass 2 -> transpose back -> inverse pass 1
//   global <-> LDS staging moves whole 768-byte row segments as float4 (coalesced), 6 per lane.
//
// CANONICAL float32 ORDER (shared with oracle/djpeg_ref.c, compared bit-for-bit on the index tensor):
//   every dot product is an fmaf chain in ascending index order starting from the first product;
//   row pass first (T = b * F^T), column pass second (X = F * T); IDCT column pass first, row pass second.
//   Compiled with -ffp-contract=off so the explicit fmaf()s are the only fused operations.
#include <type_traits>
#include "common.h"

namespace {

using namespace nimg;

constexpr int STRIP_PX = 64;              // pixels per wave strip
constexpr int ROW_STRIDE = 196;           // padded LDS row stride (floats), 16-byte aligned
constexpr int STAGE_F = 8 * ROW_STRIDE;   // staging floats per wave
constexpr int TILE_F = 3 * 8 * 72;        // transpose tiles: [ch][block][8][9]
constexpr int WAVE_LDS_F = (STAGE_F > TILE_F ? STAGE_F : TILE_F);
constexpr int WAVES_PER_BLOCK = 4;
// workgroups per CU the register allocation aims at (LDS holds 5): A/B builds override these (csrc/Makefile EXTRA=)
#ifndef DJ_FWD_WGS
#define DJ_FWD_WGS 4
#endif
#ifndef DJ_BWD_WGS
#define DJ_BWD_WGS 3
#endif

// Every LDS region of these kernels (staging rows, transpose tiles) is PRIVATE to one wave, and a wave's LDS operations
// complete in issue order: what has to be prevented is only the compiler moving reads above writes.  A workgroup barrier here
// would make four independent waves wait for each other twice per 8x8 transpose.
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }

// 4-decimal DCT matrix, literal (models/jpeg.py:78-85)
__device__ constexpr float kF[8][8] = {
    {0.3536f, 0.3536f, 0.3536f, 0.3536f, 0.3536f, 0.3536f, 0.3536f, 0.3536f},
    {0.4904f, 0.4157f, 0.2778f, 0.0975f, -0.0975f, -0.2778f, -0.4157f, -0.4904f},
    {0.4619f, 0.1913f, -0.1913f, -0.4619f, -0.4619f, -0.1913f, 0.1913f, 0.4619f},
    {0.4157f, -0.0975f, -0.4904f, -0.2778f, 0.2778f, 0.4904f, 0.0975f, -0.4157f},
    {0.3536f, -0.3536f, -0.3536f, 0.3536f, 0.3536f, -0.3536f, -0.3536f, 0.3536f},
    {0.2778f, -0.4904f, 0.0975f, 0.4157f, -0.4157f, -0.0975f, 0.4904f, -0.2778f},
    {0.1913f, -0.4619f, 0.4619f, -0.1913f, -0.1913f, 0.4619f, -0.4619f, 0.1913f},
    {0.0975f, -0.2778f, 0.4157f, -0.4904f, 0.4904f, -0.4157f, 0.2778f, -0.0975f}};

// colour matrices, bias in column 0 (models/jpeg.py:74-75)
__device__ constexpr float kCF[3][4] = {{0.0f, 0.299f, 0.587f, 0.114f},
                                        {128.0f, -0.168736f, -0.331264f, 0.5f},
                                        {128.0f, 0.5f, -0.418688f, -0.081312f}};
__device__ constexpr float kCI[3][4] = {{(float)(-1.402 * 128), 1.0f, 0.0f, 1.402f},
                                        {(float)(1.058272 * 128), 1.0f, -0.344136f, -0.714136f},
                                        {(float)(-1.772 * 128), 1.0f, 1.772f, 0.0f}};

// out[k] = sum_m in[m] * F[k][m]      (forward DCT along the register axis)
__device__ __forceinline__ void dct_fwd8(const float (&in)[8], float (&out)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float acc = in[0] * kF[k][0];
#pragma unroll
        for (int m = 1; m < 8; ++m) acc = __builtin_fmaf(in[m], kF[k][m], acc);
        out[k] = acc;
    }
}
// out[m] = sum_k in[k] * F[k][m]      (inverse DCT along the register axis)
__device__ __forceinline__ void dct_inv8(const float (&in)[8], float (&out)[8]) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        float acc = in[0] * kF[0][m];
#pragma unroll
        for (int k = 1; k < 8; ++k) acc = __builtin_fmaf(in[k], kF[k][m], acc);
        out[m] = acc;
    }
}

struct Task {
    int n, by, x0;     // image, block row, first pixel of the strip
    bool valid;
};

__device__ __forceinline__ Task decode_task(long task, long n_tasks, int hb, int strips) {
    Task t;
    t.valid = task < n_tasks;
    const long tt = t.valid ? task : 0;
    t.x0 = (int)(tt % strips) * STRIP_PX;
    t.by = (int)((tt / strips) % hb);
    t.n = (int)(tt / ((long)strips * hb));
    return t;
}

// One strip of a task, global <-> registers: 8 rows x up to 64 px x 3 ch as 6 float4 per lane (w % 8 == 0 => 24-float = 96-byte
// block granules, always 16-byte aligned); item it * 64 + lane = (row, float4 column) of the strip.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fetch_strip(const float* __restrict__ src, f32x4 (&pre)[6], const Task& t, int h, int w,
                                            int lane) {
    const int strip_f = min(STRIP_PX, w - t.x0) * 3;      // valid floats per row
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int idx = it * 64 + lane;                   // 0..383 float4 slots
        const int row = idx / 48, c4 = idx % 48;
        if (t.valid && c4 * 4 < strip_f)
            pre[it] = *reinterpret_cast<const f32x4*>(src + (((long)t.n * h + t.by * 8 + row) * w + t.x0) * 3 + c4 * 4);
    }
}
// registers -> LDS staging rows (every slot is written: lanes beyond a ragged strip stage stale values nobody reads back)
__device__ __forceinline__ void commit_strip(float* lds, const f32x4 (&pre)[6], int lane) {
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int idx = it * 64 + lane;
        *reinterpret_cast<f32x4*>(lds + (idx / 48) * ROW_STRIDE + (idx % 48) * 4) = pre[it];
    }
}

__device__ __forceinline__ void store_strip(float* __restrict__ dst, const float* lds, const Task& t, int h, int w,
                                            int lane) {
    const int strip_f = min(STRIP_PX, w - t.x0) * 3;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int idx = it * 64 + lane;
        const int row = idx / 48, c4 = idx % 48;
        if (t.valid && c4 * 4 < strip_f) {
            const float4 v = *reinterpret_cast<const float4*>(lds + row * ROW_STRIDE + c4 * 4);
            *reinterpret_cast<float4*>(dst + (((long)t.n * h + t.by * 8 + row) * w + t.x0) * 3 + c4 * 4) = v;
        }
    }
}

// lane (b,r): read its 8 pixels x 3 channels from the staging rows
__device__ __forceinline__ void read_row24(const float* lds, int b, int r, float (&px)[24]) {
    const float4* p = reinterpret_cast<const float4*>(lds + r * ROW_STRIDE + b * 24);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float4 v = p[k];
        px[4 * k + 0] = v.x; px[4 * k + 1] = v.y; px[4 * k + 2] = v.z; px[4 * k + 3] = v.w;
    }
}
__device__ __forceinline__ void write_row24(float* lds, int b, int r, const float (&px)[24]) {
    float4* p = reinterpret_cast<float4*>(lds + r * ROW_STRIDE + b * 24);
#pragma unroll
    for (int k = 0; k < 6; ++k) p[k] = make_float4(px[4 * k], px[4 * k + 1], px[4 * k + 2], px[4 * k + 3]);
}

// 8x8 transposes among the 8 lanes of a block through LDS tiles [ch][block][8][9]
// "rows -> cols": lane (b,r) holds v[c][k] = M_c[r][k]; afterwards lane (b,r) holds M_c[k][r]  (its column r)
__device__ __forceinline__ void transpose24(float* tile, int b, int r, float (&v)[3][8]) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[((c * 8 + b) * 8 + r) * 9 + k] = v[c][k];
    wave_sync();
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) v[c][k] = tile[((c * 8 + b) * 8 + k) * 9 + r];
    wave_sync();
}

// Rounding approximations (models/jpeg.py:40-52 / 133-149).  QMODE of the forward: ROUND and SOFT both round to nearest.
enum { Q_RINT = 0, Q_SIN = 1, Q_HARMONIC = 2, Q_IDENTITY = 3 };
__host__ __device__ constexpr int fwd_mode(int rounding) {
    return rounding == NIMG_ROUND_SIN ? Q_SIN : rounding == NIMG_ROUND_HARMONIC ? Q_HARMONIC
         : rounding == NIMG_ROUND_IDENTITY ? Q_IDENTITY : Q_RINT;
}
template <int QMODE>
__device__ __forceinline__ float quantise(float u) {
    if constexpr (QMODE == Q_RINT) return rintf(u);                                     // half-to-even == tf.round
    else if constexpr (QMODE == Q_SIN) return u - sinpif(2.0f * (u - rintf(u))) * 0.15915494309189535f;   // sin(2 pi u)/(2 pi)
    else if constexpr (QMODE == Q_HARMONIC) return u - sinpif(2.0f * (u - rintf(u))) * 0.3183098861837907f;   // sin(2 pi u)/pi
    else return u;
}
// gradient modes of the backward: ROUND has none, SOFT and SIN share 1 - cos(2 pi u)
enum { G_ZERO = 0, G_SOFT = 1, G_HARMONIC = 2, G_ONE = 3 };
__host__ __device__ constexpr int bwd_mode(int rounding) {
    return rounding == NIMG_ROUND_ROUND ? G_ZERO : rounding == NIMG_ROUND_HARMONIC ? G_HARMONIC
         : rounding == NIMG_ROUND_IDENTITY ? G_ONE : G_SOFT;
}
template <int GMODE>
__device__ __forceinline__ float quantise_grad(float u) {
    if constexpr (GMODE == G_ZERO) return 0.0f;
    else if constexpr (GMODE == G_ONE) return 1.0f;
    else {
        const float c = __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(u));   // v_cos_f32 takes revolutions
        return GMODE == G_SOFT ? 1.0f - c : 1.0f - 2.0f * c;
    }
}

// x / q.  EXACT = IEEE division.  Otherwise Markstein's correction of x * rc with rc = RN(1 / q): y0 = RN(x rc), r = RN(x - q y0)
// (exact in an fma), y = RN(y0 + r rc) - equal to RN(x / q) for every float x (no underflow) and every INTEGER q in 1 .. 255,
// which is checked exhaustively on the CPU (tools/probe/div_markstein_check.c: 255 divisors x 2^23 mantissas, the identity is
// invariant under scaling x by powers of two); 3 VALU instructions instead of hipcc's 11-instruction division.  x = -0 gives
// +0 (IEEE: -0): invisible in the indices, the image and the mask.  A table with any other entry takes the EXACT path.
template <bool EXACT>
__device__ __forceinline__ float div_q(float x, float q, float rc) {
    if constexpr (EXACT) return x / q;
    const float y0 = x * rc;
    const float r = __builtin_fmaf(-y0, q, x);
    return __builtin_fmaf(r, rc, y0);
}

// rgb (x255) -> ycbcr - 127, for the 8 pixels of this lane's row; out[c][j]
__device__ __forceinline__ void colour_fwd(const float (&px)[24], float (&ycc)[3][8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float r = 255.0f * px[3 * j], g = 255.0f * px[3 * j + 1], bb = 255.0f * px[3 * j + 2];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float acc = kCF[c][0];
            acc = __builtin_fmaf(r, kCF[c][1], acc);
            acc = __builtin_fmaf(g, kCF[c][2], acc);
            acc = __builtin_fmaf(bb, kCF[c][3], acc);
            ycc[c][j] = acc - 127.0f;
        }
    }
}

// The quantisation tables of a workgroup in LDS, transposed to [c][r][u] so that lane (b, r) reads the 8 entries of its column
// with two 16-byte reads (8 distinct addresses per wave), next to their reciprocals; returns whether every entry is an integer
// in 1 .. 255 (the IJG tables: always; trainable tables: not after the first update), i.e. whether div_q<false> is exact.
__device__ __forceinline__ bool stage_tables(const float* __restrict__ qtab, float* qT, int* flag) {
    const int tid = threadIdx.x;
    if (tid == 0) *flag = 1;
    __syncthreads();
    if (tid < 192) {
        const float q = qtab[tid];
        const int c = tid >> 6, u = (tid >> 3) & 7, r = tid & 7;
        qT[(c * 8 + r) * 8 + u] = q;
        qT[192 + (c * 8 + r) * 8 + u] = 1.0f / q;
        if (!(q >= 1.0f && q <= 255.0f && q == rintf(q))) *flag = 0;
    }
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(*flag) != 0;
}
__device__ __forceinline__ void read_tables(const float* qT, int c, int r, float (&q)[8], float (&rc)[8]) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float4 a = *reinterpret_cast<const float4*>(qT + (c * 8 + r) * 8 + 4 * k);
        const float4 b = *reinterpret_cast<const float4*>(qT + 192 + (c * 8 + r) * 8 + 4 * k);
        q[4 * k] = a.x; q[4 * k + 1] = a.y; q[4 * k + 2] = a.z; q[4 * k + 3] = a.w;
        rc[4 * k] = b.x; rc[4 * k + 1] = b.y; rc[4 * k + 2] = b.z; rc[4 * k + 3] = b.w;
    }
}

// column pass, quantisation, inverse column pass of one wave task (v: T columns in, S columns out)
template <int QMODE, bool EXACT>
__device__ __forceinline__ void quantise_columns(float (&v)[3][8], const float* qT, const Task& t, int b, int r, int hb, int wb,
                                                 int16_t* __restrict__ idx, float* __restrict__ xdq) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float X[8], q[8], rc[8];
        dct_fwd8(v[c], X);                                       // X[u][col r] = sum_i F[u][i] T[i][r]
        read_tables(qT, c, r, q, rc);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float qi = quantise<QMODE>(div_q<EXACT>(X[u], q[u], rc[u]));
            X[u] = qi * q[u];
            if (idx || xdq) {
                const long o = ((((long)t.n * 3 + c) * hb + t.by) * wb + (t.x0 / 8 + b)) * 64 + u * 8 + r;
                if (idx) idx[o] = (int16_t)qi;
                if (xdq) xdq[o] = X[u];
            }
        }
        dct_inv8(X, v[c]);                                       // S[i][col r] = sum_u F[u][i] Xd[u][r]
    }
}

// One task per wave; the 5 workgroups a CU holds (LDS) overlap one wave's memory round trip with the others' arithmetic.
template <int QMODE>
__global__ __launch_bounds__(256, DJ_FWD_WGS) void djpeg_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ qtab, uint8_t* __restrict__ mask,
                                                        int16_t* __restrict__ idx, float* __restrict__ xdq, int n,
                                                        int h, int w) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = lane >> 3, r = lane & 7;
    float* lds = smem + wave * WAVE_LDS_F;
    float* qT = smem + WAVES_PER_BLOCK * WAVE_LDS_F;
    const int hb = h / 8, wb = w / 8, strips = (w + STRIP_PX - 1) / STRIP_PX;
    const long n_tasks = (long)n * hb * strips;
    const Task t = decode_task((long)blockIdx.x * WAVES_PER_BLOCK + wave, n_tasks, hb, strips);
    const bool active = t.valid && (t.x0 + b * 8 < w);
    constexpr float kRc255 = 1.0f / 255.0f;

    f32x4 pre[6] = {};
    fetch_strip(x, pre, t, h, w, lane);                              // in flight while the tables are staged
    const bool fast = stage_tables(qtab, qT, reinterpret_cast<int*>(qT + 384));
    commit_strip(lds, pre, lane);
    wave_sync();
    float px[24];
    float v[3][8];
    read_row24(lds, b, r, px);
    wave_sync();
    {
        float ycc[3][8];
        colour_fwd(px, ycc);
#pragma unroll
        for (int c = 0; c < 3; ++c) dct_fwd8(ycc[c], v[c]);          // T[r][k] = sum_j b[r][j] F[k][j]
    }
    transpose24(lds, b, r, v);                                       // lane now holds T[0..7][col r]
    if (active) {
        if (fast) quantise_columns<QMODE, false>(v, qT, t, b, r, hb, wb, idx, xdq);
        else quantise_columns<QMODE, true>(v, qT, t, b, r, hb, wb, idx, xdq);
    }
    transpose24(lds, b, r, v);                                       // lane holds S[row r][0..7]
    {
        float q3[3][8];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dct_inv8(v[c], q3[c]);                                   // xi[r][j] = sum_v S[r][v] F[v][j]
#pragma unroll
            for (int j = 0; j < 8; ++j) q3[c][j] += 127.0f;
        }
        uint32_t mlo = 0, mhi = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint32_t m = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float acc = kCI[c][0];
                acc = __builtin_fmaf(q3[0][j], kCI[c][1], acc);
                acc = __builtin_fmaf(q3[1][j], kCI[c][2], acc);
                acc = __builtin_fmaf(q3[2][j], kCI[c][3], acc);
                acc = div_q<false>(acc, 255.0f, kRc255);
                const float cl = __builtin_amdgcn_fmed3f(acc, 0.0f, 1.0f);          // = fminf(fmaxf(acc, 0), 1)
                m |= cl == acc ? (1u << c) : 0u;                                    // inside [0, 1]: the clip passes the gradient
                px[3 * j + c] = cl;
            }
            if (j < 4) mlo |= m << (8 * j); else mhi |= m << (8 * (j - 4));
        }
        write_row24(lds, b, r, px);
        if (mask && active)
            *reinterpret_cast<uint2*>(mask + ((long)t.n * h + t.by * 8 + r) * w + t.x0 + b * 8) = make_uint2(mlo, mhi);
    }
    wave_sync();
    store_strip(y, lds, t, h, w, lane);
}

// DQ: also the gradient of the quantisation tables (trainable=True, models/jpeg.py:57-62; X' = quant(X / Q) * Q at :129-131):
// d X' / d Q = quant(z) - z quant'(z) with z = X / Q, summed over every block of the luma / of both chroma channels.  A lane
// holds column r of its block, so it accumulates 2 x 8 table entries (u, r); the 8 blocks of the wave are folded with three
// shuffles and lanes 0..7 write the wave's 128 partial sums (dq_partial[task][cls][u][r]); a fixed-order reduction finishes.
// Always the IEEE division (trained tables are not integers).
template <int RND, bool DQ>
__global__ __launch_bounds__(256, DQ ? 3 : DJ_BWD_WGS) void djpeg_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                        const uint8_t* __restrict__ mask,
                                                        const float* __restrict__ qtab, float* __restrict__ gx,
                                                        float* __restrict__ dq_partial, int n, int h, int w) {
    constexpr int GMODE = bwd_mode(RND), QMODE = fwd_mode(RND);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = lane >> 3, r = lane & 7;
    float* lds = smem + wave * WAVE_LDS_F;
    float* qT = smem + WAVES_PER_BLOCK * WAVE_LDS_F;
    const int hb = h / 8, strips = (w + STRIP_PX - 1) / STRIP_PX;
    const long n_tasks = (long)n * hb * strips;
    const Task t = decode_task((long)blockIdx.x * WAVES_PER_BLOCK + wave, n_tasks, hb, strips);
    const bool active = t.valid && (t.x0 + b * 8 < w);
    f32x4 pre[6] = {};
    fetch_strip(x, pre, t, h, w, lane);                              // in flight while the tables are staged
    const bool fast = stage_tables(qtab, qT, reinterpret_cast<int*>(qT + 384)) && !DQ;

    float px[24];
    float tx[3][8];    // forward coefficients path (from x)
    float tg[3][8];    // gradient path (from gy)
    // ---- recompute the forward row pass from x
    commit_strip(lds, pre, lane);
    fetch_strip(gy, pre, t, h, w, lane);                             // the gradient strip arrives under the first row pass
    uint2 mm = make_uint2(0u, 0u);
    if (active) mm = *reinterpret_cast<const uint2*>(mask + ((long)t.n * h + t.by * 8 + r) * w + t.x0 + b * 8);
    wave_sync();
    read_row24(lds, b, r, px);
    wave_sync();
    {
        float ycc[3][8];
        colour_fwd(px, ycc);
#pragma unroll
        for (int c = 0; c < 3; ++c) dct_fwd8(ycc[c], tx[c]);
    }
    // ---- gradient: clip mask, /255, transpose(colour_I), row pass (d xi = F^T dXd F  =>  dXd = F g F^T)
    commit_strip(lds, pre, lane);
    wave_sync();
    read_row24(lds, b, r, px);
    wave_sync();
    {
#pragma clang fp contract(fast)       // gradient arithmetic carries no bit-exact contract (the recomputed forward above does)
        float gq[3][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t m = ((j < 4 ? mm.x >> (8 * j) : mm.y >> (8 * (j - 4)))) & 0xffu;
            float g[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) g[c] = (m >> c) & 1u ? px[3 * j + c] * (1.0f / 255.0f) : 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c)       // d/d(q_c) = sum_c' g_c' * CI[c'][1+c]
                gq[c][j] = g[0] * kCI[0][1 + c] + g[1] * kCI[1][1 + c] + g[2] * kCI[2][1 + c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) dct_fwd8(gq[c], tg[c]);
    }
    transpose24(lds, b, r, tx);
    transpose24(lds, b, r, tg);
    float dqa[DQ ? 2 : 1][8];
    if constexpr (DQ) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int u = 0; u < 8; ++u) dqa[k][u] = 0.0f;
    }
    if (active) {
        auto columns = [&](auto exact) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float X[8], G[8], q[8], rc[8];
                dct_fwd8(tx[c], X);
                dct_fwd8(tg[c], G);
                read_tables(qT, c, r, q, rc);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float z = div_q<decltype(exact)::value>(X[u], q[u], rc[u]);
                    const float qg = quantise_grad<GMODE>(z);
                    if constexpr (DQ) dqa[c == 0 ? 0 : 1][u] += G[u] * (quantise<QMODE>(z) - z * qg);
                    G[u] *= qg;                                      // (X/Q -> quant -> *Q): the Q factors cancel
                }
                dct_inv8(G, tg[c]);                                  // d b = F^T gX F : column pass
            }
        };
        if (fast) columns(std::false_type{});
        else columns(std::true_type{});
    }
    if constexpr (DQ) {
        const long task = (long)blockIdx.x * WAVES_PER_BLOCK + wave;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float v = dqa[k][u];
                v += __shfl_xor(v, 8, 64);
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (lane < 8) dq_partial[task * 128 + (k * 8 + u) * 8 + r] = v;
            }
    }
    transpose24(lds, b, r, tg);
    {
#pragma clang fp contract(fast)
        float gb[3][8];
#pragma unroll
        for (int c = 0; c < 3; ++c) dct_inv8(tg[c], gb[c]);          // row pass
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c)       // d/d(x_c) = 255 * sum_c' gycc_c' * CF[c'][1+c]
                px[3 * j + c] =
                    255.0f * (gb[0][j] * kCF[0][1 + c] + gb[1][j] * kCF[1][1 + c] + gb[2][j] * kCF[2][1 + c]);
        write_row24(lds, b, r, px);
    }
    wave_sync();
    store_strip(gx, lds, t, h, w, lane);
}

inline int persistent_grid(long tasks) { return (int)((tasks + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK); }   // one task per wave
constexpr size_t kLds = ((size_t)WAVES_PER_BLOCK * WAVE_LDS_F + 384 + 4) * sizeof(float);

template <bool DQ, typename... A>
void launch_bwd(int rounding, int grid, hipStream_t s, A... a) {
    const dim3 g(grid), b(256);
    switch (rounding) {
        case NIMG_ROUND_ROUND: hipLaunchKernelGGL((djpeg_bwd_kernel<NIMG_ROUND_ROUND, DQ>), g, b, kLds, s, a...); break;
        case NIMG_ROUND_SIN:                               // without the table gradient SIN and SOFT are one kernel
            if constexpr (DQ) { hipLaunchKernelGGL((djpeg_bwd_kernel<NIMG_ROUND_SIN, DQ>), g, b, kLds, s, a...); break; }
            [[fallthrough]];
        case NIMG_ROUND_SOFT: hipLaunchKernelGGL((djpeg_bwd_kernel<NIMG_ROUND_SOFT, DQ>), g, b, kLds, s, a...); break;
        case NIMG_ROUND_HARMONIC: hipLaunchKernelGGL((djpeg_bwd_kernel<NIMG_ROUND_HARMONIC, DQ>), g, b, kLds, s, a...); break;
        default: hipLaunchKernelGGL((djpeg_bwd_kernel<NIMG_ROUND_IDENTITY, DQ>), g, b, kLds, s, a...); break;
    }
}

}  // namespace

extern "C" {

int nimg_abi_version(void) { return NIMG_ABI_VERSION; }

int nimg_djpeg_fwd(const float* x, float* y, const float* qtab, uint8_t* mask, int16_t* idx, float* xdq, int n,
                   int h, int w, int rounding, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!x || !y || !qtab || n < 0 || h <= 0 || w <= 0 || (h % 8) || (w % 8)) return NIMG_ERR_ARG;
    if (rounding < NIMG_ROUND_ROUND || rounding > NIMG_ROUND_IDENTITY) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    const long tasks = (long)n * (h / 8) * ((w + STRIP_PX - 1) / STRIP_PX);
    const dim3 grid(persistent_grid(tasks)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (fwd_mode(rounding)) {
        case Q_RINT: hipLaunchKernelGGL(djpeg_fwd_kernel<Q_RINT>, grid, block, kLds, s, x, y, qtab, mask, idx, xdq, n, h, w); break;
        case Q_SIN: hipLaunchKernelGGL(djpeg_fwd_kernel<Q_SIN>, grid, block, kLds, s, x, y, qtab, mask, idx, xdq, n, h, w); break;
        case Q_HARMONIC:
            hipLaunchKernelGGL(djpeg_fwd_kernel<Q_HARMONIC>, grid, block, kLds, s, x, y, qtab, mask, idx, xdq, n, h, w);
            break;
        default: hipLaunchKernelGGL(djpeg_fwd_kernel<Q_IDENTITY>, grid, block, kLds, s, x, y, qtab, mask, idx, xdq, n, h, w); break;
    }
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_djpeg_bwd(const float* x, const float* gy, const uint8_t* mask, const float* qtab, float* gx, int n, int h,
                   int w, int rounding, void* stream) {
    if (n == 0) return NIMG_OK;        /* empty batch: nothing to do (its buffers may be null) */
    if (!x || !gy || !mask || !qtab || !gx || n < 0 || h <= 0 || w <= 0 || (h % 8) || (w % 8)) return NIMG_ERR_ARG;
    if (rounding < NIMG_ROUND_ROUND || rounding > NIMG_ROUND_IDENTITY) return NIMG_ERR_ARG;
    if (n == 0) return NIMG_OK;
    const long tasks = (long)n * (h / 8) * ((w + STRIP_PX - 1) / STRIP_PX);
    launch_bwd<false>(rounding, persistent_grid(tasks), (hipStream_t)stream, x, gy, mask, qtab, gx, (float*)nullptr, n, h, w);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

size_t nimg_djpeg_dq_workspace_bytes(int n, int h, int w) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    const long tasks = (long)n * (h / 8) * ((w + STRIP_PX - 1) / STRIP_PX);
    return (size_t)nimg::cdiv(tasks, WAVES_PER_BLOCK) * WAVES_PER_BLOCK * 128 * sizeof(float);
}

int nimg_djpeg_bwd_dq(const float* x, const float* gy, const uint8_t* mask, const float* qtab, float* gx, float* dq,
                      int n, int h, int w, int rounding, int accumulate, void* workspace, size_t workspace_bytes,
                      void* stream) {
    if (n == 0) return NIMG_OK;
    if (!x || !gy || !mask || !qtab || !gx || !dq || !workspace || n < 0 || h <= 0 || w <= 0 || (h % 8) || (w % 8))
        return NIMG_ERR_ARG;
    if (rounding < NIMG_ROUND_ROUND || rounding > NIMG_ROUND_IDENTITY) return NIMG_ERR_ARG;
    if (workspace_bytes < nimg_djpeg_dq_workspace_bytes(n, h, w)) return NIMG_ERR_WORKSPACE;
    const long tasks = (long)n * (h / 8) * ((w + STRIP_PX - 1) / STRIP_PX);
    const int grid = persistent_grid(tasks);
    launch_bwd<true>(rounding, grid, (hipStream_t)stream, x, gy, mask, qtab, gx, (float*)workspace, n, h, w);
    NIMG_CHECK_LAUNCH();
    nimg::launch_reduce2((const float*)workspace, dq, 128, grid * WAVES_PER_BLOCK, nullptr, nullptr, 0, 0, accumulate,
                         (hipStream_t)stream);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

// IJG quality scaling (compression/jpeg_helpers.py:264-305), host side
int nimg_jpeg_qtable(int quality, int channel, float* out64) {
    static const float luma[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55,
                                   14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                   18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                                   49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
    static const float chroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                                     24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
    if (!out64) return NIMG_ERR_ARG;
    quality = quality > 100 ? 100 : (quality < 1 ? 1 : quality);
    const float s = quality < 50 ? (float)(5000.0 / quality) : (float)(200 - 2 * quality);
    const float* t = channel == 0 ? luma : chroma;
    for (int k = 0; k < 64; ++k) {
        float v = __builtin_floorf((t[k] * s + 50.0f) / 100.0f);
        out64[k] = v < 1.0f ? 1.0f : (v > 255.0f ? 255.0f : v);
    }
    return NIMG_OK;
}

}  // extern "C"
