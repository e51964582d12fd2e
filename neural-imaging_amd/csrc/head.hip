// The FAN's head in throughput mode (models/forensics.py:76-94: Conv2D(nf, 1x1, leaky_relu) -> GlobalAveragePooling2D -> Dense
// softmax) as two skinny bf16 GEMMs with everything around them fused in:
//
//   head_fwd:    act = LeakyReLU(X W + b), gap[n] = mean over the image's pixels of act - the activation tensor itself NEVER
//                reaches HBM: the backward pass only needs the SIGN of each value (LeakyReLU'), which leaves as one bit per
//                value (mask[px][co / 32], 1/32 of the float32 tensor the generic path wrote, re-read for the pooling and re-read
//                twice by the backward pass);
//   head_dgrad:  dX = (dAct W^T) * LeakyReLU'(X), where dAct[px][co] = g[n][co] * (bit ? 1 : alpha), g[n] = Wd dlogits[n] / HW, is
//                BUILT in registers as the matrix instruction's A operand from the mask bits and the 256 values of g - the
//                (N, HW, C) float32 gradient tensor the pooling's backward used to write (84 MB at 320 images) does not exist;
//   head_dact:   that tensor as bf16, for the weight-gradient kernel only (side stream, off the critical path).
//
// Was (C4, 320 images of 16 x 16 x 256): 1x1 forward 57 us + pooling 20 us, pooling backward 32 us + 1x1 input gradient 105 us on
// the launch stream.  Geometry: one workgroup per image, one wave per 32 pixels (HW = 32 NW), N dimension = all C channels in
// NF = C / 32 fragments of v_mfma_f32_32x32x16_bf16 (128 accumulator registers at C = 256), K in chunks of 64 through LDS
// (weights: [row][64 + 8] bf16, register-staged one chunk ahead).  C in {64, 128, 256}, HW in {64, 128, 256}; anything else keeps
// the generic kernels (ops.head_fused_ok).
#include "common.h"

namespace {

using namespace nimg;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct HeadParams {
    const void* x;            // fwd: (N * HW, C) bf16 input of the 1x1 layer
    const void* wimg;         // bf16 weight image (nimg_conv_weights_bf16 layout, taps = 1): fwd mode 0, dgrad mode 1
    const float* bias;        // fwd: (C)
    unsigned* mask;           // (N * HW, C / 32) words, bit j of word f = act[px][32 f + j] > 0.  fwd: out (or null); dgrad: in
    unsigned* mask_p;         // fwd: (N, HW / 32, C) words, bit j of word [n][b][co] = act[pixel 32 b + j][co] > 0 - or null
    float* gap;               // fwd: (N, C) out
    const float* dlogits;     // dgrad: (N, K)
    const float* wdense;      // dgrad: (C, K)
    const void* in_mask;      // dgrad: (N * HW, C) bf16 - LeakyReLU' of the layer below is taken from its sign - or null
    void* dx;                 // dgrad: (N * HW, C) bf16 out
    int N, HW, C, K;
    float alpha;
};

constexpr int ROWB = 144;     // LDS bytes per staged row: 64 bf16 + 16 B pad
constexpr int ROWF = 272;     // dgrad epilogue scratch: 64 float32 + 16 B pad per pixel row

// forward epilogue, accumulator element J of a fragment: bias, LeakyReLU into the pooling sum; the sign bits of the wave's 64 values
// (one ballot = channels l32 of pixel rows r0 and r0 + 4) go to lanes r0 / r0 + 4 of `w` - every lane ends up with the word of ITS row
template <int J>
__device__ __forceinline__ void head_rows(const f32x16& a, float bias, float alpha, bool want_mask, float& s, unsigned& w,
                                          unsigned& colbits) {
    if constexpr (J < 16) {
        const float v = a[J] + bias;
        const bool pos = v > 0.f;
        s += pos ? v : alpha * v;
        colbits |= (pos ? 1u : 0u) << (8 * (J >> 2) + (J & 3));   // this lane's channel, pixel rows 8 (J / 4) + J % 4 (+ 4 half)
        if (want_mask) {                                         // (wave-uniform)
            const unsigned long long bal = __ballot(pos);
            constexpr int r0 = 8 * (J >> 2) + (J & 3);
            // (s_nop: the ballot is a VALU write of an SGPR pair; v_writelane reading it as DATA right behind it saw the old value
            //  on gfx950 - the compiler inserts no wait states for operands of inline asm)
            asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %2\n\tv_writelane_b32 %0, %3, %4"
                         : "+v"(w) : "s"((unsigned)bal), "n"(r0), "s"((unsigned)(bal >> 32)), "n"(r0 + 4));
        }
        head_rows<J + 1>(a, bias, alpha, want_mask, s, w, colbits);
    }
}

template <int NW, int NF, int MODE>
__global__ __launch_bounds__(NW * 64, 2) void head_kernel(const HeadParams p) {
    constexpr int C = NF * 32, KC = C / 64, NT = NW * 64, HW = NW * 32;
    constexpr int BI = C * 8 / NT;                               // 16-byte items of a weight chunk per thread
    constexpr int SB = C * ROWB;                                 // (behind it - sA - fwd: the input chunk; dgrad: g and alpha g)
    constexpr int EPI = MODE == 1 ? NW * 32 * ROWF : 0;          // dgrad epilogue: [32][64 + 4] float32 per wave
    static_assert(C * 8 % NT == 0, "weight chunk divides over the threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* sB = smem_raw;
    unsigned char* sA = smem_raw + (SB > EPI ? SB : EPI);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x;
    const int half = lane >> 5, l32 = lane & 31;
    const long px0 = (long)n * HW;
    const unsigned char* wimg = reinterpret_cast<const unsigned char*>(p.wimg);

    f32x16 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[f][j] = 0.f;

    // ---- staging: weight chunk kc = image sub-chunks 4 kc .. 4 kc + 3, each [C rows][16] bf16
    u32x4 preB[BI], preA[4];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int item = tid + i * NT, r = item >> 3, s = (item >> 1) & 3, h = item & 1;
            preB[i] = *reinterpret_cast<const u32x4*>(wimg + ((long)((kc * 4 + s) * C + r)) * 32 + h * 16);
        }
        if constexpr (MODE == 0) {
            const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x) + px0 * C * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {                        // HW rows x 8 items = 4 per thread
                const int item = tid + i * NT, r = item >> 3, s = item & 7;
                preA[i] = *reinterpret_cast<const u32x4*>(xb + ((long)r * C + kc * 64 + s * 8) * 2);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int item = tid + i * NT, r = item >> 3, s = (item >> 1) & 3, h = item & 1;
            *reinterpret_cast<u32x4*>(sB + r * ROWB + s * 32 + h * 16) = preB[i];
        }
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int item = tid + i * NT, r = item >> 3, s = item & 7;
                *reinterpret_cast<u32x4*>(sA + r * ROWB + s * 16) = preA[i];
            }
        }
    };

    float biasr[NF];                                             // fwd: this lane's NF output channels (a load inside the epilogue is
    if constexpr (MODE == 0) {                                   // a memory round trip on the critical path)
#pragma unroll
        for (int f = 0; f < NF; ++f) biasr[f] = p.bias[f * 32 + l32];
    }
    unsigned mrow[NF];                                           // dgrad: the C mask bits of this lane's pixel
    if constexpr (MODE == 1) {
        // g[co] = (sum_j dlogits[n][j] wdense[co][j]) / HW, the gradient of the pooled feature spread over the image's pixels
        float* sG = reinterpret_cast<float*>(sA);
        for (int co = tid; co < C; co += NT) {
            float a = 0.f;
            for (int j = 0; j < p.K; ++j) a = fmaf(p.dlogits[(long)n * p.K + j], p.wdense[(long)co * p.K + j], a);
            a /= (float)HW;
            sG[co] = a;
            sG[C + co] = a * p.alpha;
        }
        const unsigned* mr = p.mask + (px0 + wave * 32 + l32) * NF;
#pragma unroll
        for (int f = 0; f < NF; ++f) mrow[f] = mr[f];
    }
    fetch(0);
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        if (kc > 0) __syncthreads();                             // every wave is done with the previous chunk
        commit();
        __syncthreads();
        if (kc + 1 < KC) fetch(kc + 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a;
            if constexpr (MODE == 0) {
                a = *reinterpret_cast<const bf16x8*>(sA + (wave * 32 + l32) * ROWB + ks * 32 + half * 16);
            } else {
                const float* sG = reinterpret_cast<const float*>(sA);
                const int co = kc * 64 + ks * 16;                // this lane: channels co + 8 half .. + 7
                const unsigned bits = (mrow[co >> 5] >> ((co & 31) + 8 * half)) & 0xffu;
                const float4 g0 = *reinterpret_cast<const float4*>(sG + co + 8 * half);
                const float4 g1 = *reinterpret_cast<const float4*>(sG + co + 8 * half + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(sG + C + co + 8 * half);
                const float4 h1 = *reinterpret_cast<const float4*>(sG + C + co + 8 * half + 4);
                const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float hv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = (__bf16)(((bits >> e) & 1u) ? gv[e] : hv[e]);
            }
            bf16x8 bfr[NF];                                      // every fragment requested before the first product waits
#pragma unroll
            for (int f = 0; f < NF; ++f)
                bfr[f] = *reinterpret_cast<const bf16x8*>(sB + (f * 32 + l32) * ROWB + ks * 32 + half * 16);
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfr[f], acc[f], 0, 0, 0);
        }
    }
    __syncthreads();                                             // the tiles are free: the epilogues reuse them

    if constexpr (MODE == 0) {
        // accumulator element j of fragment f: channel 32 f + l32, pixel row 8 (j / 4) + 4 half + j % 4 of this wave's 32
        float* sGap = reinterpret_cast<float*>(smem_raw);        // [NW][C]
        unsigned mword[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const float bias = biasr[f];
            float s = 0.f;
            mword[f] = 0u;
            unsigned colbits = 0u;
            head_rows<0>(acc[f], bias, p.alpha, p.mask != nullptr, s, mword[f], colbits);
            s += __shfl_xor(s, 32, 64);                          // the two row halves of the fragment
            if (half == 0) sGap[wave * C + f * 32 + l32] = s;
            if (p.mask_p) {                                      // the same bits, pixel-major: one word per (32-pixel block, channel)
                colbits <<= 4 * half;
                colbits |= (unsigned)__shfl_xor((int)colbits, 32, 64);
                if (half == 0) p.mask_p[((long)n * NW + wave) * C + f * 32 + l32] = colbits;
            }
        }
        if (p.mask && half == 0) {
            unsigned* mr = p.mask + (px0 + wave * 32 + l32) * NF;
#pragma unroll
            for (int f = 0; f < NF; ++f) mr[f] = mword[f];
        }
        __syncthreads();
        for (int co = tid; co < C; co += NT) {
            float s = sGap[co];
#pragma unroll
            for (int w = 1; w < NW; ++w) s += sGap[w * C + co];
            p.gap[(long)n * C + co] = s / (float)HW;
        }
    } else {
        // dX tile of this wave, 64 channels at a time through its own LDS scratch: written from the accumulator layout, read back
        // as 16-byte runs along the channels, times LeakyReLU' of the layer below, stored as 16-byte runs
        unsigned char* scr = smem_raw + wave * (32 * ROWF);
        const unsigned char* im = reinterpret_cast<const unsigned char*>(p.in_mask);
        unsigned char* dx = reinterpret_cast<unsigned char*>(p.dx);
#pragma unroll
        for (int fp = 0; fp < NF / 2; ++fp) {
#pragma unroll
            for (int ff = 0; ff < 2; ++ff)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int row = 8 * (j >> 2) + 4 * half + (j & 3);
                    *reinterpret_cast<float*>(scr + row * ROWF + (ff * 32 + l32) * 4) = acc[2 * fp + ff][j];
                }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = i * 64 + lane, row = idx >> 3, seg = idx & 7;
                const float4 v0 = *reinterpret_cast<const float4*>(scr + row * ROWF + seg * 32);
                const float4 v1 = *reinterpret_cast<const float4*>(scr + row * ROWF + seg * 32 + 16);
                float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                const long o = ((px0 + wave * 32 + row) * C + fp * 64 + seg * 8) * 2;
                if (im) {
                    const bf16x8 m = *reinterpret_cast<const bf16x8*>(im + o);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] *= (float)m[e] > 0.f ? 1.0f : p.alpha;
                }
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)f[e];
                *reinterpret_cast<bf16x8*>(dx + o) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- weight gradient of the fused head's 1x1 layer: dW[ci][co] = sum over pixels of X[px][ci] dAct[px][co], db[co] = sum dAct.
// Both operands of a weight gradient are pixel-major while the matrix instruction wants 8 consecutive K (= pixels) per lane:
// X is TRANSPOSED while it is staged (two pixels per thread and channel packed into one dword: [ci][pixel] rows in LDS), dAct
// is BUILT transposed from the pixel-major sign words (mask_p) and g - it never exists in HBM.  16-byte column blocks of a row are
// XOR-swizzled by (row / 8) & 7: the eight channel groups a wave writes at one pixel position land in eight different bank
// groups.  A workgroup (one wave per 32 input channels) owns the whole C x C gradient for its share of the images, PB pixels at
// a time; one slab per workgroup, the library's fixed-order slab reduction behind it.
struct HeadWParams {
    const void* x;            // (N * HW, C) bf16
    const unsigned* mask_p;   // (N, HW / 32, C) words
    const float* dlogits;     // (N, K)
    const float* wdense;      // (C, K)
    float* partial;           // [S][C][C]
    float* db_partial;        // [S][C]
    int N, HW, K, S;
    float alpha;
};

template <int NF, int PB>
__global__ __launch_bounds__(NF * 64, 2) void head_wgrad_kernel(const HeadWParams p) {
    constexpr int C = NF * 32, NT = NF * 64, RB = PB * 2 + 16, CB = PB / 8;      // row bytes, 16-byte column blocks per row
    static_assert(CB >= 8 && (PB % 32) == 0, "swizzle over 8 column blocks; whole mask words");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* sX = smem_raw;                                // [C][PB] bf16 (+ pad), swizzled
    unsigned char* sD = smem_raw + C * RB;                       // [C][PB] bf16 (+ pad), swizzled
    float* sG = reinterpret_cast<float*>(smem_raw + 2 * C * RB); // g, alpha g
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int nblk = p.HW / PB, wpb = PB / 32;                   // stages per image, mask words per stage and channel
    f32x16 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[f][j] = 0.f;
    float dbacc = 0.f;
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x);
    for (int n = blockIdx.x; n < p.N; n += p.S) {
        __syncthreads();                                         // sG (and the tiles) of the previous image are done with
        for (int co = tid; co < C; co += NT) {
            float a = 0.f;
            for (int j = 0; j < p.K; ++j) a = fmaf(p.dlogits[(long)n * p.K + j], p.wdense[(long)co * p.K + j], a);
            a /= (float)p.HW;
            sG[co] = a;
            sG[C + co] = a * p.alpha;
            int cnt = 0;                                         // bias gradient: g (count of positives) + alpha g (the rest)
            for (int b = 0; b < p.HW / 32; ++b) cnt += __builtin_popcount(p.mask_p[((long)n * (p.HW / 32) + b) * C + co]);
            dbacc += a * (float)cnt + a * p.alpha * (float)(p.HW - cnt);
        }
        __syncthreads();                                         // g is read by every thread that builds dAct
        for (int blk = 0; blk < nblk; ++blk) {
            if (blk > 0) __syncthreads();
            // X, transposed: item = (pixel pair pp, 8-channel group cg); lanes: 8 consecutive cg x 8 consecutive pp
            constexpr int XI = (PB / 2) * (C / 8) / NT;
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const int item = tid + i * NT;
                const int cg = (item & 7) | ((item >> 6) % (C / 64)) << 3, pp = ((item >> 3) & 7) | ((item >> 6) / (C / 64)) << 3;
                const long px = (long)n * p.HW + blk * PB + 2 * pp;
                const u32x4 v0 = *reinterpret_cast<const u32x4*>(xb + (px * C + cg * 8) * 2);
                const u32x4 v1 = *reinterpret_cast<const u32x4*>(xb + ((px + 1) * C + cg * 8) * 2);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned a = v0[e >> 1], b = v1[e >> 1];
                    const unsigned d = (e & 1) ? __builtin_amdgcn_perm(b, a, 0x07060302u) : __builtin_amdgcn_perm(b, a, 0x05040100u);
                    const int row = cg * 8 + e;
                    *reinterpret_cast<unsigned*>(sX + row * RB + (((pp >> 2) ^ (cg & 7)) << 4) + (pp & 3) * 4) = d;
                }
            }
            // dAct, built transposed: item = (channel co, 8-pixel group g8)
            constexpr int DI = C * CB / NT;
#pragma unroll
            for (int i = 0; i < DI; ++i) {
                const int item = tid + i * NT, co = item % C, g8 = item / C;
                const unsigned word = p.mask_p[((long)n * (p.HW / 32) + blk * wpb + (g8 >> 2)) * C + co];
                const unsigned bits = (word >> (8 * (g8 & 3))) & 0xffu;
                const float gv = sG[co], hv = sG[C + co];
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)(((bits >> e) & 1u) ? gv : hv);
                *reinterpret_cast<bf16x8*>(sD + co * RB + ((g8 ^ ((co >> 3) & 7)) << 4)) = v;
            }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < PB / 16; ++ks) {
                const int kg = 2 * ks + half, ci = wave * 32 + l32;
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(sX + ci * RB + ((kg ^ ((ci >> 3) & 7)) << 4));
                bf16x8 bfr[NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int co = f * 32 + l32;
                    bfr[f] = *reinterpret_cast<const bf16x8*>(sD + co * RB + ((kg ^ ((co >> 3) & 7)) << 4));
                }
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfr[f], acc[f], 0, 0, 0);
            }
        }
    }
    float* slab = p.partial + (long)blockIdx.x * C * C;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int ci = wave * 32 + 8 * (j >> 2) + 4 * half + (j & 3);
            slab[(long)ci * C + f * 32 + l32] = acc[f][j];
        }
    if (p.db_partial && tid < C) p.db_partial[(long)blockIdx.x * C + tid] = dbacc;
}

template <int NF, int PB>
int launch_wgrad(const HeadWParams& p, hipStream_t s) {
    constexpr int C = NF * 32, RB = PB * 2 + 16;
    constexpr size_t lds = (size_t)2 * C * RB + 2 * C * 4;
    static_assert(lds <= 160 * 1024, "LDS");
    auto k = head_wgrad_kernel<NF, PB>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)p.S), dim3(NF * 64), lds, s, p);
    return hipGetLastError() == hipSuccess ? NIMG_OK : NIMG_ERR_LAUNCH;
}

static inline int head_wgrad_splits(int n) {
    const int ipw = (n + 119) / 120;               // images per workgroup: at most ~120 slabs to write and reduce
    return (n + ipw - 1) / ipw;
}

// dAct[px][co] = g[n][co] * (bit ? 1 : alpha) as bf16, for the weight-gradient kernel
__global__ __launch_bounds__(256) void head_dact_kernel(const unsigned* __restrict__ mask, const float* __restrict__ dlogits,
                                                        const float* __restrict__ wdense, void* __restrict__ dact, int hw, int c,
                                                        int k, float alpha, int parts) {
    __shared__ float sG[512];
    const int n = blockIdx.x / parts, part = blockIdx.x % parts, tid = threadIdx.x;
    for (int co = tid; co < c; co += 256) {
        float a = 0.f;
        for (int j = 0; j < k; ++j) a = fmaf(dlogits[(long)n * k + j], wdense[(long)co * k + j], a);
        a /= (float)hw;
        sG[co] = a;
        sG[256 + co] = a * alpha;
    }
    __syncthreads();
    const int c8 = c >> 3, nf = c >> 5;
    for (int i = part * 256 + tid; i < hw * c8; i += parts * 256) {           // one 16-byte run of 8 channels per item
        const int px = i / c8, co = (i % c8) * 8;
        const unsigned bits = (mask[((long)n * hw + px) * nf + (co >> 5)] >> (co & 31)) & 0xffu;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (__bf16)(((bits >> e) & 1u) ? sG[co + e] : sG[256 + co + e]);
        *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(dact) + ((long)n * hw + px) * c + co) = v;
    }
}

template <int NW, int NF, int MODE>
int launch(const HeadParams& p, hipStream_t s) {
    constexpr int C = NF * 32, HW = NW * 32;
    constexpr int SB = C * ROWB, EPI = MODE == 1 ? NW * 32 * ROWF : 0, SA = MODE == 0 ? HW * ROWB : 2 * C * 4;
    constexpr size_t lds = (size_t)(SB > EPI ? SB : EPI) + SA;
    static_assert(lds <= 160 * 1024 && (size_t)NW * C * 4 <= lds, "LDS");
    auto k = head_kernel<NW, NF, MODE>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)p.N), dim3(NW * 64), lds, s, p);
    return hipGetLastError() == hipSuccess ? NIMG_OK : NIMG_ERR_LAUNCH;
}

template <int MODE>
int dispatch(const HeadParams& p, hipStream_t s) {
#define NIMG_HEAD(NW_, NF_) if (p.HW == NW_ * 32 && p.C == NF_ * 32) return launch<NW_, NF_, MODE>(p, s)
    NIMG_HEAD(8, 8); NIMG_HEAD(4, 8); NIMG_HEAD(2, 8);
    NIMG_HEAD(8, 4); NIMG_HEAD(4, 4); NIMG_HEAD(2, 4);
    NIMG_HEAD(8, 2); NIMG_HEAD(4, 2); NIMG_HEAD(2, 2);
#undef NIMG_HEAD
    return NIMG_ERR_ARG;
}

}  // namespace

extern "C" {

int nimg_head_fused_ok(int hw, int c) {
    return (hw == 64 || hw == 128 || hw == 256) && (c == 64 || c == 128 || c == 256) ? 1 : 0;
}

int nimg_head_fwd(const void* x, const void* wimg, const float* bias, unsigned* mask, unsigned* mask_p, float* gap, int n, int hw,
                  int c, float alpha, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!x || !wimg || !bias || !gap || n < 0 || !nimg_head_fused_ok(hw, c) || (mask_p && !mask)) return NIMG_ERR_ARG;
    HeadParams p = {};
    p.x = x; p.wimg = wimg; p.bias = bias; p.mask = mask; p.mask_p = mask_p; p.gap = gap; p.N = n; p.HW = hw; p.C = c; p.alpha = alpha;
    return dispatch<0>(p, (hipStream_t)stream);
}

int nimg_head_dgrad(const unsigned* mask, const float* dlogits, const float* wdense, int k, const void* wimg_t,
                    const void* in_mask, void* dx, int n, int hw, int c, float alpha, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!mask || !dlogits || !wdense || !wimg_t || !dx || n < 0 || k <= 0 || !nimg_head_fused_ok(hw, c)) return NIMG_ERR_ARG;
    HeadParams p = {};
    p.wimg = wimg_t; p.mask = const_cast<unsigned*>(mask); p.dlogits = dlogits; p.wdense = wdense; p.in_mask = in_mask; p.dx = dx;
    p.N = n; p.HW = hw; p.C = c; p.K = k; p.alpha = alpha;
    return dispatch<1>(p, (hipStream_t)stream);
}

size_t nimg_head_wgrad_workspace_bytes(int n, int c) {
    if (n <= 0 || c <= 0) return 0;
    return (size_t)head_wgrad_splits(n) * ((size_t)c * c + c) * sizeof(float);
}

int nimg_head_wgrad(const void* x, const unsigned* mask_p, const float* dlogits, const float* wdense, int k, float* dw, float* db,
                    int n, int hw, int c, float alpha, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !mask_p || !dlogits || !wdense || !dw || !workspace || n <= 0 || k <= 0 || !nimg_head_fused_ok(hw, c))
        return NIMG_ERR_ARG;
    if (workspace_bytes < nimg_head_wgrad_workspace_bytes(n, c)) return NIMG_ERR_WORKSPACE;
    HeadWParams p;
    p.x = x; p.mask_p = mask_p; p.dlogits = dlogits; p.wdense = wdense; p.N = n; p.HW = hw; p.K = k; p.alpha = alpha;
    p.S = head_wgrad_splits(n);
    p.partial = (float*)workspace;
    p.db_partial = db ? p.partial + (size_t)p.S * c * c : nullptr;
    hipStream_t s = (hipStream_t)stream;
    int rc = NIMG_ERR_ARG;
#define NIMG_HW(NF_) \
    if (c == NF_ * 32) rc = hw == 64 ? launch_wgrad<NF_, 64>(p, s) : launch_wgrad<NF_, 128>(p, s)
    NIMG_HW(8); NIMG_HW(4); NIMG_HW(2);
#undef NIMG_HW
    if (rc != NIMG_OK) return rc;
    launch_reduce2(p.partial, dw, (long)c * c, p.S, p.db_partial, db, (long)c, p.S, accumulate, s);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

int nimg_head_dact(const unsigned* mask, const float* dlogits, const float* wdense, int k, void* dact, int n, int hw, int c,
                   float alpha, void* stream) {
    if (n == 0) return NIMG_OK;
    if (!mask || !dlogits || !wdense || !dact || n < 0 || k <= 0 || hw <= 0 || c <= 0 || c > 256 || (c & 31)) return NIMG_ERR_ARG;
    const int parts = n >= 1024 ? 1 : (n >= 256 ? 4 : 16);
    hipLaunchKernelGGL(head_dact_kernel, dim3((unsigned)(n * parts)), dim3(256), 0, (hipStream_t)stream, mask, dlogits, wdense,
                       dact, hw, c, k, alpha, parts);
    NIMG_CHECK_LAUNCH();
    return NIMG_OK;
}

}  // extern "C"
